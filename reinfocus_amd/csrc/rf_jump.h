// rf_jump.h -- host-side GF(2) jump-ahead tables for xoroshiro128+.
//
// numba seeds state[i] = jump(state[i-1]) sequentially on the host
// (numba/cuda/random.py init_xoroshiro128p_states_cpu, reached from
// graphics/random.py:18).  The state transition is linear over GF(2), so
// jump == multiplication by J = T^(2^64) where T is the 128x128 one-step matrix.
// We build T from the step function, square it 64 times to get J, check J against
// numba's published jump polynomial (0xbeac0467eba5facb, 0xd86b048b86aa9922) on
// probe states, and tabulate J^(2^k) so a GPU thread can reach state[i] directly.
#pragma once

#include <stdint.h>

#include <vector>

namespace rf {

struct S128 {
    uint64_t s0, s1;
};

inline uint64_t h_rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

inline S128 h_step(S128 s)
{
    uint64_t s0 = s.s0, s1 = s.s1;
    s1 ^= s0;
    return S128{h_rotl(s0, 55) ^ s1 ^ (s1 << 14), h_rotl(s1, 36)};
}

// numba's xoroshiro128p_jump, literal
inline S128 h_jump_reference(S128 s)
{
    static const uint64_t poly[2] = {0xbeac0467eba5facbull, 0xd86b048b86aa9922ull};
    uint64_t a0 = 0, a1 = 0;
    for (int i = 0; i < 2; ++i)
        for (int b = 0; b < 64; ++b) {
            if (poly[i] & (1ull << b)) {
                a0 ^= s.s0;
                a1 ^= s.s1;
            }
            s = h_step(s);
        }
    return S128{a0, a1};
}

inline S128 h_splitmix(uint64_t seed)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return S128{z, z};
}

struct Mat128 {
    S128 col[128]; // col[j] = image of basis vector e_j (bit j of s0, bit j-64 of s1)
};

inline S128 h_matvec(const Mat128 &m, S128 v)
{
    uint64_t r0 = 0, r1 = 0;
    for (int j = 0; j < 64; ++j)
        if ((v.s0 >> j) & 1) {
            r0 ^= m.col[j].s0;
            r1 ^= m.col[j].s1;
        }
    for (int j = 0; j < 64; ++j)
        if ((v.s1 >> j) & 1) {
            r0 ^= m.col[64 + j].s0;
            r1 ^= m.col[64 + j].s1;
        }
    return S128{r0, r1};
}

inline void h_matmul(const Mat128 &a, const Mat128 &b, Mat128 &out) // out = a * b
{
    for (int j = 0; j < 128; ++j)
        out.col[j] = h_matvec(a, b.col[j]);
}

// tables[k] = J^(2^k), k in [0, count).  Returns false if J disagrees with numba's
// jump polynomial on the probe states (would mean the step function is wrong).
inline bool h_build_jump_tables(int count, std::vector<Mat128> &tables)
{
    Mat128 p, q;
    for (int j = 0; j < 128; ++j) {
        S128 e{j < 64 ? (1ull << j) : 0ull, j < 64 ? 0ull : (1ull << (j - 64))};
        p.col[j] = h_step(e);
    }
    for (int i = 0; i < 64; ++i) { // T^(2^64)
        h_matmul(p, p, q);
        p = q;
    }
    const S128 probes[3] = {h_splitmix(0), h_splitmix(12345), S128{1ull, 0x8000000000000000ull}};
    for (const S128 &s : probes) {
        S128 a = h_matvec(p, s), b = h_jump_reference(s);
        if (a.s0 != b.s0 || a.s1 != b.s1)
            return false;
    }
    tables.resize(count);
    tables[0] = p;
    for (int k = 1; k < count; ++k)
        h_matmul(tables[k - 1], tables[k - 1], tables[k]);
    return true;
}

} // namespace rf
