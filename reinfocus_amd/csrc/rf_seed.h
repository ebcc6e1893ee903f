// rf_seed.h -- seed_kernel: make_random_states (graphics/random.py:8-18) by GF(2) jump-ahead (tables: rf_jump.h)
#pragma once

#include "rf_common.h"

namespace rf {

// ---------------------------------------------------------------------------
// seeding: state[i] = J^(first + i) * s_init over GF(2), J = 2^64-step jump matrix.
// mats[k] = J^(2^k) as 128 columns of 128 bits.  A wave owns 64*R consecutive
// states: lane l starts at base + l and strides by 64 (= mats[6]), so every store
// instruction writes 1 KiB contiguous.
// ---------------------------------------------------------------------------
constexpr int kSeedMats = 48;
constexpr int kSeedRun = 32;

__device__ __forceinline__ ulonglong2 gf2_matvec(const ulonglong2 *__restrict__ cols, ulonglong2 v)
{
    unsigned long long r0 = 0, r1 = 0;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {
        const ulonglong2 c = cols[j];
        const unsigned long long m = 0ull - ((v.x >> j) & 1ull);
        r0 ^= c.x & m;
        r1 ^= c.y & m;
    }
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {
        const ulonglong2 c = cols[64 + j];
        const unsigned long long m = 0ull - ((v.y >> j) & 1ull);
        r0 ^= c.x & m;
        r1 ^= c.y & m;
    }
    return make_ulonglong2(r0, r1);
}

__global__ __launch_bounds__(kBlock) void seed_kernel(ulonglong2 *states, unsigned long long n,
                                                     unsigned long long first, ulonglong2 s_init,
                                                     const ulonglong2 *__restrict__ mats)
{
    const unsigned long long gid = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    const unsigned long long wave = gid >> 6;
    const unsigned lane = (unsigned)(gid & 63);
    const unsigned long long base = wave * (64ull * kSeedRun);
    if (base >= n)
        return;
    unsigned long long i = base + lane;
    const unsigned long long gidx = first + i;

    ulonglong2 s = s_init;
    for (int k = 0; k < kSeedMats; ++k) {
        // wave-level skip keeps the matrix loads scalar and skips unused high bits
        const bool bit = (gidx >> k) & 1ull;
        if (__any(bit)) {
            const ulonglong2 t = gf2_matvec(mats + k * 128, s);
            if (bit)
                s = t;
        }
    }
    for (int j = 0; j < kSeedRun; ++j) {
        if (i < n)
            states[i] = s;
        i += 64;
        if (j + 1 < kSeedRun)
            s = gf2_matvec(mats + 6 * 128, s);
    }
}

} // namespace rf
