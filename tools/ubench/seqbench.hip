// seqbench.hip -- how the ORDER of fast-class (v_xor ...) and slow-class (v_alignbit ...) VALU
// instructions within a wave changes their cost on gfx950, plus a few candidate instructions for the
// xoroshiro step (v_bitop3_b32 as a three-way xor, v_lshrrev_b64).  Development tool, not product.
// Prints wall-clock cycles (at the clock given on the command line, default 2.4 GHz) per
// wave-instruction per SIMD at 1 / 2 / 4 / 8 waves per SIMD; 8 independent chains per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define X(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 9\n"
#define B(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n"
#define F(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define C(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define L(i) "v_lshrrev_b32 %" #i ", 9, %" #i "\n"
#define M(i) "v_mul_f32 %" #i ", %" #i ", %9\n"

#define KERNEL(NAME, BODY, N)                                                                     \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, int iters)                          \
    {                                                                                             \
        unsigned r0 = threadIdx.x * 2654435761u + 1, r1 = r0 ^ 0x9e3779b9u, r2 = r0 + 77, r3 = r1 + 99, \
                 r4 = r0 * 3, r5 = r1 * 5, r6 = r2 * 7, r7 = r3 * 11;                              \
        unsigned e = threadIdx.x | 1, f = 0x3f800001u;                                            \
        for (int i = 0; i < iters; ++i) {                                                         \
            asm volatile(BODY BODY BODY BODY : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), \
                         "+v"(r7) : "v"(e), "v"(f));                                             \
        }                                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;        \
    }                                                                                             \
    static const int NAME##_n = 4 * (N);

KERNEL(k_x8, X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7), 8)
KERNEL(k_a8, A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7), 8)
KERNEL(k_xa, X(0) A(1) X(2) A(3) X(4) A(5) X(6) A(7), 8)
KERNEL(k_xxaa, X(0) X(1) A(2) A(3) X(4) X(5) A(6) A(7), 8)
KERNEL(k_x4a4, X(0) X(1) X(2) X(3) A(4) A(5) A(6) A(7), 8)
KERNEL(k_x6a2, X(0) X(1) X(2) X(3) X(4) X(5) A(6) A(7), 8)
KERNEL(k_x2a6, X(0) X(1) A(2) A(3) A(4) A(5) A(6) A(7), 8)
KERNEL(k_xxxa, X(0) X(1) X(2) A(3) X(4) X(5) X(6) A(7), 8)
KERNEL(k_b8, B(0) B(1) B(2) B(3) B(4) B(5) B(6) B(7), 8)
KERNEL(k_ba, B(0) A(1) B(2) A(3) B(4) A(5) B(6) A(7), 8)
KERNEL(k_bx, B(0) X(1) B(2) X(3) B(4) X(5) B(6) X(7), 8)
KERNEL(k_f8, F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7), 8)
KERNEL(k_fa, F(0) A(1) F(2) A(3) F(4) A(5) F(6) A(7), 8)
KERNEL(k_fc, F(0) C(1) F(2) C(3) F(4) C(5) F(6) C(7), 8)
KERNEL(k_ffcc, F(0) F(1) C(2) C(3) F(4) F(5) C(6) C(7), 8)
KERNEL(k_l8, L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7), 8)
KERNEL(k_xm, X(0) M(1) X(2) M(3) X(4) M(5) X(6) M(7), 8)

// one xoroshiro128+ step per state, 2 states per wave-iteration slot, in the order hipcc emits today
// (s0 = {%0,%1}, s1 = {%2,%3}; second state {%4..%7}); the sum goes to a scratch pair
#define STEP_NOW(a0, a1, b0, b1)                                                                  \
    "v_lshl_add_u64 v[20:21], v[" b0 ":" b1 "], 0, v[" a0 ":" a1 "]\n"                               \
    "v_xor_b32 v" b1 ", v" b1 ", v" a1 "\n"                                                       \
    "v_xor_b32 v" b0 ", v" b0 ", v" a0 "\n"                                                       \
    "v_alignbit_b32 v22, v" a1 ", v" a0 ", 9\n"                                                   \
    "v_alignbit_b32 v23, v" a0 ", v" a1 ", 9\n"                                                   \
    "v_lshlrev_b64 v[" a0 ":" a1 "], 14, v[" b0 ":" b1 "]\n"                                        \
    "v_xor_b32 v" a0 ", v22, v" a0 "\n"                                                           \
    "v_xor_b32 v" a1 ", v23, v" a1 "\n"                                                           \
    "v_xor_b32 v" a0 ", v" a0 ", v" b0 "\n"                                                       \
    "v_xor_b32 v" a1 ", v" a1 ", v" b1 "\n"                                                       \
    "v_alignbit_b32 v23, v" b0 ", v" b1 ", 28\n"                                                  \
    "v_alignbit_b32 v" b0 ", v" b1 ", v" b0 ", 28\n"                                              \
    "v_mov_b32 v" b1 ", v23\n"
// the same with the two-xor tails as one v_bitop3_b32 each
#define STEP_B3(a0, a1, b0, b1)                                                                   \
    "v_lshl_add_u64 v[20:21], v[" b0 ":" b1 "], 0, v[" a0 ":" a1 "]\n"                               \
    "v_xor_b32 v" b1 ", v" b1 ", v" a1 "\n"                                                       \
    "v_xor_b32 v" b0 ", v" b0 ", v" a0 "\n"                                                       \
    "v_alignbit_b32 v22, v" a1 ", v" a0 ", 9\n"                                                   \
    "v_alignbit_b32 v23, v" a0 ", v" a1 ", 9\n"                                                   \
    "v_lshlrev_b64 v[" a0 ":" a1 "], 14, v[" b0 ":" b1 "]\n"                                        \
    "v_bitop3_b32 v" a0 ", v22, v" a0 ", v" b0 " bitop3:0x96\n"                                    \
    "v_bitop3_b32 v" a1 ", v23, v" a1 ", v" b1 " bitop3:0x96\n"                                    \
    "v_alignbit_b32 v23, v" b0 ", v" b1 ", 28\n"                                                  \
    "v_alignbit_b32 v" b0 ", v" b1 ", v" b0 ", 28\n"                                              \
    "v_mov_b32 v" b1 ", v23\n"

#define RNGKERNEL(NAME, BODY, N)                                                                  \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, int iters)                          \
    {                                                                                             \
        unsigned s = threadIdx.x * 2654435761u + 1;                                               \
        unsigned acc;                                                                             \
        asm volatile("v_mov_b32 v0, %1\nv_xor_b32 v1, 0x1234567, %1\nv_add_u32 v2, 77, %1\nv_xor_b32 v3, 0x7654321, %1\n" \
                     "v_add_u32 v4, 5, %1\nv_xor_b32 v5, 0x2222222, %1\nv_add_u32 v6, 9, %1\nv_xor_b32 v7, 0x3333333, %1\n" \
                     "v_add_u32 v8, 15, %1\nv_xor_b32 v9, 0x4444444, %1\nv_add_u32 v10, 19, %1\nv_xor_b32 v11, 0x5555555, %1\n" \
                     "v_mov_b32 v24, 0\n"                                                         \
                     "s_mov_b32 s20, %2\n"                                                        \
                     "1:\n" BODY BODY                                                             \
                     "v_xor_b32 v24, v24, v21\n"                                                  \
                     "s_sub_u32 s20, s20, 1\ns_cmp_lg_u32 s20, 0\ns_cbranch_scc1 1b\n"            \
                     "v_xor_b32 %0, v24, v0\n"                                                    \
                     : "=v"(acc) : "v"(s), "s"(iters)                                            \
                     : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v20", "v21", "v22", "v23", \
                       "v24", "s20", "scc");                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                         \
    }                                                                                             \
    static const int NAME##_n = 2 * (N);

RNGKERNEL(k_rng_now, STEP_NOW("0", "1", "2", "3") STEP_NOW("4", "5", "6", "7") STEP_NOW("8", "9", "10", "11"), 3 * 13 )
RNGKERNEL(k_rng_b3, STEP_B3("0", "1", "2", "3") STEP_B3("4", "5", "6", "7") STEP_B3("8", "9", "10", "11"), 3 * 11 )

typedef void (*kern_t)(unsigned *, int);
struct Entry { const char *name; kern_t fn; int per_iter; };

int main(int argc, char **argv)
{
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
#define E(label, k) {label, k, k##_n}
    Entry entries[] = {
        E("xor x8 (fast)", k_x8), E("alignbit x8 (slow)", k_a8), E("x a x a x a x a", k_xa), E("x x a a x x a a", k_xxaa),
        E("x x x x a a a a", k_x4a4), E("x x x x x x a a", k_x6a2), E("x x a a a a a a", k_x2a6), E("x x x a x x x a", k_xxxa),
        E("bitop3 x8", k_b8), E("bitop3 / alignbit", k_ba), E("bitop3 / xor", k_bx), E("fma x8 (3 regs)", k_f8),
        E("fma / alignbit", k_fa), E("fma / cvt", k_fc), E("f f c c", k_ffcc), E("lshrrev x8", k_l8), E("xor / mul_f32", k_xm),
        E("xoroshiro step (as compiled, 13 inst)", k_rng_now), E("xoroshiro step with bitop3 (11 inst)", k_rng_b3),
    };
    const int iters = 4000;
    const int maxblocks = 256 * 8;
    unsigned *out;
    CHECK(hipMalloc(&out, (size_t)maxblocks * 256 * 4));
    printf("%-42s %9s %9s %9s %9s   (cycles @ %.2f GHz per wave-instruction per SIMD; last column pair: cycles per loop body)\n",
           "sequence", "1w/SIMD", "2w/SIMD", "4w/SIMD", "8w/SIMD", ghz);
    for (Entry &en : entries) {
        printf("%-42s", en.name);
        double last = 0;
        for (int w : {1, 2, 4, 8}) {
            const int blocks = 256 * w;
            hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, out, 10);
            CHECK(hipDeviceSynchronize());
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, out, iters);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            last = (double)ms * 1e-3 * ghz * 1e9 / ((double)iters * en.per_iter * w);
            printf(" %9.2f", last);
        }
        printf("   %8.1f\n", last * en.per_iter);
        fflush(stdout);
    }
    return 0;
}
