// ubench.hip -- gfx950 VALU issue-rate microbenchmark (development tool, not product).
//
// For each instruction form used by the render kernel: 4 independent dependency chains,
// 64 instructions per loop trip, W waves per SIMD (W = 1, 2, 4, 8), every CU busy.
// Prints cycles per wave-instruction per SIMD, using the kernel's own s_memtime deltas.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define R4(s) s s s s
#define R16(s) R4(R4(s))

// BODY uses %0..%3 as the four chain registers (32-bit), %4/%5 as extra inputs
#define KERNEL32(NAME, BODY)                                                               \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned long long *cyc, int iters) \
    {                                                                                      \
        unsigned a = threadIdx.x * 2654435761u + 1, b = a ^ 0x9e3779b9u, c = a + 77, d = b + 99; \
        unsigned e = threadIdx.x | 1, f = 0x3f800001u;                                     \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        for (int i = 0; i < iters; ++i) {                                                  \
            asm volatile(R16(BODY) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f)); \
        }                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;                        \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                   \
    }

// 64-bit chains: %0..%3 are 64-bit register pairs
#define KERNEL64(NAME, BODY)                                                               \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned long long *cyc, int iters) \
    {                                                                                      \
        unsigned long long a = threadIdx.x * 0x9e3779b97f4a7c15ull + 1, b = a ^ 0x123456789ull, c = a + 77, d = b + 99; \
        double e = 1.0000001, f = 0.9999999;                                               \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        for (int i = 0; i < iters; ++i) {                                                  \
            asm volatile(R16(BODY) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f)); \
        }                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(a ^ b ^ c ^ d);            \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                   \
    }

#define B4(op, tail) op " %0, " tail "\n" op " %1, " tail "\n" op " %2, " tail "\n" op " %3, " tail "\n"

KERNEL32(k_xor, "v_xor_b32 %0, %0, %4\nv_xor_b32 %1, %1, %4\nv_xor_b32 %2, %2, %4\nv_xor_b32 %3, %3, %4\n")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %4, %5\nv_and_or_b32 %1, %1, %4, %5\nv_and_or_b32 %2, %2, %4, %5\nv_and_or_b32 %3, %3, %4, %5\n")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %4, 9\nv_alignbit_b32 %1, %1, %4, 9\nv_alignbit_b32 %2, %2, %4, 9\nv_alignbit_b32 %3, %3, %4, 9\n")
KERNEL32(k_lshl, "v_lshlrev_b32 %0, 14, %0\nv_lshlrev_b32 %1, 14, %1\nv_lshlrev_b32 %2, 14, %2\nv_lshlrev_b32 %3, 14, %3\n")
KERNEL32(k_add_u32, "v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %4\nv_add_u32 %2, %2, %4\nv_add_u32 %3, %3, %4\n")
KERNEL32(k_add_co, "v_add_co_u32 %0, vcc, %0, %4\nv_addc_co_u32 %1, vcc, %1, %4, vcc\nv_add_co_u32 %2, vcc, %2, %4\nv_addc_co_u32 %3, vcc, %3, %4, vcc\n")
KERNEL32(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0\nv_cvt_f32_u32 %1, %1\nv_cvt_f32_u32 %2, %2\nv_cvt_f32_u32 %3, %3\n")
KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %5, %5\nv_fma_f32 %1, %1, %5, %5\nv_fma_f32 %2, %2, %5, %5\nv_fma_f32 %3, %3, %5, %5\n")
KERNEL32(k_mul_f32, "v_mul_f32 %0, %0, %5\nv_mul_f32 %1, %1, %5\nv_mul_f32 %2, %2, %5\nv_mul_f32 %3, %3, %5\n")
KERNEL32(k_add_f32, "v_add_f32 %0, %0, %5\nv_add_f32 %1, %1, %5\nv_add_f32 %2, %2, %5\nv_add_f32 %3, %3, %5\n")
KERNEL32(k_ldexp_f32, "v_ldexp_f32 %0, %0, %4\nv_ldexp_f32 %1, %1, %4\nv_ldexp_f32 %2, %2, %4\nv_ldexp_f32 %3, %3, %4\n")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %4, vcc\nv_cndmask_b32 %1, %1, %4, vcc\nv_cndmask_b32 %2, %2, %4, vcc\nv_cndmask_b32 %3, %3, %4, vcc\n")
KERNEL32(k_cmp_cnd, "v_cmp_lt_u32 vcc, %0, %4\nv_cndmask_b32 %1, %1, %4, vcc\nv_cmp_lt_u32 vcc, %2, %4\nv_cndmask_b32 %3, %3, %4, vcc\n")
KERNEL32(k_ffbh, "v_ffbh_u32 %0, %0\nv_ffbh_u32 %1, %1\nv_ffbh_u32 %2, %2\nv_ffbh_u32 %3, %3\n")
KERNEL32(k_bfe, "v_bfe_u32 %0, %0, 11, 5\nv_bfe_u32 %1, %1, 11, 5\nv_bfe_u32 %2, %2, 11, 5\nv_bfe_u32 %3, %3, 11, 5\n")
KERNEL32(k_min_u32, "v_min_u32 %0, %0, %4\nv_min_u32 %1, %1, %4\nv_min_u32 %2, %2, %4\nv_min_u32 %3, %3, %4\n")
KERNEL32(k_rcp_f32, "v_rcp_f32 %0, %0\nv_rcp_f32 %1, %1\nv_rcp_f32 %2, %2\nv_rcp_f32 %3, %3\n")
KERNEL32(k_sqrt_f32, "v_sqrt_f32 %0, %0\nv_sqrt_f32 %1, %1\nv_sqrt_f32 %2, %2\nv_sqrt_f32 %3, %3\n")
KERNEL32(k_floor_f32, "v_floor_f32 %0, %0\nv_floor_f32 %1, %1\nv_floor_f32 %2, %2\nv_floor_f32 %3, %3\n")
KERNEL32(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0\nv_cvt_i32_f32 %1, %1\nv_cvt_i32_f32 %2, %2\nv_cvt_i32_f32 %3, %3\n")
KERNEL32(k_div_scale, "v_div_scale_f32 %0, vcc, %0, %5, %0\nv_div_scale_f32 %1, vcc, %1, %5, %1\nv_div_scale_f32 %2, vcc, %2, %5, %2\nv_div_scale_f32 %3, vcc, %3, %5, %3\n")
KERNEL32(k_div_fixup, "v_div_fixup_f32 %0, %0, %5, %5\nv_div_fixup_f32 %1, %1, %5, %5\nv_div_fixup_f32 %2, %2, %5, %5\nv_div_fixup_f32 %3, %3, %5, %5\n")
KERNEL32(k_perm, "v_perm_b32 %0, %0, %4, %5\nv_perm_b32 %1, %1, %4, %5\nv_perm_b32 %2, %2, %4, %5\nv_perm_b32 %3, %3, %4, %5\n")
KERNEL32(k_mix_xor_fma, "v_xor_b32 %0, %0, %4\nv_fma_f32 %1, %1, %5, %5\nv_xor_b32 %2, %2, %4\nv_fma_f32 %3, %3, %5, %5\n")
KERNEL32(k_mix_xor_alignbit, "v_xor_b32 %0, %0, %4\nv_alignbit_b32 %1, %1, %4, 9\nv_xor_b32 %2, %2, %4\nv_alignbit_b32 %3, %3, %4, 9\n")

KERNEL32(k_and, "v_and_b32 %0, %0, %4\nv_and_b32 %1, %1, %4\nv_and_b32 %2, %2, %4\nv_and_b32 %3, %3, %4\n")
KERNEL32(k_or, "v_or_b32 %0, %0, %4\nv_or_b32 %1, %1, %4\nv_or_b32 %2, %2, %4\nv_or_b32 %3, %3, %4\n")
KERNEL32(k_sub_u32, "v_sub_u32 %0, %0, %4\nv_sub_u32 %1, %1, %4\nv_sub_u32 %2, %2, %4\nv_sub_u32 %3, %3, %4\n")
KERNEL32(k_mov, "v_mov_b32 %0, %4\nv_mov_b32 %1, %4\nv_mov_b32 %2, %4\nv_mov_b32 %3, %4\n")
KERNEL32(k_fmac, "v_fmac_f32 %0, %5, %5\nv_fmac_f32 %1, %5, %5\nv_fmac_f32 %2, %5, %5\nv_fmac_f32 %3, %5, %5\n")
KERNEL32(k_fma_distinct, "v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n")
KERNEL32(k_mul_u24, "v_mul_u32_u24 %0, %0, %4\nv_mul_u32_u24 %1, %1, %4\nv_mul_u32_u24 %2, %2, %4\nv_mul_u32_u24 %3, %3, %4\n")
KERNEL32(k_mad_u24, "v_mad_u32_u24 %0, %0, %4, %5\nv_mad_u32_u24 %1, %1, %4, %5\nv_mad_u32_u24 %2, %2, %4, %5\nv_mad_u32_u24 %3, %3, %4, %5\n")
KERNEL32(k_mul_lo, "v_mul_lo_u32 %0, %0, %4\nv_mul_lo_u32 %1, %1, %4\nv_mul_lo_u32 %2, %2, %4\nv_mul_lo_u32 %3, %3, %4\n")
KERNEL32(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %4\nv_lshl_or_b32 %1, %1, 3, %4\nv_lshl_or_b32 %2, %2, 3, %4\nv_lshl_or_b32 %3, %3, 3, %4\n")
KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %4\nv_lshl_add_u32 %1, %1, 3, %4\nv_lshl_add_u32 %2, %2, 3, %4\nv_lshl_add_u32 %3, %3, 3, %4\n")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %4, %5\nv_add3_u32 %1, %1, %4, %5\nv_add3_u32 %2, %2, %4, %5\nv_add3_u32 %3, %3, %4, %5\n")
KERNEL32(k_bfi, "v_bfi_b32 %0, %0, %4, %5\nv_bfi_b32 %1, %1, %4, %5\nv_bfi_b32 %2, %2, %4, %5\nv_bfi_b32 %3, %3, %4, %5\n")
KERNEL32(k_max_f32, "v_max_f32 %0, %0, %5\nv_max_f32 %1, %1, %5\nv_max_f32 %2, %2, %5\nv_max_f32 %3, %3, %5\n")
KERNEL32(k_max_u32, "v_max_u32 %0, %0, %4\nv_max_u32 %1, %1, %4\nv_max_u32 %2, %2, %4\nv_max_u32 %3, %3, %4\n")
KERNEL32(k_sub_f32, "v_sub_f32 %0, %0, %5\nv_sub_f32 %1, %1, %5\nv_sub_f32 %2, %2, %5\nv_sub_f32 %3, %3, %5\n")
KERNEL32(k_lshr, "v_lshrrev_b32 %0, 9, %0\nv_lshrrev_b32 %1, 9, %1\nv_lshrrev_b32 %2, 9, %2\nv_lshrrev_b32 %3, 9, %3\n")
KERNEL32(k_cmp_only, "v_cmp_lt_u32 vcc, %0, %4\nv_cmp_lt_u32 vcc, %1, %4\nv_cmp_lt_u32 vcc, %2, %4\nv_cmp_lt_u32 vcc, %3, %4\n")
KERNEL32(k_xor_chain1, "v_xor_b32 %0, %0, %4\nv_xor_b32 %0, %0, %5\nv_xor_b32 %0, %0, %4\nv_xor_b32 %0, %0, %5\n")
KERNEL32(k_alignbit_chain1, "v_alignbit_b32 %0, %0, %4, 9\nv_alignbit_b32 %0, %0, %4, 9\nv_alignbit_b32 %0, %0, %4, 9\nv_alignbit_b32 %0, %0, %4, 9\n")
KERNEL64(k_lshl_b64, "v_lshlrev_b64 %0, 14, %0\nv_lshlrev_b64 %1, 14, %1\nv_lshlrev_b64 %2, 14, %2\nv_lshlrev_b64 %3, 14, %3\n")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %4\nv_lshl_add_u64 %1, %1, 0, %4\nv_lshl_add_u64 %2, %2, 0, %4\nv_lshl_add_u64 %3, %3, 0, %4\n")
KERNEL64(k_mul_f64, "v_mul_f64 %0, %0, %4\nv_mul_f64 %1, %1, %4\nv_mul_f64 %2, %2, %4\nv_mul_f64 %3, %3, %4\n")
KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %4, %5\nv_fma_f64 %1, %1, %4, %5\nv_fma_f64 %2, %2, %4, %5\nv_fma_f64 %3, %3, %4, %5\n")
KERNEL64(k_add_f64, "v_add_f64 %0, %0, %4\nv_add_f64 %1, %1, %4\nv_add_f64 %2, %2, %4\nv_add_f64 %3, %3, %4\n")
KERNEL64(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n")
KERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\nv_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5\n")

// conversions between widths: separate source/dest
__global__ __launch_bounds__(256) void k_cvt_f64_f32(unsigned *out, unsigned long long *cyc, int iters)
{
    float a = threadIdx.x + 0.5f, b = a + 1, c = a + 2, d = a + 3;
    double A = 0, B = 0, C = 0, D = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(R16("v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %5\nv_cvt_f64_f32 %2, %6\nv_cvt_f64_f32 %3, %7\n")
                     : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(a), "v"(b), "v"(c), "v"(d));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(A + B + C + D);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(256) void k_cvt_f32_f64(unsigned *out, unsigned long long *cyc, int iters)
{
    double a = threadIdx.x + 0.5, b = a + 1, c = a + 2, d = a + 3;
    float A = 0, B = 0, C = 0, D = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(R16("v_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %5\nv_cvt_f32_f64 %2, %6\nv_cvt_f32_f64 %3, %7\n")
                     : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(a), "v"(b), "v"(c), "v"(d));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(A + B + C + D);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*kern_t)(unsigned *, unsigned long long *, int);
struct Entry { const char *name; kern_t fn; };

int main()
{
    Entry entries[] = {
        {"v_xor_b32", k_xor}, {"v_and_or_b32", k_and_or}, {"v_alignbit_b32", k_alignbit}, {"v_lshlrev_b32", k_lshl},
        {"v_add_u32", k_add_u32}, {"v_add_co/addc", k_add_co}, {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_fma_f32", k_fma_f32},
        {"v_mul_f32", k_mul_f32}, {"v_add_f32", k_add_f32}, {"v_ldexp_f32", k_ldexp_f32}, {"v_cndmask_b32", k_cndmask},
        {"v_cmp+v_cndmask", k_cmp_cnd}, {"v_ffbh_u32", k_ffbh}, {"v_bfe_u32", k_bfe}, {"v_min_u32", k_min_u32},
        {"v_rcp_f32", k_rcp_f32}, {"v_sqrt_f32", k_sqrt_f32}, {"v_floor_f32", k_floor_f32}, {"v_cvt_i32_f32", k_cvt_i32_f32},
        {"v_div_scale_f32", k_div_scale}, {"v_div_fixup_f32", k_div_fixup}, {"v_perm_b32", k_perm},
        {"v_and_b32", k_and}, {"v_or_b32", k_or}, {"v_sub_u32", k_sub_u32}, {"v_mov_b32", k_mov}, {"v_fmac_f32", k_fmac},
        {"v_fma_f32 (3 regs)", k_fma_distinct}, {"v_mul_u32_u24", k_mul_u24}, {"v_mad_u32_u24", k_mad_u24}, {"v_mul_lo_u32", k_mul_lo},
        {"v_lshl_or_b32", k_lshl_or}, {"v_lshl_add_u32", k_lshl_add}, {"v_add3_u32", k_add3}, {"v_bfi_b32", k_bfi},
        {"v_max_f32", k_max_f32}, {"v_max_u32", k_max_u32}, {"v_sub_f32", k_sub_f32}, {"v_lshrrev_b32", k_lshr}, {"v_cmp_lt_u32", k_cmp_only},
        {"xor 1 chain", k_xor_chain1}, {"alignbit 1 chain", k_alignbit_chain1},
        {"mix xor+fma", k_mix_xor_fma}, {"mix xor+alignbit", k_mix_xor_alignbit},
        {"v_lshlrev_b64", k_lshl_b64}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_mul_f64", k_mul_f64}, {"v_fma_f64", k_fma_f64},
        {"v_add_f64", k_add_f64}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_fma_f32", k_pk_fma_f32},
        {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f32_f64", k_cvt_f32_f64},
    };
    const int iters = 2000;       // x 64 instructions
    const int maxblocks = 256 * 8;
    unsigned *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, (size_t)maxblocks * 256 * 4));
    CHECK(hipMalloc(&cyc, (size_t)maxblocks * 8));
    std::vector<unsigned long long> h(maxblocks);
    printf("%-20s %11s %11s %11s %11s   (memtime/wall@2.4GHz cycles per wave-instruction per SIMD; s_memtime ticks)\n", "instruction", "1w/SIMD", "2w/SIMD", "4w/SIMD", "8w/SIMD");
    for (Entry &en : entries) {
        printf("%-20s", en.name);
        for (int w : {1, 2, 4, 8}) {
            int blocks = 256 * w;
            hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, out, cyc, 10);   // warm
            CHECK(hipDeviceSynchronize());
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.begin() + blocks);
            double med = (double)h[blocks / 2];
            // each wave issued iters*64 instructions; w waves share a SIMD
            double per = med / ((double)iters * 64.0 * w);
            double wall = (double)ms * 1e-3 * 2.4e9 / ((double)iters * 64.0 * w);
            printf(" %5.2f/%5.2f", per, wall);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
