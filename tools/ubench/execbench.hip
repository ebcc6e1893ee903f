// execbench.hip -- does gfx950 issue a VALU instruction faster when only part of EXEC is set?
// (development tool, not product).  Same chains as ubench.hip, run under a lane mask:
// contiguous low lanes [0, n) and a strided mask with the same population.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define R4(s) s s s s
#define R16(s) R4(R4(s))

#define KERNEL(NAME, BODY)                                                                 \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, int iters, unsigned long long mask, int mixed) \
    {                                                                                      \
        unsigned a = threadIdx.x * 2654435761u + 1, b = a ^ 0x9e3779b9u, c = a + 77, d = b + 99; \
        unsigned e = threadIdx.x | 1, f = 0x3f800001u;                                     \
        /* mixed: only the odd waves of a block are masked, the even ones run full */      \
        const bool full = mixed && !((threadIdx.x >> 6) & 1);                              \
        if (full || ((mask >> (threadIdx.x & 63)) & 1)) {                                  \
            for (int i = 0; i < iters; ++i) {                                              \
                asm volatile(R16(BODY) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f)); \
            }                                                                              \
        }                                                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;                        \
    }

KERNEL(k_xor, "v_xor_b32 %0, %0, %4\nv_xor_b32 %1, %1, %4\nv_xor_b32 %2, %2, %4\nv_xor_b32 %3, %3, %4\n")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %4, 9\nv_alignbit_b32 %1, %1, %4, 9\nv_alignbit_b32 %2, %2, %4, 9\nv_alignbit_b32 %3, %3, %4, 9\n")
KERNEL(k_fma, "v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n")
KERNEL(k_cvt, "v_cvt_f32_u32 %0, %0\nv_cvt_f32_u32 %1, %1\nv_cvt_f32_u32 %2, %2\nv_cvt_f32_u32 %3, %3\n")

typedef void (*kern_t)(unsigned *, int, unsigned long long, int);

int main()
{
    struct { const char *name; kern_t fn; } entries[] = {{"v_xor_b32", k_xor}, {"v_alignbit_b32", k_alignbit}, {"v_fma_f32", k_fma}, {"v_cvt_f32_u32", k_cvt}};
    struct { const char *name; unsigned long long mask; } masks[] = {
        {"all64", ~0ull}, {"low32", 0xFFFFFFFFull}, {"high32", 0xFFFFFFFF00000000ull}, {"low16", 0xFFFFull}, {"low8", 0xFFull}, {"lane0", 1ull},
        {"even32", 0x5555555555555555ull}, {"every4th", 0x1111111111111111ull}, {"q0+q2", 0x0000FFFF0000FFFFull},
    };
    const int iters = 2000, w = 8, blocks = 256 * w;
    unsigned *out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    printf("%-20s", "cycles/inst/SIMD");
    for (auto &m : masks) printf(" %8s", m.name);
    printf("\n");
    for (int mixed = 0; mixed < 2; ++mixed)
    for (auto &en : entries) {
        printf("%-20s", mixed ? (std::string(en.name) + "/mix").c_str() : en.name);
        for (auto &m : masks) {
            hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, out, 10, m.mask, mixed);
            CHECK(hipDeviceSynchronize());
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, out, iters, m.mask, mixed);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %8.2f", (double)ms * 1e-3 * 2.4e9 / ((double)iters * 64.0 * w));
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
