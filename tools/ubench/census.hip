// census.hip -- how does the dispatcher spread the workgroups of a launch that does NOT fill the chip?  (development tool)
// Every wave records (XCC, SE, CU, SIMD) and spins long enough for the whole grid to be resident together; the host prints
// the histogram of waves per SIMD and per CU.  usage: census <workgroups> <threads per workgroup> [lds bytes]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void probe(unsigned *out, int spin)
{
    extern __shared__ int sink[];
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20); // HW_REG_XCC_ID
    unsigned v = threadIdx.x;
    for (int i = 0; i < spin; ++i)
        v = v * 1664525u + 1013904223u;
    if (v == 0xdeadbeef)
        sink[0] = 1;
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = hw;
        out[2 * w + 1] = xcc;
    }
}

int main(int argc, char **argv)
{
    const int groups = argc > 1 ? atoi(argv[1]) : 704, threads = argc > 2 ? atoi(argv[2]) : 64, lds = argc > 3 ? atoi(argv[3]) : 4096;
    const int waves = groups * (threads / 64);
    unsigned *d;
    CHECK(hipMalloc(&d, (size_t)waves * 2 * 4));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(groups), dim3(threads), lds, 0, d, 200000);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned> h((size_t)waves * 2);
    CHECK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    std::map<unsigned, int> per_simd, per_cu, per_xcc;
    for (int w = 0; w < waves; ++w) {
        const unsigned hw = h[2 * (size_t)w], xcc = h[2 * (size_t)w + 1] & 15;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned key = xcc << 16 | se << 8 | sh << 4 | cu;
        per_cu[key]++;
        per_simd[key << 2 | simd]++;
        per_xcc[xcc]++;
    }
    int hs[64] = {}, hc[64] = {};
    for (auto &kv : per_simd) hs[kv.second < 63 ? kv.second : 63]++;
    for (auto &kv : per_cu) hc[kv.second < 63 ? kv.second : 63]++;
    printf("%d workgroups of %d threads, %d B LDS: %d waves on %zu CUs / %zu SIMDs\n", groups, threads, lds, waves, per_cu.size(), per_simd.size());
    printf("  waves per SIMD (SIMDs with that many; %zu SIMDs got none):", 1024 - per_simd.size());
    for (int i = 1; i < 64; ++i) if (hs[i]) printf("  %d: %d", i, hs[i]);
    printf("\n  waves per CU (CUs with that many; %zu CUs got none):", 256 - per_cu.size());
    for (int i = 1; i < 64; ++i) if (hc[i]) printf("  %d: %d", i, hc[i]);
    printf("\n  per XCC:");
    for (auto &kv : per_xcc) printf(" %d", kv.second);
    printf("\n");
    return 0;
}
