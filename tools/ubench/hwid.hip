// hwid.hip -- where do the waves of a 256-thread block run?  (development tool, not product)
// Every wave records HW_ID (SIMD, CU, SE ...) and XCC_ID; the host prints, per wave index within
// the block, the histogram of SIMD ids, and how many distinct CUs / XCCs the blocks of a grid use.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <set>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void probe(unsigned *out, int spin)
{
    __shared__ int sink;
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20); // HW_REG_XCC_ID
    // keep the block resident for a while so that the grid really fills the chip
    unsigned v = threadIdx.x;
    for (int i = 0; i < spin; ++i)
        v = v * 1664525u + 1013904223u;
    if (v == 0xdeadbeef)
        sink = 1;
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = hw;
        out[2 * w + 1] = xcc;
    }
}

int main()
{
    const int blocks = 256 * 7 * 2;
    unsigned *d;
    CHECK(hipMalloc(&d, (size_t)blocks * 4 * 2 * 4));
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, d, 20000);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> h((size_t)blocks * 8);
    CHECK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    int hist[4][4] = {};
    std::set<unsigned> cus;
    std::map<unsigned, int> per_cu_first;
    int same_cu_blocks = 0;
    for (int b = 0; b < blocks; ++b) {
        unsigned cu_key0 = 0;
        for (int w = 0; w < 4; ++w) {
            const unsigned hw = h[(size_t)(b * 4 + w) * 2], xcc = h[(size_t)(b * 4 + w) * 2 + 1];
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            hist[w][simd]++;
            const unsigned key = (xcc & 15) << 16 | se << 8 | sh << 4 | cu;
            cus.insert(key);
            if (w == 0)
                cu_key0 = key;
            else if (key != cu_key0)
                same_cu_blocks++;
        }
        if (b < 24) {
            printf("block %3d:", b);
            for (int w = 0; w < 4; ++w) {
                const unsigned hw = h[(size_t)(b * 4 + w) * 2], xcc = h[(size_t)(b * 4 + w) * 2 + 1];
                printf("  w%d xcc%u se%u cu%2u simd%u wave%2u", w, xcc & 15, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15);
            }
            printf("\n");
        }
    }
    printf("wave index in block -> SIMD id histogram (%d blocks)\n", blocks);
    for (int w = 0; w < 4; ++w)
        printf("  wave %d: simd0 %5d simd1 %5d simd2 %5d simd3 %5d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("distinct (xcc, se, sh, cu): %zu; waves of a block on a different CU than wave 0: %d\n", cus.size(), same_cu_blocks);
    return 0;
}
