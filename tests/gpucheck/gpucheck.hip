// gpucheck.hip -- TEST INFRASTRUCTURE: exhaustive on-device checks of the render kernel's
// custom correctly-rounded sqrt / reciprocal (reinfocus_amd/csrc/rf_math.h) against the
// compiler's IEEE expansions, for every float in the range the fast paths accept.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../reinfocus_amd/csrc/rf_math.h"

__global__ void check_range(uint32_t first_bits, uint32_t count, unsigned long long *bad /*[2]*/)
{
    unsigned long long bad_sqrt = 0, bad_rcp = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t b = first_bits + (uint32_t)i;
        float x;
        __builtin_memcpy(&x, &b, 4);
        if (!rf::in_fast_range(x))
            continue;
        if (!(rf::sqrt_rn_fast(x) == __builtin_sqrtf(x)))
            ++bad_sqrt;
        // the reciprocal is applied to len = sqrt(sq): x in [2^-50, 2^50]
        if (x >= 8.8817841970012523e-16f && x <= 1125899906842624.0f && !(rf::rcp_rn_fast(x) == 1.0f / x))
            ++bad_rcp;
    }
    if (bad_sqrt)
        atomicAdd(&bad[0], bad_sqrt);
    if (bad_rcp)
        atomicAdd(&bad[1], bad_rcp);
}

extern "C" int gc_check_sqrt_rcp(unsigned long long out[3])
{
    unsigned long long *d_bad;
    if (hipMalloc((void **)&d_bad, 16) != hipSuccess)
        return -1;
    if (hipMemset(d_bad, 0, 16) != hipSuccess)
        return -1;
    // every positive float from 2^-101 to 2^101 (covers the accepted range and its edges)
    const uint32_t lo = 0x0D000000u, hi = 0x72000000u;
    unsigned long long total = 0;
    for (uint64_t first = lo; first < hi; first += (1u << 28)) {
        uint32_t count = (uint32_t)((hi - first) < (1u << 28) ? (hi - first) : (1u << 28));
        hipLaunchKernelGGL(check_range, dim3(4096), dim3(256), 0, 0, (uint32_t)first, count, d_bad);
        total += count;
    }
    if (hipDeviceSynchronize() != hipSuccess)
        return -2;
    if (hipMemcpy(out, d_bad, 16, hipMemcpyDeviceToHost) != hipSuccess)
        return -3;
    out[2] = total;
    (void)hipFree(d_bad);
    return 0;
}
