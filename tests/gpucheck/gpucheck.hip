// gpucheck.hip -- TEST INFRASTRUCTURE: exhaustive on-device checks of the render kernel's
// custom correctly-rounded sqrt / reciprocal (reinfocus_amd/csrc/rf_math.h) against the
// compiler's IEEE expansions, for every float in the range the fast paths accept.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../reinfocus_amd/csrc/rf_math.h"
#include "probe_general.h"

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

// approx_pm1 (rf_math.h) builds its candidate from a SUBNORMAL float (the top 23 bits of the draw read
// as float bits) and one fma: correct only if the device neither flushes f32 subnormals nor rounds
// the fma.  Every one of the 2^32 high words against the same value computed in float64.
__global__ void check_approx_pm1_kernel(unsigned long long *bad)
{
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32);
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t hi = (uint32_t)i;
        const double want = (double)(hi >> 9) * (1.0 / 4194304.0) - 1.0; // k 2^-22 - 1: 24 bits, exact
        if (!((double)rf::approx_pm1(hi) == want))
            ++mine;
    }
    if (mine)
        atomicAdd(bad, mine);
}

extern "C" int gc_check_approx_pm1(unsigned long long *out)
{
    unsigned long long *d_bad;
    if (hipMalloc((void **)&d_bad, 8) != hipSuccess || hipMemset(d_bad, 0, 8) != hipSuccess)
        return -1;
    hipLaunchKernelGGL(check_approx_pm1_kernel, dim3(8192), dim3(256), 0, 0, d_bad);
    if (hipDeviceSynchronize() != hipSuccess)
        return -2;
    if (hipMemcpy(out, d_bad, 8, hipMemcpyDeviceToHost) != hipSuccess)
        return -3;
    (void)hipFree(d_bad);
    return 0;
}

// ---- probes of the general renderer's float64 library calls (see probe_general.h) ----------
__global__ void probe_f64_kernel(int op, const double *a, const double *b, double *out, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = probe::f64_op(op, a[i], b[i]);
}

__global__ void probe_uv_kernel(const float *normals, float *uv, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        probe::sphere_uv(normals + 3 * i, uv + 2 * i);
}

__global__ void probe_checker_kernel(const float *f, const float *u, int *sign, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        sign[i] = rf::checker_sign_general(f[i], u[i]);
}

__global__ void probe_sphere_red_kernel(const float *normals, const float *fu, const float *fv, int *red, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        red[i] = probe::sphere_red_fast(normals + 3 * i, fu[i], fv[i]);
}

namespace {
template <typename F>
int with_buffers(const void *const *host_in, const size_t *in_bytes, int n_in, void *host_out, size_t out_bytes, F launch)
{
    void *d_in[3] = {nullptr, nullptr, nullptr}, *d_out = nullptr;
    int rc = 0;
    for (int i = 0; i < n_in && rc == 0; ++i) {
        if (hipMalloc(&d_in[i], in_bytes[i]) != hipSuccess ||
            hipMemcpy(d_in[i], host_in[i], in_bytes[i], hipMemcpyHostToDevice) != hipSuccess)
            rc = -1;
    }
    if (rc == 0 && hipMalloc(&d_out, out_bytes) != hipSuccess)
        rc = -1;
    if (rc == 0) {
        launch(d_in, d_out);
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host_out, d_out, out_bytes, hipMemcpyDeviceToHost) != hipSuccess)
            rc = -2;
    }
    for (int i = 0; i < n_in; ++i)
        if (d_in[i])
            (void)hipFree(d_in[i]);
    if (d_out)
        (void)hipFree(d_out);
    return rc;
}
} // namespace

extern "C" int gc_probe_f64(int op, const double *a, const double *b, double *out, uint64_t n)
{
    const void *in[2] = {a, b};
    const size_t bytes[2] = {n * 8, n * 8};
    return with_buffers(in, bytes, 2, out, n * 8, [&](void **d, void *o) {
        hipLaunchKernelGGL(probe_f64_kernel, dim3(1024), dim3(256), 0, 0, op, (const double *)d[0], (const double *)d[1],
                           (double *)o, n);
    });
}

extern "C" int gc_probe_uv(const float *normals, float *uv, uint64_t n)
{
    const void *in[1] = {normals};
    const size_t bytes[1] = {n * 12};
    return with_buffers(in, bytes, 1, uv, n * 8, [&](void **d, void *o) {
        hipLaunchKernelGGL(probe_uv_kernel, dim3(1024), dim3(256), 0, 0, (const float *)d[0], (float *)o, n);
    });
}

extern "C" int gc_probe_checker(const float *f, const float *u, int *sign, uint64_t n)
{
    const void *in[2] = {f, u};
    const size_t bytes[2] = {n * 4, n * 4};
    return with_buffers(in, bytes, 2, sign, n * 4, [&](void **d, void *o) {
        hipLaunchKernelGGL(probe_checker_kernel, dim3(1024), dim3(256), 0, 0, (const float *)d[0], (const float *)d[1],
                           (int *)o, n);
    });
}

extern "C" int gc_probe_sphere_red(const float *normals, const float *fu, const float *fv, int *red, uint64_t n)
{
    const void *in[3] = {normals, fu, fv};
    const size_t bytes[3] = {n * 12, n * 4, n * 4};
    return with_buffers(in, bytes, 3, red, n * 4, [&](void **d, void *o) {
        hipLaunchKernelGGL(probe_sphere_red_kernel, dim3(1024), dim3(256), 0, 0, (const float *)d[0], (const float *)d[1],
                           (const float *)d[2], (int *)o, n);
    });
}
