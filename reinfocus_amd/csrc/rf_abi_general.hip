// rf_abi_general.hip -- rf_render_general: the general renderer (kernels in
// rf_general_kernels.h, rf_general_one.h)
#include "rf_host.h"

#include <algorithm>
#include <vector>

#include "rf_general_one.h"

using namespace rfh;

namespace {

// make_random_states(n, seed 0) for rf_render_general: the jump-ahead seeding the first time a size is
// asked for, a device-to-device copy of the remembered result afterwards (same bytes; sizes above
// 4 GiB of states are seeded every time rather than remembered)
int seed_zero_cached(rf_ctx *ctx, uint64_t n_states)
{
    constexpr uint64_t kMaxCachedStates = (4ull << 30) / sizeof(ulonglong2);
    if (ctx->d_seed_cache && ctx->seed_cache_n == n_states && ctx->n_states == n_states) {
        RF_HIP(hipSetDevice(ctx->device));
        drop_env_graph(ctx);
        RF_HIP(hipMemcpyAsync(ctx->d_states, ctx->d_seed_cache, n_states * sizeof(ulonglong2), hipMemcpyDeviceToDevice,
                              ctx->stream));
        return RF_OK;
    }
    int rc = rf_seed(ctx, n_states, 0, 0);
    if (rc != RF_OK || n_states > kMaxCachedStates)
        return rc;
    if (ctx->seed_cache_n != n_states) {
        RF_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->d_seed_cache)
            RF_HIP(hipFree(ctx->d_seed_cache));
        ctx->d_seed_cache = nullptr;
        ctx->seed_cache_n = 0;
        if (hipMalloc((void **)&ctx->d_seed_cache, n_states * sizeof(ulonglong2)) != hipSuccess) {
            (void)hipGetLastError(); // no room for the copy: keep seeding every time
            ctx->d_seed_cache = nullptr;
            return RF_OK;
        }
        ctx->seed_cache_n = n_states;
    }
    RF_HIP(hipMemcpyAsync(ctx->d_seed_cache, ctx->d_states, n_states * sizeof(ulonglong2), hipMemcpyDeviceToDevice,
                          ctx->stream));
    return RF_OK;
}

} // namespace

extern "C" {

unsigned rf_general_redo_pixels(rf_ctx *ctx) { return ctx ? ctx->general_redo_last : 0u; }

int rf_render_general(rf_ctx *ctx, int n, int h, int w, int spp, const double *cameras, const float *params,
                      const int32_t *types, const int32_t *sizes, int most, int width, uint8_t *host_out)
{
    RF_REQUIRE(ctx != nullptr && cameras && params && types && sizes, "rf_render_general: NULL argument");
    drop_env_graph(ctx);
    RF_REQUIRE(n > 0 && h > 0 && w > 0 && spp > 0 && most > 0 && width >= 7, "rf_render_general: bad sizes");
    RF_REQUIRE((uint64_t)h * (uint64_t)w < (1ull << 31), "rf_render_general: frame too large");
    for (int e = 0; e < n; ++e) {
        RF_REQUIRE(sizes[e] >= 0 && sizes[e] <= most, "rf_render_general: sizes[%d]=%d exceeds %d", e, sizes[e], most);
        for (int i = 0; i < sizes[e]; ++i)
            RF_REQUIRE(types[(size_t)e * most + i] == 0 || types[(size_t)e * most + i] == 1,
                       "rf_render_general: unknown shape type");
    }
    int rc = seed_zero_cached(ctx, (uint64_t)n * h * w); // render.py:115: fresh seed-0 states per call
    if (rc != RF_OK)
        return rc;
    rc = ensure_frames(ctx, n, h, w);
    if (rc != RF_OK)
        return rc;
    std::vector<rf::GeneralCamera> cams((size_t)n);
    for (int e = 0; e < n; ++e)
        cams[(size_t)e] = rf::general_camera(cameras + (size_t)e * 19);
    const size_t b_cam = (size_t)n * sizeof(rf::GeneralCamera), b_par = (size_t)n * most * width * sizeof(float),
                 b_typ = (size_t)n * most * sizeof(int32_t), b_siz = (size_t)n * sizeof(int32_t);
    // worlds of one rectangle per environment (and frames the quick pixel coordinates are proven for) take the cooperative
    // kernel of rf_general_one.h; everything else the literal one
    // worlds of exactly one shape per environment, the same kind in all of them: the cooperative kernel (rf_general_one.h)
    bool one_shape = ctx->general_one && h <= 4096 && w <= 4096 && width >= 7;
    const bool one_sphere = one_shape && n > 0 && types[0] == 0;
    // ... for launches that fill the device: the notebooks' one or two environments are a few hundred blocks, bound by the
    // latency of a sample, and there the literal kernel (one pixel per thread, no barriers) is up to three times faster
    // (1 x 300^2 x 100: 0.35 ms against 1.02); the cooperative kernel wins from about 2 M pixels per launch on with a
    // rectangle, 3 M with a sphere (profiles/r04_ab.txt section 19)
    if (one_shape && !ctx->general_one_always && (uint64_t)n * (uint64_t)h * (uint64_t)w <= (one_sphere ? 3000000u : 2000000u))
        one_shape = false;
    for (int e = 0; one_shape && e < n; ++e)
        one_shape = sizes[e] == 1 && types[(size_t)e * most] == (one_sphere ? 0 : 1);
    // environments per launch: the grid's y limit, and (cooperative kernel) pixel indices of the fix-up list in 32 bits
    const uint64_t hw64 = (uint64_t)h * (uint64_t)w;
    const int chunk = one_shape ? (int)std::min<uint64_t>(65535, 0xFFFFFFFFull / hw64) : 65535;
    const size_t b_redo = one_shape ? 256 + (size_t)std::min<uint64_t>((uint64_t)n, (uint64_t)chunk) * hw64 * sizeof(unsigned) : 0;
    const size_t o_par = (b_cam + 255) & ~(size_t)255, o_typ = o_par + ((b_par + 255) & ~(size_t)255),
                 o_siz = o_typ + ((b_typ + 255) & ~(size_t)255), o_redo = o_siz + ((b_siz + 255) & ~(size_t)255),
                 total = o_redo + b_redo;
    if (total > ctx->general_scratch_bytes) { // grown on demand, owned by the ctx
        RF_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->general_scratch)
            RF_HIP(hipFree(ctx->general_scratch));
        ctx->general_scratch = nullptr;
        ctx->general_scratch_bytes = 0;
        RF_HIP(hipMalloc(&ctx->general_scratch, total));
        ctx->general_scratch_bytes = total;
    }
    char *const scratch = (char *)ctx->general_scratch;
    hipError_t he = hipMemcpyAsync(scratch, cams.data(), b_cam, hipMemcpyHostToDevice, ctx->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(scratch + o_par, params, b_par, hipMemcpyHostToDevice, ctx->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(scratch + o_typ, types, b_typ, hipMemcpyHostToDevice, ctx->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(scratch + o_siz, sizes, b_siz, hipMemcpyHostToDevice, ctx->stream);
    if (he == hipSuccess) {
        rf::GeneralArgs a;
        a.frames = ctx->d_frames;
        a.states = ctx->d_states;
        a.cameras = (const rf::GeneralCamera *)scratch;
        a.params = (const float *)(scratch + o_par);
        a.types = (const int32_t *)(scratch + o_typ);
        a.sizes = (const int32_t *)(scratch + o_siz);
        a.n = n;
        a.h = h;
        a.w = w;
        a.spp = spp;
        a.hw = h * w;
        a.most = most;
        a.width = width;
        a.scale = (float)(255.0 / (double)spp);
        const int gx = (a.hw + rf::kBlock - 1) / rf::kBlock;
        const bool pow2 = is_pow2(h) && is_pow2(w);
        Timed timed(ctx, &ctx->ev_render);
        for (int e0 = 0; e0 < n && he == hipSuccess; e0 += chunk) {
            const int ne = (n - e0) < chunk ? (n - e0) : chunk;
            rf::GeneralArgs b = a;
            b.frames = a.frames + (size_t)e0 * a.hw * 3;
            b.states = a.states + (size_t)e0 * a.hw;
            b.cameras = a.cameras + (size_t)e0;
            b.params = a.params + (size_t)e0 * most * width;
            b.types = a.types + (size_t)e0 * most;
            b.sizes = a.sizes + e0;
            b.n = ne;
            if (one_shape) {
                rf::GeneralOneArgs d;
                d.g = b;
                d.redo_count = (unsigned *)(scratch + o_redo);
                d.redo_list = (unsigned *)(scratch + o_redo + 256);
                d.w64 = (double)w;
                d.h64 = (double)h;
                d.rw64 = 1.0 / (double)w;
                d.rh64 = 1.0 / (double)h;
                d.inv_w = 1.0f / (float)w;
                d.inv_h = 1.0f / (float)h;
                he = hipMemsetAsync(d.redo_count, 0, sizeof(unsigned), ctx->stream);
                if (he != hipSuccess)
                    break;
                // tiles of 128 x 6 or of 64 x 12, whichever leaves fewer dead columns
                const bool narrow = ((w + 63) / 64) * 64 < ((w + 127) / 128) * 128;
                const dim3 tiles(narrow ? (unsigned)(((w + 63) / 64) * ((h + 4 * rf::kSets - 1) / (4 * rf::kSets)))
                                        : (unsigned)(((w + 127) / 128) * ((h + 2 * rf::kSets - 1) / (2 * rf::kSets))), ne);
                const uint64_t blocks = ((uint64_t)ne * hw64 + rf::kBlock - 1) / rf::kBlock;
                const dim3 fix((unsigned)std::min<uint64_t>(blocks, 2048));
#define RF_LAUNCH_ONE(P, S, WXV)                                                                                          \
    do {                                                                                                               \
        hipLaunchKernelGGL((rf::render_general_one_kernel<P, S, WXV>), tiles, dim3(rf::kBlock2), 0, ctx->stream, d);   \
        hipLaunchKernelGGL(rf::render_general_fixup_kernel<P>, fix, dim3(rf::kBlock), 0, ctx->stream, d);             \
        ctx->render_kernel = "render_general_one_kernel<" #P ", " #S ", " #WXV ">";                                    \
    } while (0)
                if (one_sphere && pow2 && narrow) RF_LAUNCH_ONE(true, true, 2);
                else if (one_sphere && pow2) RF_LAUNCH_ONE(true, true, 4);
                else if (one_sphere && narrow) RF_LAUNCH_ONE(false, true, 2);
                else if (one_sphere) RF_LAUNCH_ONE(false, true, 4);
                else if (pow2 && narrow) RF_LAUNCH_ONE(true, false, 2);
                else if (pow2) RF_LAUNCH_ONE(true, false, 4);
                else if (narrow) RF_LAUNCH_ONE(false, false, 2);
                else RF_LAUNCH_ONE(false, false, 4);
#undef RF_LAUNCH_ONE
            } else if (pow2) {
                hipLaunchKernelGGL(rf::render_general_kernel<true>, dim3(gx, ne), dim3(rf::kBlock), 0, ctx->stream, b);
                ctx->render_kernel = "render_general_kernel<true>";
            } else {
                hipLaunchKernelGGL(rf::render_general_kernel<false>, dim3(gx, ne), dim3(rf::kBlock), 0, ctx->stream, b);
                ctx->render_kernel = "render_general_kernel<false>";
            }
            he = hipGetLastError();
        }
        ctx->general_redo_last = 0;
        if (he == hipSuccess && one_shape) // (diagnostics: rf_general_redo_pixels)
            he = hipMemcpyAsync(&ctx->general_redo_last, scratch + o_redo, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (he == hipSuccess)
        he = hipStreamSynchronize(ctx->stream);
    if (he != hipSuccess) {
        set_err("rf_render_general: %s", hipGetErrorString(he));
        return RF_ERR_HIP;
    }
    if (host_out)
        return rf_get_frames(ctx, 0, n, host_out);
    return RF_OK;
}

} // extern "C"
