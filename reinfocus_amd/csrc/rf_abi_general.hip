// rf_abi_general.hip -- rf_render_general: the general renderer (kernels in
// rf_general_kernels.h, rf_general_one.h)
#include "rf_host.h"

#include <stdlib.h>

#include <algorithm>
#include <set>
#include <vector>

#include "rf_general_chunk.h"
#include "rf_general_one.h"

using namespace rfh;

namespace {

// make_random_states(n, seed 0) for rf_render_general: the jump-ahead seeding the first time a size is asked for, with a
// copy of the result kept; afterwards the kernels read their pixels' states straight from that copy (GeneralArgs::states_in)
// and write the advanced ones to the context's array -- no copy per call (round 5 copied 32 bytes per pixel in front of
// every render: 0.1 of 3.47 ms at 256 x 256^2 x 16).  Sizes above 4 GiB of states are seeded every time rather than remembered.
// *states_in: where this call's start states are.
int seed_zero_cached(rf_ctx *ctx, uint64_t n_states, const ulonglong2 **states_in)
{
    constexpr uint64_t kMaxCachedStates = (4ull << 30) / sizeof(ulonglong2);
    if (ctx->d_seed_cache && ctx->seed_cache_n == n_states && ctx->n_states == n_states) {
        RF_HIP(hipSetDevice(ctx->device));
        drop_env_graph(ctx);
        *states_in = ctx->d_seed_cache;
        return RF_OK;
    }
    int rc = rf_seed(ctx, n_states, 0, 0);
    *states_in = ctx->d_states;
    if (rc != RF_OK || n_states > kMaxCachedStates)
        return rc;
    if (ctx->seed_cache_n != n_states) {
        RF_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->d_seed_cache)
            RF_HIP(hipFree(ctx->d_seed_cache));
        ctx->d_seed_cache = nullptr;
        ctx->seed_cache_n = 0;
        if (dev_malloc((void **)&ctx->d_seed_cache, n_states * sizeof(ulonglong2)) != hipSuccess) {
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
    RF_HIP(hipSetDevice(ctx->device));
    drop_env_graph(ctx);
    RF_REQUIRE(n > 0 && h > 0 && w > 0 && spp > 0 && most > 0 && width >= 7, "rf_render_general: bad sizes");
    RF_REQUIRE((uint64_t)h * (uint64_t)w < (1ull << 31), "rf_render_general: frame too large");
    for (int e = 0; e < n; ++e) {
        RF_REQUIRE(sizes[e] >= 0 && sizes[e] <= most, "rf_render_general: sizes[%d]=%d exceeds %d", e, sizes[e], most);
        for (int i = 0; i < sizes[e]; ++i)
            RF_REQUIRE(types[(size_t)e * most + i] == 0 || types[(size_t)e * most + i] == 1,
                       "rf_render_general: unknown shape type");
    }
    const uint64_t hw64 = (uint64_t)h * (uint64_t)w, pixels = (uint64_t)n * hw64;
    const ulonglong2 *states_in = nullptr;
    int rc = seed_zero_cached(ctx, pixels, &states_in); // render.py:115: fresh seed-0 states per call
    if (rc != RF_OK)
        return rc;
    rc = ensure_frames(ctx, n, h, w);
    if (rc != RF_OK)
        return rc;
    std::vector<rf::GeneralCamera> cams((size_t)n);
    for (int e = 0; e < n; ++e)
        cams[(size_t)e] = rf::general_camera(cameras + (size_t)e * 19);

    // Which kernel (all bit-identical):
    //  * kOne: worlds of exactly one shape per environment, the same kind in all of them, in launches that fill the device
    //    (more than 2 M pixels with a rectangle, 3 M with a sphere): the cooperative kernel of rf_general_one.h.  The
    //    notebooks' one or two environments are a few hundred blocks, bound by the latency of a sample, where a kernel
    //    without barriers is up to three times faster (profiles/r04_ab.txt section 19).
    //  * kDense: at most three shapes per environment, frames the quick pixel coordinates are proven for, launches of more
    //    than 2 M pixels: the float32 kernel with abstentions (rf_general_dense.h) -- its SIMPLE instances when every camera
    //    has canonical axes and a lens radius whose float32 offset is exact, the ones with the reference's float64 lens
    //    products otherwise.  Its fix-up kernel renders the pixels that abstained one thread each, which takes as long as
    //    the literal kernel takes for a launch that does not fill the device: below 2 M pixels the literal kernel is up to
    //    twice as fast (1 x 300 x 600 x 100: 0.52 against 1.03 ms; 32 x 256^2 x 16: 0.63 against 0.61: profiles/r05_ab.txt
    //    section 8).
    //  * kLiteral: everything else (more shapes, frames beyond 4096 pixels, small launches).
    enum { kLiteral, kOne, kDense } kind = kLiteral;
    const bool quick_frame = h <= 4096 && w <= 4096;
    bool uniform_count = true;
    for (int e = 0; uniform_count && e < n; ++e)
        uniform_count = sizes[e] == most;
    const bool one_sphere = types[0] == 0;
    bool simple_cameras = false;
    bool one_shape = ctx->general_one && quick_frame && uniform_count && most == 1 &&
                     (ctx->general_one_always || pixels > (one_sphere ? 3000000u : 2000000u));
    for (int e = 0; one_shape && e < n; ++e)
        one_shape = types[(size_t)e] == (one_sphere ? 0 : 1);
    // every camera: canonical axes (cheap, first) and a lens radius whose float32 offset is KNOWN to be exact -- the
    // reference's aperture, or a radius that has come back often enough to have been worth its 60 ms proof
    // (rf_abi_ctx.hip lens_exact_if_known); a launch with more than a handful of different apertures takes the instances
    // with the reference's float64 lens products right away
    auto cameras_simple = [&]() {
        std::set<double> radii;
        for (int e = 0; e < n; ++e) {
            const double radius = cams[(size_t)e].lens_radius;
            if (!rf::camera_axes_simple(cams[(size_t)e]) || !(radius == radius))
                return false;
            radii.insert(radius);
            if (radii.size() > 4)
                return false;
        }
        bool all = true;
        for (const double radius : radii) // (every radius is asked about: each counts its calls)
            all = lens_exact_if_known(radius) && all;
        return all;
    };
    if (one_shape) {
        kind = kOne;
        simple_cameras = cameras_simple();
    } else if (ctx->general_dense && quick_frame && most <= 3 && (ctx->general_dense_always || pixels > 2000000u)) {
        kind = kDense;
        simple_cameras = cameras_simple();
    }
    const bool listed = kind != kLiteral;

    // environments per launch: the grid's y limit, and (kernels with a fix-up list: one-dimensional grids) pixel indices and
    // launched threads in 32 bits (rf_general_chunk.h)
    bool tiled = false, narrow = false;
    const uint64_t per_env = kind == kDense ? rf::dense_blocks_per_env(h, w, &tiled)
                             : kind == kOne ? rf::one_blocks_per_env(h, w, rf::kSets, &narrow) : 0;
    const int chunk = listed ? rf::general_listed_chunk(hw64, per_env) : 65535;
    const int n_chunks = (n + chunk - 1) / chunk;
    // the fix-up list: about one pixel in 10^3 abstains (one in 10^2 at 100 samples); a launch that abstains more often
    // than the list holds is rendered again by the literal kernel (below)
    const uint64_t chunk_pixels = std::min<uint64_t>((uint64_t)n, (uint64_t)chunk) * hw64;
    uint64_t cap = std::min<uint64_t>(chunk_pixels, std::max<uint64_t>(65536, chunk_pixels / 16));
    if (const char *v = getenv("REINFOCUS_GENERAL_REDO_CAP")) { // (tests: the overflow path at small sizes)
        char *end = nullptr;
        const long forced = strtol(v, &end, 10);
        if (end != v && forced >= 1)
            cap = std::min<uint64_t>(chunk_pixels, (uint64_t)forced);
    }
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_cam = (size_t)n * sizeof(rf::GeneralCamera), b_par = (size_t)n * most * width * sizeof(float),
                 b_typ = (size_t)n * most * sizeof(int32_t), b_siz = (size_t)n * sizeof(int32_t),
                 b_shp = kind == kDense ? (size_t)n * most * sizeof(rf::ShapeConst) : 0,
                 b_cnt = listed ? (size_t)n_chunks * sizeof(unsigned) : 0, b_lst = listed ? (size_t)cap * sizeof(unsigned) : 0;
    const size_t o_par = pad(b_cam), o_typ = o_par + pad(b_par), o_siz = o_typ + pad(b_typ), o_shp = o_siz + pad(b_siz),
                 o_cnt = o_shp + pad(b_shp), o_lst = o_cnt + pad(b_cnt), total = o_lst + pad(b_lst);
    if (total > ctx->general_scratch_bytes) { // grown on demand, owned by the ctx
        RF_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->general_scratch)
            RF_HIP(hipFree(ctx->general_scratch));
        ctx->general_scratch = nullptr;
        ctx->general_scratch_bytes = 0;
        RF_HIP(dev_malloc(&ctx->general_scratch, total));
        ctx->general_scratch_bytes = total;
    }
    char *const scratch = (char *)ctx->general_scratch;
    std::vector<rf::ShapeConst> shapes;
    if (kind == kDense) {
        shapes.resize((size_t)n * most);
        for (size_t i = 0; i < shapes.size(); ++i) // (rows beyond an environment's count are never read by the kernel)
            shapes[i] = rf::shape_const(params + i * width, width, (int)(i % (size_t)most) < sizes[i / (size_t)most] ? types[i] : 1);
    }
    RF_HIP(hipMemcpyAsync(scratch, cams.data(), b_cam, hipMemcpyHostToDevice, ctx->stream));
    RF_HIP(hipMemcpyAsync(scratch + o_par, params, b_par, hipMemcpyHostToDevice, ctx->stream));
    RF_HIP(hipMemcpyAsync(scratch + o_typ, types, b_typ, hipMemcpyHostToDevice, ctx->stream));
    RF_HIP(hipMemcpyAsync(scratch + o_siz, sizes, b_siz, hipMemcpyHostToDevice, ctx->stream));
    if (b_shp)
        RF_HIP(hipMemcpyAsync(scratch + o_shp, shapes.data(), b_shp, hipMemcpyHostToDevice, ctx->stream));
    if (b_cnt)
        RF_HIP(hipMemsetAsync(scratch + o_cnt, 0, b_cnt, ctx->stream));

    rf::GeneralArgs a;
    a.frames = ctx->d_frames;
    a.states = ctx->d_states;
    a.states_in = states_in;
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
    auto chunk_args = [&](int e0, int ne) {
        rf::GeneralArgs b = a;
        b.frames = a.frames + (size_t)e0 * a.hw * 3;
        b.states = a.states + (size_t)e0 * a.hw;
        b.states_in = a.states_in + (size_t)e0 * a.hw;
        b.cameras = a.cameras + (size_t)e0;
        b.params = a.params + (size_t)e0 * most * width;
        b.types = a.types + (size_t)e0 * most;
        b.sizes = a.sizes + e0;
        b.n = ne;
        return b;
    };
    auto launch_literal = [&](const rf::GeneralArgs &b) {
        if (pow2)
            hipLaunchKernelGGL(rf::render_general_kernel<true>, dim3(gx, b.n), dim3(rf::kBlock), 0, ctx->stream, b);
        else
            hipLaunchKernelGGL(rf::render_general_kernel<false>, dim3(gx, b.n), dim3(rf::kBlock), 0, ctx->stream, b);
    };
    {
        Timed timed(ctx, &ctx->ev_render);
        for (int c = 0; c < n_chunks; ++c) {
            const int e0 = c * chunk, ne = std::min(chunk, n - e0);
            const rf::GeneralArgs b = chunk_args(e0, ne);
            if (!listed) {
                launch_literal(b);
                ctx->render_kernel = pow2 ? "render_general_kernel<true>" : "render_general_kernel<false>";
                continue;
            }
            rf::GeneralOneArgs d;
            d.g = b;
            d.redo_count = (unsigned *)(scratch + o_cnt) + c;
            d.redo_list = (unsigned *)(scratch + o_lst);
            d.redo_cap = (unsigned)cap;
            d.shapes = (const rf::ShapeConst *)(scratch + o_shp) + (size_t)e0 * most;
            d.simple_cameras = simple_cameras ? 1 : 0;
            d.fc = rf::frame_const(h, w);
            // (the fix-up kernel: grid-stride over the list; a launch of few pixels needs fewer blocks than the full grid)
            const uint64_t blocks = ((uint64_t)ne * hw64 + 3) / 4;
            const dim3 fix((unsigned)std::min<uint64_t>(blocks, rf::kFixupBlocks));
            if (kind == kDense) {
                // 16 x 16 tiles (waves of 8 x 8 pixels) unless they pad the frame much more than 256-pixel runs do (frames
                // narrower or lower than a tile)
                RF_REQUIRE(per_env * (uint64_t)ne * 256 <= 0xFFFFFFFFull, "rf_render_general: too many threads for one launch");
                const dim3 grid_t((unsigned)(per_env * (uint64_t)ne)); // (the environment is the fastest index)
#define RF_LAUNCH_DENSE_TS(P, NS, T, S)                                                                                   \
    do {                                                                                                               \
        hipLaunchKernelGGL((rf::render_general_dense_kernel<P, NS, T, S>), grid_t, dim3(rf::kBlock), 0, ctx->stream, d); \
        hipLaunchKernelGGL(rf::render_general_fixup_kernel<P>, fix, dim3(rf::kBlock), 0, ctx->stream, d);             \
        ctx->render_kernel = "render_general_dense_kernel<" #P ", " #NS ", " #T ", " #S ">";                           \
    } while (0)
#define RF_LAUNCH_DENSE(P, NS)                                                                                          \
    do {                                                                                                               \
        if (tiled && simple_cameras) RF_LAUNCH_DENSE_TS(P, NS, true, true);                                            \
        else if (tiled) RF_LAUNCH_DENSE_TS(P, NS, true, false);                                                        \
        else if (simple_cameras) RF_LAUNCH_DENSE_TS(P, NS, false, true);                                               \
        else RF_LAUNCH_DENSE_TS(P, NS, false, false);                                                                  \
    } while (0)
                if (pow2 && most == 1) RF_LAUNCH_DENSE(true, 1);
                else if (pow2 && most == 2) RF_LAUNCH_DENSE(true, 2);
                else if (pow2) RF_LAUNCH_DENSE(true, 3);
                else if (most == 1) RF_LAUNCH_DENSE(false, 1);
                else if (most == 2) RF_LAUNCH_DENSE(false, 2);
                else RF_LAUNCH_DENSE(false, 3);
#undef RF_LAUNCH_DENSE
#undef RF_LAUNCH_DENSE_TS
                continue;
            }
            // tiles of 128 x 6 or of 64 x 12, whichever leaves fewer dead columns
            RF_REQUIRE(per_env * (uint64_t)ne * 256 <= 0xFFFFFFFFull, "rf_render_general: too many threads for one launch");
            const dim3 tiles((unsigned)(per_env * (uint64_t)ne)); // (one-dimensional, the environment fastest)
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
        }
        RF_HIP(hipGetLastError());
    } // (the events close here: around the kernels, not around the read-back of the counts and its synchronisation)
    {
        ctx->general_redo_last = 0;
        if (listed) {
            // how many pixels abstained, per launch; a launch with more of them than the list holds is rendered again, whole,
            // by the literal kernel from the call's fresh states (every call starts from seed-0 states: render.py:115)
            std::vector<unsigned> counts((size_t)n_chunks);
            RF_HIP(hipMemcpyAsync(counts.data(), scratch + o_cnt, b_cnt, hipMemcpyDeviceToHost, ctx->stream));
            RF_HIP(hipStreamSynchronize(ctx->stream));
            unsigned long long redo = 0;
            for (int c = 0; c < n_chunks; ++c) {
                redo += counts[(size_t)c];
                if (counts[(size_t)c] <= cap)
                    continue;
                const int e0 = c * chunk, ne = std::min(chunk, n - e0);
                const uint64_t first = (uint64_t)e0 * hw64, count = (uint64_t)ne * hw64;
                rc = seed_range(ctx, first, count, 0, first); // (the jump-ahead kernel, not the remembered copy: one path, rarely taken)
                if (rc != RF_OK)
                    return rc;
                rf::GeneralArgs again = chunk_args(e0, ne);
                again.states_in = again.states; // (freshly seeded in place)
                launch_literal(again);
                RF_HIP(hipGetLastError());
            }
            ctx->general_redo_last = (unsigned)std::min<unsigned long long>(redo, 0xFFFFFFFFull);
        }
    }
    RF_HIP(hipStreamSynchronize(ctx->stream));
    if (host_out)
        return rf_get_frames(ctx, 0, n, host_out);
    return RF_OK;
}

} // extern "C"
