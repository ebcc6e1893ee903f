// rf_abi_render.hip -- the fast path's launches: which render kernel a launch takes, the frame buffer, the focus
// measure (rf_render, rf_get_frames, rf_upload_frames, rf_focus, rf_step).
#include "rf_host.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "rf_coop2.h"
#include "rf_focus.h"
#include "rf_render.h"
#include "rf_wave.h"

using namespace rfh;

namespace {

// Tile layout of render_kernel_coop2 for a frame size (index into the table in launch_render):
//   0: 128 x 6 (4 waves of 32 x 2 side by side)   1: 64 x 12 (2 x 2 such waves)
//   2: 256 x 3 (4 waves of 64 x 1)                3: 128 x 6 (2 x 2 waves of 64 x 1)
//   4: 64 x 12 (4 waves of 16 x 4 side by side)
// Cost model fitted to tools/ablayout.sh (G samples/s at 128 / 256 / 300 / 384 / 512 / 600 px):
// time ~ padded area x shape factor x (1 + 0.35 x share of tile columns that lie entirely
// inside the target).  Such tiles have more stragglers than the 256-entry cooperative list holds
// and fall back to two in-wave sphere attempts; 64 x 1 waves are cut more often by the target's
// vertical edges (factor 1.045).  `hit_fraction` = width of the target / width of the frame.
int pick_tile_layout(int h, int w, double hit_fraction)
{
    static const int tile_w[4] = {128, 64, 256, 128}, tile_h[4] = {2 * rf::kSets, 4 * rf::kSets, rf::kSets, 2 * rf::kSets};
    static const double shape[4] = {1.0, 1.0, 1.045, 1.048};
    const double lo = 0.5 * (1.0 - hit_fraction) * w, hi = 0.5 * (1.0 + hit_fraction) * w;
    int best = 0;
    double best_cost = 0.0;
    for (int l = 0; l < 4; ++l) {
        const int cols = (w + tile_w[l] - 1) / tile_w[l], rows = (h + tile_h[l] - 1) / tile_h[l];
        int inside = 0;
        for (int c = 0; c < cols; ++c)
            inside += (c * tile_w[l] >= lo && (c + 1) * tile_w[l] <= hi) ? 1 : 0;
        const double cost = (double)cols * tile_w[l] * rows * tile_h[l] * shape[l] * (1.0 + 0.35 * inside / cols);
        if (l == 0 || cost < best_cost) {
            best = l;
            best_cost = cost;
        }
    }
    // 64 x 12 tiles of 16 x 4 pixel waves (layout 4) behave like layout 1 (within 2 % from 64 to 600
    // px) except around 128 px, where the narrower waves fit the target's edges better: 130.5
    // against 126 for layouts 0 / 1 / 3
    if (w > 64 && w <= 128 && (best == 0 || best == 1))
        return 4;
    return best;
}

// Which of the three organisations a launch of the canonical camera takes, by its pixels (all bit-identical; round 6,
// fused step, us per env step of in-wave / wave-cooperative / block-cooperative: profiles/r06_ab.txt section 3):
//   * few blocks -- the reference's own default is ONE environment of 300 x 300 pixels at 100 samples
//     (gym.make("DiscreteSteps-v0")) -- are bound by the latency of a sample: render_kernel<AXIS, POW2>, one pixel per
//     thread, rejection loops inside the wave, nothing to wait for (1 x 300^2 x 100: 253 us per step against 688 for
//     three pixels per thread with block-cooperative tails; 4: 501 / 575 / 832; 5: 570 / 642 / 910).
//   * launches that are about ONE round of resident waves -- the reference's training shape, 8 environments of 300^2 at
//     100 samples (examples/ppo_tuned.yml:5): 3752 waves on 1024 SIMDs -- are bound by vector issue at 3-4 waves per
//     SIMD, where nothing hides a block barrier: render_kernel_wave<.., 3> (rf_wave.h; 6 x 300^2 x 100: 703 / 661 / 954,
//     8: 876 / 757 / 1046, 16: 1519 / 1305 / 1317, 24: 2150 / 1753 / 1809; 16 x 256^2 x 16: 230 / 212 / 222;
//     128 x 128^2 x 4: 159 / 126 / 128; 4 x 512^2 x 64: 722 / 650 / 653).
//   * from about two rounds on, three pixels per thread with block-cooperative tails (render_kernel_coop2 and its strip
//     form; 48 x 256^2 x 16: 536 / 430 / 416, 256 x 128^2 x 4: 254 / 188 / 187).
constexpr uint64_t kFewPixels = 500000, kOneRoundPixels = 2200000;
bool few_blocks(uint64_t n, uint64_t h, uint64_t w) { return n * h * w <= kFewPixels; }
bool one_round(uint64_t n, uint64_t h, uint64_t w) { return n * h * w <= kOneRoundPixels; }

} // namespace

namespace rfh {

int ensure_frames(rf_ctx *ctx, int n, int h, int w)
{
    const size_t need = (size_t)n * h * w * 3 + 64; // + slack for dword tails
    if (need > ctx->frames_cap) {
        if (ctx->d_frames)
            RF_HIP(hipFree(ctx->d_frames));
        ctx->d_frames = nullptr;
        ctx->frames_cap = 0;
        RF_HIP(dev_malloc((void **)&ctx->d_frames, need));
        ctx->frames_cap = need;
    }
    ctx->fn = n;
    ctx->fh = h;
    ctx->fw = w;
    return RF_OK;
}

int ensure_frames2(rf_ctx *ctx, int n, int h, int w)
{
    const size_t need = (size_t)n * h * w * 3 + 64;
    if (need > ctx->frames2_cap) {
        if (ctx->d_frames2)
            RF_HIP(hipFree(ctx->d_frames2));
        ctx->d_frames2 = nullptr;
        ctx->frames2_cap = 0;
        RF_HIP(dev_malloc((void **)&ctx->d_frames2, need));
        ctx->frames2_cap = need;
    }
    return RF_OK;
}

int render_form(const rf_ctx *ctx, int n, int h, int w)
{
    if (!ctx->coop || ctx->one_px)
        return 0;
    if (ctx->wave_sets)
        return 20 + ctx->wave_sets;
    if (!ctx->auto_form)
        return 3;
    if (few_blocks((uint64_t)n, (uint64_t)h, (uint64_t)w))
        return 0;
    if (one_round((uint64_t)n, (uint64_t)h, (uint64_t)w))
        return 23;
    return 3;
}

// enqueues the render of n envs whose scene arrays are cam / rect (device pointers)
int launch_render(rf_ctx *ctx, int n, int h, int w, int spp, const float *cam, const float *rect, bool axis,
                  bool count_pixels, const SecondPass *second)
{
    int rc = ensure_frames(ctx, n, h, w);
    if (rc == RF_OK && second)
        rc = ensure_frames2(ctx, n, h, w);
    if (rc != RF_OK)
        return rc;
    const int form = render_form(ctx, n, h, w);
    RF_REQUIRE(!second || axis, "launch_render: no two-pass instance of the general-camera kernel");
    rf::RenderArgs a;
    a.frames = ctx->d_frames;
    a.states = ctx->d_states;
    a.cam_dyn = cam;
    a.rect = rect;
    a.cs = ctx->cs;
    a.tab = ctx->tab;
    a.n = n;
    a.h = h;
    a.w = w;
    a.spp = spp;
    a.hw = h * w;
    a.scale = (float)(255.0 / (double)spp);
    const bool pow2 = is_pow2(h) && is_pow2(w);
    a.fc = rf::frame_const(h, w);
    a.count2 = second ? second->count : nullptr;
    a.cam_dyn2 = second ? second->cam : nullptr;
    a.rect2 = second ? second->rect : nullptr;
    a.frames2 = second ? ctx->d_frames2 : nullptr;
    a.env0 = 0;
    a.main_tiles = 0;
    a.strip_x0 = 0;

    const int gx = (a.hw + rf::kBlock - 1) / rf::kBlock;
    {
        Timed timed(ctx, &ctx->ev_render);
        for (int e0 = 0; e0 < n; e0 += 65535) {
            const int ne = (n - e0) < 65535 ? (n - e0) : 65535;
            rf::RenderArgs b = a;
            b.frames = a.frames + (size_t)e0 * a.hw * 3;
            b.states = a.states + (size_t)e0 * a.hw;
            b.cam_dyn = a.cam_dyn + (size_t)e0 * 9;
            b.rect = a.rect + (size_t)e0 * 2;
            b.n = ne;
            if (second) {
                b.cam_dyn2 = a.cam_dyn2 + (size_t)e0 * 9;
                b.rect2 = a.rect2 + (size_t)e0 * 2;
                b.frames2 = a.frames2 + (size_t)e0 * a.hw * 3;
                b.env0 = e0;
            }
            const dim3 grid(gx, ne), block(rf::kBlock);
            // A block's tile is WX waves of WW x 64/WW pixels side by side, 4/WX down, kSets sets:
            //   A 128 x 6  (WX 4, WW 32)   B 64 x 12 (2, 32)   C 256 x 3 (4, 64)   D 128 x 6 (2, 64)
            // see pick_tile_layout.
            const int layout = pick_tile_layout(h, w, ctx->hit_fraction);
            static const int kLayoutWX[5] = {4, 2, 4, 2, 4};
            static const int kLayoutWW[5] = {32, 32, 64, 64, 16};
            const int layout_w = kLayoutWX[layout] * kLayoutWW[layout],
                      layout_h = (4 / kLayoutWX[layout]) * (64 / kLayoutWW[layout]) * rf::kSets;
            const dim3 tiles2(((w + layout_w - 1) / layout_w) * ((h + layout_h - 1) / layout_h), ne);
            const dim3 block2(rf::kBlock2);
            const bool lens32 = a.cs.lens_f32 != 0;
            // widths beyond 128 that are not a multiple of 64: the remainder (<= 48 columns) as a strip of 48 x 16 tiles
            // next to the main ones (render_kernel_coop2_strip) instead of a last tile column that is mostly dead lanes
            // (measured: +4.0 % at 300 px, +2.1 ... 2.4 % at 200 / 400 / 600 px; at 100 px the 16 x 4 pixel waves of
            // layout 4 are 12 % faster than a 64-column main part: profiles/r04_ab.txt section 17)
            const int rem = w % 64;
            if (axis && form == 3 && ctx->strip && !pow2 && w > 128 && rem > 0 &&
                rem <= 48) {
                b.strip_x0 = w - rem;
                const bool wide = b.strip_x0 % 128 == 0; // main tiles of 128 x 6 where they fit, else 64 x 12 (+0.8 % at 300 px)
                b.main_tiles = wide ? (b.strip_x0 / 128) * ((h + 2 * rf::kSets - 1) / (2 * rf::kSets))
                                    : (b.strip_x0 / 64) * ((h + 4 * rf::kSets - 1) / (4 * rf::kSets));
                const dim3 tiles_s((unsigned)(b.main_tiles + (h + 15) / 16), ne);
                // (one instance for one and for two passes -- the two-pass form, which reads its arguments afresh in each
                // tile shape's code, is also the one that compiles without spills: a single pass is a count of zero)
                if (!second)
                    b.count2 = ctx->d_zero;
                if (lens32 && wide) {
                    hipLaunchKernelGGL((rf::render_kernel_coop2_strip<1, 4>), tiles_s, block2, 0, ctx->stream, b);
                    ctx->render_kernel = "render_kernel_coop2_strip<1, 4>";
                } else if (lens32) {
                    hipLaunchKernelGGL((rf::render_kernel_coop2_strip<1, 2>), tiles_s, block2, 0, ctx->stream, b);
                    ctx->render_kernel = "render_kernel_coop2_strip<1, 2>";
                } else if (wide) {
                    hipLaunchKernelGGL((rf::render_kernel_coop2_strip<0, 4>), tiles_s, block2, 0, ctx->stream, b);
                    ctx->render_kernel = "render_kernel_coop2_strip<0, 4>";
                } else {
                    hipLaunchKernelGGL((rf::render_kernel_coop2_strip<0, 2>), tiles_s, block2, 0, ctx->stream, b);
                    ctx->render_kernel = "render_kernel_coop2_strip<0, 2>";
                }
            } else if (axis && form > 20) {
                const int sets = form - 20;
                const int chunks = (a.hw + 63) / 64; // sets of 64 consecutive pixels; a wave takes `sets` of them
                const dim3 waves((unsigned)((chunks + sets - 1) / sets), ne);
#define RF_LAUNCHW_ONE(P, L, K)                                                                            \
    if (second) {                                                                                          \
        hipLaunchKernelGGL((rf::render_kernel_wave<P, L, K, true>), waves, dim3(64), 0, ctx->stream, b);    \
        ctx->render_kernel = "render_kernel_wave<" #P ", " #L ", " #K ", true>";                           \
    } else {                                                                                               \
        hipLaunchKernelGGL((rf::render_kernel_wave<P, L, K, false>), waves, dim3(64), 0, ctx->stream, b);   \
        ctx->render_kernel = "render_kernel_wave<" #P ", " #L ", " #K ", false>";                          \
    }
#define RF_LAUNCHW(P, L)                                                                                   \
    do {                                                                                                   \
        switch (sets) {                                                                                    \
        case 1: RF_LAUNCHW_ONE(P, L, 1); break;                                                            \
        case 2: RF_LAUNCHW_ONE(P, L, 2); break;                                                            \
        default: RF_LAUNCHW_ONE(P, L, 3); break;                                                           \
        }                                                                                                  \
    } while (0)
                if (pow2 && lens32)
                    RF_LAUNCHW(true, 1);
                else if (pow2)
                    RF_LAUNCHW(true, 0);
                else if (lens32)
                    RF_LAUNCHW(false, 1);
                else
                    RF_LAUNCHW(false, 0);
#undef RF_LAUNCHW_ONE
#undef RF_LAUNCHW
            } else if (axis && form == 3) {
#define RF_LAUNCH2_ONE(P, L, WX, WW)                                                                       \
    if (second) {                                                                                          \
        hipLaunchKernelGGL((rf::render_kernel_coop2<P, L, WX, WW, true>), tiles2, block2, 0, ctx->stream, b); \
        ctx->render_kernel = "render_kernel_coop2<" #P ", " #L ", " #WX ", " #WW ", true>";                \
    } else {                                                                                               \
        hipLaunchKernelGGL((rf::render_kernel_coop2<P, L, WX, WW>), tiles2, block2, 0, ctx->stream, b);     \
        ctx->render_kernel = "render_kernel_coop2<" #P ", " #L ", " #WX ", " #WW ">";                      \
    }
#define RF_LAUNCH2(P, L)                                                                                   \
    do {                                                                                                   \
        switch (layout) {                                                                                  \
        case 0: RF_LAUNCH2_ONE(P, L, 4, 32); break;                                                        \
        case 1: RF_LAUNCH2_ONE(P, L, 2, 32); break;                                                        \
        case 2: RF_LAUNCH2_ONE(P, L, 4, 64); break;                                                        \
        case 3: RF_LAUNCH2_ONE(P, L, 2, 64); break;                                                        \
        default: RF_LAUNCH2_ONE(P, L, 4, 16); break;                                                       \
        }                                                                                                  \
    } while (0)
                if (pow2 && lens32)
                    RF_LAUNCH2(true, 1);
                else if (pow2)
                    RF_LAUNCH2(true, 0);
                else if (lens32)
                    RF_LAUNCH2(false, 1);
                else
                    RF_LAUNCH2(false, 0);
#undef RF_LAUNCH2_ONE
#undef RF_LAUNCH2
            }
            else if (axis && pow2 && second) {
                hipLaunchKernelGGL((rf::render_kernel<true, true, true>), grid, block, 0, ctx->stream, b);
                ctx->render_kernel = "render_kernel<true, true, true>";
            } else if (axis && second) {
                hipLaunchKernelGGL((rf::render_kernel<true, false, true>), grid, block, 0, ctx->stream, b);
                ctx->render_kernel = "render_kernel<true, false, true>";
            } else if (axis && pow2) {
                hipLaunchKernelGGL((rf::render_kernel<true, true>), grid, block, 0, ctx->stream, b);
                ctx->render_kernel = "render_kernel<true, true>";
            } else if (axis) {
                hipLaunchKernelGGL((rf::render_kernel<true, false>), grid, block, 0, ctx->stream, b);
                ctx->render_kernel = "render_kernel<true, false>";
            } else if (pow2) {
                hipLaunchKernelGGL((rf::render_kernel<false, true>), grid, block, 0, ctx->stream, b);
                ctx->render_kernel = "render_kernel<false, true>";
            } else {
                hipLaunchKernelGGL((rf::render_kernel<false, false>), grid, block, 0, ctx->stream, b);
                ctx->render_kernel = "render_kernel<false, false>";
            }
        }
    }
    RF_HIP(hipGetLastError());
    if (count_pixels) // (enqueue_env_step counts after the step: its launches may be replayed, and skip slots)
        rfh::count_pixels((unsigned long long)n * (unsigned long long)a.hw);
    if (ctx->ev_render.size() > 512)
        return drain_events(ctx->ev_render, ctx->render_ms, ctx->render_n);
    return RF_OK;
}

// the reduction buffers of the focus measure for n frames (grown on demand; the device-resident environment keeps a
// pointer to the sums: it follows, and a captured step that holds the old pointers is dropped)
int ensure_focus(rf_ctx *ctx, int n)
{
    if (n <= ctx->focus_cap)
        return RF_OK;
    drop_env_graph(ctx);
    RF_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->d_sums) RF_HIP(hipFree(ctx->d_sums));
    if (ctx->d_var) RF_HIP(hipFree(ctx->d_var));
    ctx->d_sums = nullptr;
    ctx->d_var = nullptr;
    ctx->env.sums = nullptr;
    ctx->focus_cap = 0;
    RF_HIP(dev_malloc((void **)&ctx->d_sums, (size_t)n * 2 * sizeof(unsigned long long)));
    RF_HIP(dev_malloc((void **)&ctx->d_var, (size_t)n * sizeof(double)));
    ctx->focus_cap = n;
    ctx->env.sums = ctx->d_sums;
    return RF_OK;
}

// enqueues the focus measure of the first n frames: sums into ctx->d_sums, variances into ctx->d_var (device).
// in_env_step: the sums were zeroed by the environment kernel before (env_pre_kernel / env_reset_kernel) and the
// variance is taken from them by the one after (env_post_kernel / env_reset_post_kernel, the same expression as
// focus_finalize): no memset and no finalize launch -- two nodes less per focus measure of a replayed step.
// fused_count != null: both measures of a fused environment step as one launch of 2 n rows (FocusArgs::count2)
int launch_focus(rf_ctx *ctx, int n, int h, int w, int gray_mode, const float *skip_rect, bool in_env_step, const int *fused_count)
{
    // Widths that are a multiple of 4 (>= 8): focus_kernel_roll, four pixels per lane and everything in registers; any
    // other width: the byte-per-thread kernel over column tiles.  Both take frames of any size (vision.py:11-39 scores
    // whatever it is handed).  REINFOCUS_FOCUS_KERNEL=quad / byte, REINFOCUS_FOCUS_BAND: rf_host.h.
    const int focus_choice = ctx->focus_choice, focus_band = ctx->focus_band;
    const size_t lds_quad = (((size_t)(2 * rf::kBandQ + 6) * w) + 15) & ~(size_t)15;
    const bool roll = (w & 3) == 0 && w >= 8 && focus_choice == 0;
    const bool quad = !roll && (w & 3) == 0 && w >= 4 && lds_quad <= 64 * 1024 && focus_choice == 1;
    const size_t lds_byte = ((size_t)(rf::kBand + 4) * (rf::kTileB + 4) + (size_t)(rf::kBand + 2) * (rf::kTileB + 2) + 15) & ~(size_t)15;
    int rc = ensure_focus(ctx, n);
    if (rc != RF_OK)
        return rc;
    if (!in_env_step)
        RF_HIP(hipMemsetAsync(ctx->d_sums, 0, (size_t)n * 2 * sizeof(unsigned long long), ctx->stream));
    {
        Timed timed(ctx, &ctx->ev_focus);
        const int rows = fused_count ? 2 * n : n;
        // focus_kernel_roll: rows per band.  A band of R rows reads R + 4 (the halo rows cost 4 / R), and a launch wants
        // at least a few waves per SIMD: the largest of 64 / 32 / 16 / 8 that leaves 8192 waves, else 8.
        const int groups = w >> 2;
        const bool halo = roll && (64 % groups) != 0; // (groups >= 2 there)
        int band = 8;
        if (roll) {
            for (int r = rf::kRollBandMax; r >= 8; r >>= 1) {
                if (r >= 2 * h && r > 8) // (a band taller than the frame only walks rows that do not exist)
                    continue;
                const uint64_t lanes = (uint64_t)((h + r - 1) / r) * groups;
                const uint64_t waves = (lanes + (halo ? 61 : 63)) / (halo ? 62 : 64) * (uint64_t)rows;
                if (waves >= 8192 || r == 8) {
                    band = r;
                    break;
                }
            }
            if (focus_band >= 1 && focus_band <= rf::kRollBandMax)
                band = focus_band;
        }
        for (int e0 = 0; e0 < rows; e0 += 65535) {
            const int ne = (rows - e0) < 65535 ? (rows - e0) : 65535;
            rf::FocusArgs a;
            a.frames = ctx->d_frames + (fused_count ? 0 : (size_t)e0 * h * w * 3);
            a.sums = ctx->d_sums + (fused_count ? 0 : (size_t)e0 * 2);
            a.n = ne;
            a.h = h;
            a.w = w;
            a.gray15 = gray_mode == RF_GRAY_15BIT;
            a.skip_rect = skip_rect ? skip_rect + (size_t)e0 * 2 : nullptr;
            a.count2 = fused_count;
            a.frames2 = ctx->d_frames2;
            a.sums2 = ctx->env.sums2;
            a.n_step = n;
            a.row0 = e0;
            if (roll) {
                rf::FocusRollArgs ra;
                ra.f = a;
                ra.dot = rf::gray_dot(a.gray15);
                ra.band = band;
                ra.groups = groups;
                ra.bands = (h + band - 1) / band;
                const uint64_t lanes = (uint64_t)ra.bands * groups;
                const uint64_t gx = (lanes + (halo ? 61 : 63)) / (halo ? 62 : 64);
                RF_REQUIRE(gx < (1ull << 31), "rf_focus: frame %d x %d too large", h, w);
                if (halo)
                    hipLaunchKernelGGL(rf::focus_kernel_roll<true>, dim3((unsigned)gx, ne), dim3(64), 0, ctx->stream, ra);
                else
                    hipLaunchKernelGGL(rf::focus_kernel_roll<false>, dim3((unsigned)gx, ne), dim3(64), 0, ctx->stream, ra);
            } else if (quad) {
                hipLaunchKernelGGL(rf::focus_kernel_quad, dim3((h + rf::kBandQ - 1) / rf::kBandQ, ne), dim3(rf::kBlock), lds_quad,
                                   ctx->stream, a);
            } else {
                const uint64_t gx = (uint64_t)((h + rf::kBand - 1) / rf::kBand) * (uint64_t)((w + rf::kTileB - 1) / rf::kTileB);
                RF_REQUIRE(gx < (1ull << 31), "rf_focus: frame %d x %d too large", h, w);
                hipLaunchKernelGGL(rf::focus_kernel, dim3((unsigned)gx, ne), dim3(rf::kBlock), lds_byte, ctx->stream, a);
            }
        }
        if (!in_env_step)
            hipLaunchKernelGGL(rf::focus_finalize, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_sums,
                               ctx->d_var, n, (unsigned long long)h * (unsigned long long)w);
    }
    RF_HIP(hipGetLastError());
    return RF_OK;
}

} // namespace rfh

extern "C" {

int rf_render(rf_ctx *ctx, int n, int h, int w, int spp, uint8_t *host_out)
{
    RF_REQUIRE(ctx != nullptr, "rf_render: ctx is NULL");
    RF_REQUIRE(ctx->scene_n > 0, "rf_render: no scene uploaded (rf_set_scene first)");
    RF_REQUIRE(n == ctx->scene_n, "rf_render: n=%d but the scene holds %d environments", n, ctx->scene_n);
    RF_REQUIRE(h > 0 && w > 0 && spp > 0, "rf_render: h, w, spp must be positive");
    RF_REQUIRE((uint64_t)h * (uint64_t)w < (1ull << 31), "rf_render: frame too large");
    const uint64_t need = (uint64_t)n * h * w;
    RF_REQUIRE(need <= ctx->n_states, "rf_render: %llu pixels but only %llu RNG states (rf_seed first)",
               (unsigned long long)need, (unsigned long long)ctx->n_states);
    RF_HIP(hipSetDevice(ctx->device));
    drop_env_graph(ctx);
    int rc = launch_render(ctx, n, h, w, spp, ctx->d_cam, ctx->d_rect, ctx->axis);
    if (rc != RF_OK)
        return rc;
    if (host_out)
        return rf_get_frames(ctx, 0, n, host_out);
    return RF_OK;
}

int rf_get_frames(rf_ctx *ctx, int first_env, int n_envs, uint8_t *host_out)
{
    RF_REQUIRE(ctx != nullptr && host_out != nullptr, "rf_get_frames: NULL argument");
    RF_REQUIRE(ctx->fn > 0, "rf_get_frames: no frames rendered yet");
    RF_REQUIRE(first_env >= 0 && n_envs >= 0 && first_env + n_envs <= ctx->fn,
               "rf_get_frames: env range [%d,%d) exceeds %d frames", first_env, first_env + n_envs, ctx->fn);
    RF_HIP(hipSetDevice(ctx->device));
    const size_t per = (size_t)ctx->fh * ctx->fw * 3;
    RF_HIP(hipMemcpyAsync(host_out, ctx->d_frames + per * first_env, per * n_envs, hipMemcpyDeviceToHost,
                          ctx->stream));
    RF_HIP(hipStreamSynchronize(ctx->stream));
    return RF_OK;
}

int rf_upload_frames(rf_ctx *ctx, int n, int h, int w, const uint8_t *host_in)
{
    RF_REQUIRE(ctx != nullptr && host_in != nullptr, "rf_upload_frames: NULL argument");
    RF_REQUIRE(n > 0 && h > 0 && w > 0, "rf_upload_frames: n, h, w must be positive");
    RF_HIP(hipSetDevice(ctx->device));
    drop_env_graph(ctx);
    RF_HIP(hipStreamSynchronize(ctx->stream));
    int rc = ensure_frames(ctx, n, h, w);
    if (rc != RF_OK)
        return rc;
    RF_HIP(hipMemcpyAsync(ctx->d_frames, host_in, (size_t)n * h * w * 3, hipMemcpyHostToDevice, ctx->stream));
    RF_HIP(hipStreamSynchronize(ctx->stream));
    return RF_OK;
}

int rf_focus(rf_ctx *ctx, int n, int h, int w, int gray_mode, double *host_var)
{
    RF_REQUIRE(ctx != nullptr && host_var != nullptr, "rf_focus: NULL argument");
    RF_REQUIRE(n > 0 && n <= ctx->fn && h == ctx->fh && w == ctx->fw,
               "rf_focus: asked for %dx%dx%d but the frame buffer holds %dx%dx%d", n, h, w, ctx->fn, ctx->fh,
               ctx->fw);
    RF_REQUIRE(gray_mode == RF_GRAY_15BIT || gray_mode == RF_GRAY_14BIT, "rf_focus: gray_mode must be 14 or 15");
    RF_HIP(hipSetDevice(ctx->device));
    drop_env_graph(ctx);
    int rc = launch_focus(ctx, n, h, w, gray_mode);
    if (rc != RF_OK)
        return rc;
    RF_HIP(hipMemcpyAsync(host_var, ctx->d_var, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    RF_HIP(hipStreamSynchronize(ctx->stream));
    return RF_OK;
}

int rf_step(rf_ctx *ctx, int n, int h, int w, int spp, int gray_mode, double *host_var)
{
    int rc = rf_render(ctx, n, h, w, spp, nullptr);
    if (rc != RF_OK)
        return rc;
    return rf_focus(ctx, n, h, w, gray_mode, host_var);
}

const char *rf_render_kernel_name(rf_ctx *ctx) { return ctx ? ctx->render_kernel : "none"; }


} // extern "C"
