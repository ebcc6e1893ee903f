// rf_abi_env.hip -- the device-resident environment step (kernels in rf_env.h) and its
// schedules: the fused step (one render launch, one focus launch), the separate launches, the two-phase steps of a
// sharded environment, the hipGraph replay of small configurations.
#include "rf_host.h"

#include <math.h>
#include <string.h>

#include "rf_env.h"

using namespace rfh;

namespace {

// The per-step traffic of the device-resident environment as ONE block on either side -- inputs first, then outputs --
// so that a step enqueued in one go moves it with one copy in and one copy out (a hipGraph node each, instead of two
// and four): [pool f32 n x 2 | actions i32 n | pad to 16] [rewards f64 n | observations f32 n x 4 | count i32 | truncated u8 n]
struct EnvIo {
    size_t o_pool, o_actions, in_bytes, o_rewards, o_obs, o_count, o_truncated, bytes;
    explicit EnvIo(size_t n)
    {
        o_pool = 0;
        o_actions = n * 8;
        in_bytes = (n * 12 + 15) & ~(size_t)15;
        o_rewards = in_bytes;
        o_obs = o_rewards + n * 8;
        o_count = o_obs + n * 16;
        o_truncated = o_count + 4;
        bytes = (o_truncated + n + 15) & ~(size_t)15;
    }
};

bool fused_step_possible(const rf_ctx *ctx)
{
    return ctx->env_fused && ctx->env_axis;
}

} // namespace

extern "C" {

int rf_env_configure(rf_ctx *ctx, const rf_env_config *cfg)
{
    RF_REQUIRE(ctx != nullptr && cfg != nullptr, "rf_env_configure: NULL argument");
    RF_REQUIRE(cfg->n > 0 && cfg->n_actions > 0 && cfg->n_actions <= 32, "rf_env_configure: bad n / n_actions");
    RF_REQUIRE(cfg->frame_height > 0 && cfg->spp > 0, "rf_env_configure: frame_height, spp must be positive");
    RF_REQUIRE(cfg->gray_mode == RF_GRAY_15BIT || cfg->gray_mode == RF_GRAY_14BIT, "rf_env_configure: gray_mode");
    const uint64_t need = (uint64_t)cfg->n * cfg->frame_height * cfg->frame_height;
    RF_REQUIRE(need <= ctx->n_states, "rf_env_configure: %llu pixels but only %llu RNG states (rf_seed first)",
               (unsigned long long)need, (unsigned long long)ctx->n_states);
    RF_HIP(hipSetDevice(ctx->device));
    drop_env_graph(ctx);
    RF_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->env_block) {
        RF_HIP(hipFree(ctx->env_block));
        ctx->env_block = nullptr;
    }
    ctx->env_ready = false;
    const size_t n = (size_t)cfg->n;
    // carve one allocation (256-B aligned pieces)
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_state = take(n * 8), o_steps = take(n * 4), o_div = take(n * 4), o_last = take(n * 4),
                 o_oldw = take(n * 8), o_oldf = take(n * 4), o_cam = take(n * 36), o_rect = take(n * 8),
                 o_cam2 = take(n * 36), o_rect2 = take(n * 8), o_didx = take(n * 4), o_done = take(n), o_sums2 = take(n * 16),
                 o_drank = take(n * 4);
    const EnvIo io(n);
    const size_t o_io = take(io.bytes);
    RF_HIP(dev_malloc(&ctx->env_block, off));
    RF_HIP(hipMemsetAsync(ctx->env_block, 0, off, ctx->stream));
    char *base = (char *)ctx->env_block;
    rf::EnvState &s = ctx->env;
    s.state = (float *)(base + o_state);
    s.steps = (int *)(base + o_steps);
    s.diverging = (int *)(base + o_div);
    s.last_diff = (float *)(base + o_last);
    s.old_wrapped = (float *)(base + o_oldw);
    s.old_focus = (float *)(base + o_oldf);
    s.cam_dyn = (float *)(base + o_cam);
    s.rect = (float *)(base + o_rect);
    s.cam_dyn2 = (float *)(base + o_cam2);
    s.rect2 = (float *)(base + o_rect2);
    s.done_index = (int *)(base + o_didx);
    s.done_count = (int *)(base + o_io + io.o_count);
    s.obs = (float *)(base + o_io + io.o_obs);
    s.reward = (double *)(base + o_io + io.o_rewards);
    s.truncated = (uint8_t *)(base + o_io + io.o_truncated);
    s.done = (uint8_t *)(base + o_done);
    s.sums2 = (unsigned long long *)(base + o_sums2);
    s.done_rank = (int *)(base + o_drank);
    ctx->d_actions = (int *)(base + o_io + io.o_actions);
    ctx->d_pool = (float *)(base + o_io + io.o_pool);

    rf::EnvConfig &c = ctx->env_cfg;
    c.n = cfg->n;
    c.n_actions = cfg->n_actions;
    for (int i = 0; i < 32; ++i)
        c.action_set[i] = cfg->action_set[i];
    c.limit_lo = cfg->limit_lo;
    c.limit_hi = cfg->limit_hi;
    c.max_steps = cfg->max_steps;
    c.diverge_threshold = cfg->diverge_threshold;
    c.early_end_steps = cfg->early_end_steps;
    for (int i = 0; i < 4; ++i) {
        c.mid[i] = cfg->mid[i];
        c.scale[i] = cfg->scale[i];
    }
    c.reward_scale = cfg->reward_scale;
    c.on_target_span = cfg->on_target_span;
    c.half_width = cfg->half_width;
    c.half_height = cfg->half_height;
    c.tan_half_r = cfg->tan_half_r;
    for (int i = 0; i < 3; ++i) {
        c.look_from[i] = cfg->look_from[i];
        c.cam_u[i] = cfg->cam_u[i];
        c.cam_v[i] = cfg->cam_v[i];
        c.cam_w[i] = cfg->cam_w[i];
    }
    c.frame_pixels = (unsigned long long)cfg->frame_height * (unsigned long long)cfg->frame_height;
    ctx->env_host = *cfg;
    {
        int rc = ensure_focus(ctx, cfg->n); // (the environment kernels zero and read the sums themselves)
        if (rc != RF_OK)
            return rc;
        ctx->env.sums = ctx->d_sums;
    }
    ctx->cs = rf::CamStatic{cfg->look_from[0], cfg->look_from[1], cfg->look_from[2], cfg->cam_u[0], cfg->cam_u[1],
                            cfg->cam_u[2],     cfg->cam_v[0],     cfg->cam_v[1],     cfg->cam_v[2], cfg->lens_radius,
                            0.0f,              0.0f,              0};
    lens_split(ctx->cs);
    // canonical frame -> horizontal = (h2, +0, +0), vertical = (+0, v2, +0): the AXIS kernels apply
    // the target's half side is target * tan_half_r at distance target; the frame's half width at
    // that distance is target * half_width (camera.py:147-160, world.py:114-116)
    ctx->hit_fraction = (cfg->half_width > 0.0 && cfg->tan_half_r > 0.0 && cfg->tan_half_r < cfg->half_width)
                            ? cfg->tan_half_r / cfg->half_width
                            : (cfg->tan_half_r >= cfg->half_width ? 1.0 : 0.658);
    ctx->env_axis = cfg->look_from[0] == 0.0f && cfg->look_from[1] == 0.0f && cfg->look_from[2] == 0.0f &&
                    cfg->cam_u[0] == 1.0f && cfg->cam_u[1] == 0.0f && cfg->cam_u[2] == 0.0f &&
                    cfg->cam_v[0] == 0.0f && cfg->cam_v[1] == 1.0f && cfg->cam_v[2] == 0.0f &&
                    !signbit(cfg->cam_u[1]) && !signbit(cfg->cam_u[2]) && !signbit(cfg->cam_v[0]) &&
                    !signbit(cfg->cam_v[2]) && cfg->half_width > 0.0 && cfg->half_height > 0.0;
    ctx->scene_n = 0; // the env owns the scene arrays from now on
    ctx->env_pending = -1;
    ctx->env_planned = false;
    ctx->env_ready = true;
    return RF_OK;
}

int rf_env_reset(rf_ctx *ctx, const float *host_states, float *host_obs)
{
    RF_REQUIRE(ctx != nullptr && host_states != nullptr && host_obs != nullptr, "rf_env_reset: NULL argument");
    RF_REQUIRE(ctx->env_ready, "rf_env_reset: rf_env_configure first");
    RF_HIP(hipSetDevice(ctx->device));
    ctx->env_pending = -1;
    ctx->env_planned = false;
    ctx->env_needs_reset = false;
    const rf_env_config &h = ctx->env_host;
    const int n = h.n, fh = h.frame_height;
    RF_HIP(hipMemcpyAsync(ctx->env.state, host_states, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    const dim3 grid((n + 255) / 256), block(256);
    hipLaunchKernelGGL(rf::env_pre_kernel, grid, block, 0, ctx->stream, ctx->env_cfg, ctx->env, (const int *)nullptr);
    int rc = launch_render(ctx, n, fh, fh, h.spp, ctx->env.cam_dyn, ctx->env.rect, ctx->env_axis);
    if (rc == RF_OK && fused_step_possible(ctx))
        rc = ensure_frames2(ctx, n, fh, fh); // (not inside a step: the first one after this may already be captured)
    if (rc == RF_OK)
        rc = launch_focus(ctx, n, fh, fh, h.gray_mode, nullptr, true);
    if (rc != RF_OK)
        return rc;
    hipLaunchKernelGGL(rf::env_post_kernel, grid, block, 0, ctx->stream, ctx->env_cfg, ctx->env,
                       (const double *)nullptr, 1);
    RF_HIP(hipGetLastError());
    RF_HIP(hipMemcpyAsync(host_obs, ctx->env.obs, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
    RF_HIP(hipStreamSynchronize(ctx->stream));
    ctx->env_scene_len = n;
    ctx->env_last_partial = false;
    return RF_OK;
}

} // extern "C"

namespace {

bool env_one_sync(const rf_ctx *ctx)
{
    const int n = ctx->env_host.n, fh = ctx->env_host.frame_height;
    const long tiles = (long)((fh + 127) / 128) * ((fh + 5) / 6); // (blocks of the default 128 x 6 tiles, rf_coop2.h)
    return (long)n * tiles <= ctx->env_one_sync_max;
}

// Enqueues one whole step on the ctx's stream without waiting for anything: uploads, the full
// render + focus, the glue kernels, the auto-reset render for all n slots (env_reset_kernel marks
// the unused ones, whose blocks exit at once), the downloads.  Used directly and under stream
// capture.
int enqueue_env_step(rf_ctx *ctx, const int32_t *actions, const float *pool, float *obs, double *rewards,
                     uint8_t *truncated, int *count, uint8_t *host_io = nullptr)
{
    // host_io: the host side is an image of the device's io block (EnvIo: the pinned staging buffer of the replayed
    // step) -- one copy in, one copy out; otherwise the caller's six separate arrays
    const rf_env_config &h = ctx->env_host;
    const int n = h.n, fh = h.frame_height;
    const dim3 grid((n + 255) / 256), block(256);
    const EnvIo io((size_t)n);
    uint8_t *const d_io = (uint8_t *)ctx->d_pool; // (the io block starts with the pool)
    if (host_io) {
        RF_HIP(hipMemcpyAsync(d_io, host_io, io.in_bytes, hipMemcpyHostToDevice, ctx->stream));
    } else {
        RF_HIP(hipMemcpyAsync(ctx->d_actions, actions, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        RF_HIP(hipMemcpyAsync(ctx->d_pool, pool, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    int rc = RF_OK;
    if (fused_step_possible(ctx)) {
        // One render launch and one focus launch per step.  Which environments end depends on their counters alone
        // (env_pre_kernel), so they are ranked and the compacted scene of the auto-reset is packed BEFORE the render;
        // the r-th of them is rendered as row r of that set with the RNG streams of slot r (render.py:217), i.e. right
        // after slot r's own frame: the blocks of the slots below the count make two passes (render_kernel_coop2<.., TWO>).
        // The step's frames of those slots go to frames2, so that the frame buffer ends up as the two launches leave it.
        hipLaunchKernelGGL(rf::env_reset_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->env_cfg, ctx->env,
                           (const float *)ctx->d_pool, rf::kEnvResetPlan, (const int *)ctx->d_actions);
        const SecondPass second{ctx->env.done_count, ctx->env.cam_dyn2, ctx->env.rect2};
        rc = launch_render(ctx, n, fh, fh, h.spp, ctx->env.cam_dyn, ctx->env.rect, ctx->env_axis, false, &second);
        if (rc == RF_OK)
            rc = launch_focus(ctx, n, fh, fh, h.gray_mode, nullptr, true, ctx->env.done_count);
        if (rc != RF_OK)
            return rc;
        hipLaunchKernelGGL(rf::env_finish_kernel, grid, block, 0, ctx->stream, ctx->env_cfg, ctx->env,
                           (const float *)ctx->d_pool);
    } else {
        hipLaunchKernelGGL(rf::env_pre_kernel, grid, block, 0, ctx->stream, ctx->env_cfg, ctx->env,
                           (const int *)ctx->d_actions);
        rc = launch_render(ctx, n, fh, fh, h.spp, ctx->env.cam_dyn, ctx->env.rect, ctx->env_axis, false);
        if (rc == RF_OK)
            rc = launch_focus(ctx, n, fh, fh, h.gray_mode, nullptr, true);
        if (rc != RF_OK)
            return rc;
        hipLaunchKernelGGL(rf::env_post_kernel, grid, block, 0, ctx->stream, ctx->env_cfg, ctx->env,
                           (const double *)nullptr, 0);
        hipLaunchKernelGGL(rf::env_reset_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->env_cfg, ctx->env,
                           (const float *)ctx->d_pool, rf::kEnvResetBoth);
        rc = launch_render(ctx, n, fh, fh, h.spp, ctx->env.cam_dyn2, ctx->env.rect2, ctx->env_axis, false);
        if (rc == RF_OK)
            rc = launch_focus(ctx, n, fh, fh, h.gray_mode, ctx->env.rect2, true);
        if (rc != RF_OK)
            return rc;
        hipLaunchKernelGGL(rf::env_reset_post_kernel, grid, block, 0, ctx->stream, ctx->env_cfg, ctx->env,
                           (const double *)nullptr, (const float *)nullptr);
    }
    if (host_io) {
        RF_HIP(hipMemcpyAsync(host_io + io.o_rewards, d_io + io.o_rewards, io.bytes - io.o_rewards, hipMemcpyDeviceToHost,
                              ctx->stream));
    } else {
        RF_HIP(hipMemcpyAsync(count, ctx->env.done_count, 4, hipMemcpyDeviceToHost, ctx->stream));
        RF_HIP(hipMemcpyAsync(rewards, ctx->env.reward, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
        RF_HIP(hipMemcpyAsync(truncated, ctx->env.truncated, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        RF_HIP(hipMemcpyAsync(obs, ctx->env.obs, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
    }
    return RF_OK;
}

// First half of a step: transform, enders, full render + focus, observations, rewards, flags, and
// the ranking of the environments that ended (vector_environment.py:124-135).  Synchronises once:
// *k, rewards and truncated are final on return; the observations of the environments that did not
// end are final on the device.
int env_step_begin(rf_ctx *ctx, const int32_t *host_actions, double *host_rewards, uint8_t *host_truncated, int *k)
{
    const rf_env_config &h = ctx->env_host;
    const int n = h.n, fh = h.frame_height;
    RF_HIP(hipMemcpyAsync(ctx->d_actions, host_actions, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    const dim3 grid((n + 255) / 256), block(256);
    hipLaunchKernelGGL(rf::env_pre_kernel, grid, block, 0, ctx->stream, ctx->env_cfg, ctx->env,
                       (const int *)ctx->d_actions);
    int rc = launch_render(ctx, n, fh, fh, h.spp, ctx->env.cam_dyn, ctx->env.rect, ctx->env_axis);
    if (rc == RF_OK)
        rc = launch_focus(ctx, n, fh, fh, h.gray_mode, nullptr, true);
    if (rc != RF_OK)
        return rc;
    hipLaunchKernelGGL(rf::env_post_kernel, grid, block, 0, ctx->stream, ctx->env_cfg, ctx->env,
                       (const double *)nullptr, 0);
    hipLaunchKernelGGL(rf::env_reset_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->env_cfg, ctx->env,
                       (const float *)nullptr, rf::kEnvResetRank);
    RF_HIP(hipGetLastError());
    RF_HIP(hipMemcpyAsync(k, ctx->env.done_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    RF_HIP(hipMemcpyAsync(host_rewards, ctx->env.reward, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    RF_HIP(hipMemcpyAsync(host_truncated, ctx->env.truncated, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    RF_HIP(hipStreamSynchronize(ctx->stream));
    return RF_OK;
}

// Second half: the k environments that ended take host_pool's rows 0..k-1 in index order and are
// rendered and scored again (vector_environment.py:137-151), with the launch sized by k.
int env_step_end(rf_ctx *ctx, const float *host_pool, int k, float *host_obs)
{
    const rf_env_config &h = ctx->env_host;
    const int n = h.n, fh = h.frame_height;
    if (k > 0) {
        RF_HIP(hipMemcpyAsync(ctx->d_pool, host_pool, (size_t)k * 8, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(rf::env_reset_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->env_cfg, ctx->env,
                           (const float *)ctx->d_pool, rf::kEnvResetApply);
        int rc = launch_render(ctx, k, fh, fh, h.spp, ctx->env.cam_dyn2, ctx->env.rect2, ctx->env_axis);
        if (rc == RF_OK)
            rc = launch_focus(ctx, k, fh, fh, h.gray_mode, nullptr, true);
        if (rc != RF_OK)
            return rc;
        hipLaunchKernelGGL(rf::env_reset_post_kernel, dim3((k + 255) / 256), dim3(256), 0, ctx->stream, ctx->env_cfg,
                           ctx->env, (const double *)nullptr, (const float *)nullptr);
        RF_HIP(hipGetLastError());
    }
    RF_HIP(hipMemcpyAsync(host_obs, ctx->env.obs, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
    RF_HIP(hipStreamSynchronize(ctx->stream));
    return RF_OK;
}

} // namespace

extern "C" {

int rf_env_step(rf_ctx *ctx, const int32_t *host_actions, const float *host_pool, float *host_obs,
                double *host_rewards, uint8_t *host_truncated, int *host_n_reset)
{
    RF_REQUIRE(ctx != nullptr && host_actions && host_pool && host_obs && host_rewards && host_truncated,
               "rf_env_step: NULL argument");
    RF_REQUIRE(ctx->env_ready, "rf_env_step: rf_env_configure first");
    RF_REQUIRE(ctx->env_pending < 0, "rf_env_step: a two-phase step is open (rf_env_step_end first)");
    RF_REQUIRE(!ctx->env_needs_reset, "rf_env_step: a step was aborted (rf_env_reset first)");
    RF_HIP(hipSetDevice(ctx->device));
    const rf_env_config &h = ctx->env_host;
    const int n = h.n;
    for (int i = 0; i < n; ++i)
        RF_REQUIRE(host_actions[i] >= 0 && host_actions[i] < h.n_actions, "rf_env_step: action %d of env %d out of range",
                   host_actions[i], i);
    int k = 0;
    // vector_environment.py:137-151: the envs that just ended are rendered again.  Small
    // configurations are launch- and sync-bound: their step is enqueued in one go (see
    // enqueue_env_step) and, from the second step on (all buffers have their final size by then),
    // replayed as one hipGraph through pinned staging buffers; it ends with its only host
    // synchronisation.  Large ones size the auto-reset launch by the count, which costs one round
    // trip and saves up to a few hundred thousand empty blocks.
    const bool fused = fused_step_possible(ctx); // (one render launch, no count to wait for: enqueued in one go at any size)
    if (fused || env_one_sync(ctx)) {
        const EnvIo io((size_t)n);
        const size_t bytes = io.bytes;
        const bool graph = ctx->env_graph_enabled && !ctx->timing && ctx->env_steps >= 1;
        if (!graph) {
            int rc = enqueue_env_step(ctx, host_actions, host_pool, host_obs, host_rewards, host_truncated, &k);
            if (rc != RF_OK)
                return rc;
            RF_HIP(hipGetLastError());
            RF_HIP(hipStreamSynchronize(ctx->stream));
            ctx->env_last_branch = fused ? RF_ENV_BRANCH_FUSED : RF_ENV_BRANCH_ONE_SYNC;
        } else {
            if (ctx->h_stage_bytes < bytes) {
                if (ctx->env_graph)
                    (void)hipGraphExecDestroy(ctx->env_graph);
                ctx->env_graph = nullptr;
                if (ctx->h_stage)
                    RF_HIP(hipHostFree(ctx->h_stage));
                ctx->h_stage = nullptr;
                ctx->h_stage_bytes = 0;
                RF_HIP(host_malloc((void **)&ctx->h_stage, bytes));
                ctx->h_stage_bytes = bytes;
            }
            uint8_t *st = ctx->h_stage;
            if (!ctx->env_graph) {
                // Capture problems are not the caller's problem: the step then simply keeps being
                // enqueued call by call (same kernels, same results).
                hipGraph_t captured = nullptr;
                hipError_t he = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
                int rc = RF_OK;
                if (he == hipSuccess) {
                    rc = enqueue_env_step(ctx, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, st);
                    he = hipStreamEndCapture(ctx->stream, &captured);
                    if (he == hipSuccess && rc == RF_OK && ctx->env_graph_fail_once) {
                        ctx->env_graph_fail_once = false; // test hook: behave as if instantiation had failed
                        he = hipErrorUnknown;
                    } else if (he == hipSuccess && rc == RF_OK)
                        he = hipGraphInstantiate(&ctx->env_graph, captured, nullptr, nullptr, 0);
                    if (captured)
                        (void)hipGraphDestroy(captured);
                }
                if (he != hipSuccess || rc != RF_OK || !ctx->env_graph) {
                    (void)hipGetLastError();
                    ctx->env_graph = nullptr;
                    ctx->env_graph_enabled = false;
                    rc = enqueue_env_step(ctx, host_actions, host_pool, host_obs, host_rewards, host_truncated, &k);
                    if (rc != RF_OK)
                        return rc;
                    RF_HIP(hipGetLastError());
                    RF_HIP(hipStreamSynchronize(ctx->stream));
                    rfh::count_pixels((unsigned long long)(n + k) * (unsigned long long)h.frame_height * h.frame_height);
                    ctx->env_steps += 1;
                    ctx->env_scene_len = k > 0 ? k : n;
                    ctx->env_last_partial = k > 0;
                    ctx->env_last_branch = fused ? RF_ENV_BRANCH_FUSED : RF_ENV_BRANCH_ONE_SYNC;
                    if (host_n_reset)
                        *host_n_reset = k;
                    return RF_OK;
                }
            }
            memcpy(st + io.o_actions, host_actions, (size_t)n * 4);
            memcpy(st + io.o_pool, host_pool, (size_t)n * 8);
            RF_HIP(hipGraphLaunch(ctx->env_graph, ctx->stream));
            RF_HIP(hipStreamSynchronize(ctx->stream));
            memcpy(host_obs, st + io.o_obs, (size_t)n * 16);
            memcpy(host_rewards, st + io.o_rewards, (size_t)n * 8);
            memcpy(host_truncated, st + io.o_truncated, (size_t)n);
            k = *(const int *)(st + io.o_count);
            ctx->env_last_branch = fused ? RF_ENV_BRANCH_FUSED_GRAPH : RF_ENV_BRANCH_GRAPH;
        }
        // what this step really rendered: all n environments, then the k that ended (the other slots of the
        // second launch exit at once)
        rfh::count_pixels((unsigned long long)(n + k) * (unsigned long long)h.frame_height * h.frame_height);
    } else {
        // the step's flags and rewards are final after the first half; the count sizes the partial render
        int rc = env_step_begin(ctx, host_actions, host_rewards, host_truncated, &k);
        if (rc == RF_OK)
            rc = env_step_end(ctx, host_pool, k, host_obs);
        if (rc != RF_OK)
            return rc;
        ctx->env_last_branch = RF_ENV_BRANCH_COUNT_SIZED;
    }
    ctx->env_steps += 1;
    ctx->env_scene_len = k > 0 ? k : n;
    ctx->env_last_partial = k > 0;
    if (host_n_reset)
        *host_n_reset = k;
    return RF_OK;
}

int rf_env_last_step_branch(rf_ctx *ctx, int *branch)
{
    RF_REQUIRE(ctx != nullptr && branch != nullptr, "rf_env_last_step_branch: NULL argument");
    *branch = ctx->env_last_branch;
    return RF_OK;
}

int rf_env_step_begin(rf_ctx *ctx, const int32_t *host_actions, double *host_rewards, uint8_t *host_truncated,
                      int *host_n_reset)
{
    RF_REQUIRE(ctx != nullptr && host_actions && host_rewards && host_truncated && host_n_reset,
               "rf_env_step_begin: NULL argument");
    RF_REQUIRE(ctx->env_ready, "rf_env_step_begin: rf_env_configure first");
    RF_REQUIRE(ctx->env_pending < 0, "rf_env_step_begin: the previous step was not finished (rf_env_step_end)");
    RF_REQUIRE(!ctx->env_needs_reset, "rf_env_step_begin: a step was aborted (rf_env_reset first)");
    RF_HIP(hipSetDevice(ctx->device));
    drop_env_graph(ctx);
    const rf_env_config &h = ctx->env_host;
    for (int i = 0; i < h.n; ++i)
        RF_REQUIRE(host_actions[i] >= 0 && host_actions[i] < h.n_actions,
                   "rf_env_step_begin: action %d of env %d out of range", host_actions[i], i);
    int k = 0;
    int rc = env_step_begin(ctx, host_actions, host_rewards, host_truncated, &k);
    if (rc != RF_OK)
        return rc;
    ctx->env_pending = k;
    *host_n_reset = k;
    return RF_OK;
}

int rf_env_step_end(rf_ctx *ctx, const float *host_pool, float *host_obs)
{
    RF_REQUIRE(ctx != nullptr && host_obs != nullptr, "rf_env_step_end: NULL argument");
    RF_REQUIRE(ctx->env_ready && ctx->env_pending >= 0 && !ctx->env_planned, "rf_env_step_end: rf_env_step_begin first");
    RF_REQUIRE(ctx->env_pending == 0 || host_pool != nullptr, "rf_env_step_end: %d environments ended but host_pool is NULL",
               ctx->env_pending);
    RF_HIP(hipSetDevice(ctx->device));
    const int k = ctx->env_pending;
    ctx->env_pending = -1;
    int rc = env_step_end(ctx, host_pool, k, host_obs);
    if (rc == RF_OK) {
        ctx->env_steps += 1;
        ctx->env_scene_len = k > 0 ? k : ctx->env_host.n;
        ctx->env_last_partial = k > 0;
    } else {
        ctx->env_needs_reset = true; // the second half failed part way: only a reset makes the environment usable again
    }
    return rc;
}

int rf_env_step_plan(rf_ctx *ctx, const int32_t *host_actions, int *host_n_reset)
{
    RF_REQUIRE(ctx != nullptr && host_actions && host_n_reset, "rf_env_step_plan: NULL argument");
    RF_REQUIRE(ctx->env_ready, "rf_env_step_plan: rf_env_configure first");
    RF_REQUIRE(ctx->env_pending < 0, "rf_env_step_plan: the previous step was not finished");
    RF_REQUIRE(!ctx->env_needs_reset, "rf_env_step_plan: a step was aborted (rf_env_reset first)");
    RF_HIP(hipSetDevice(ctx->device));
    drop_env_graph(ctx);
    const rf_env_config &h = ctx->env_host;
    for (int i = 0; i < h.n; ++i)
        RF_REQUIRE(host_actions[i] >= 0 && host_actions[i] < h.n_actions,
                   "rf_env_step_plan: action %d of env %d out of range", host_actions[i], i);
    RF_HIP(hipMemcpyAsync(ctx->d_actions, host_actions, (size_t)h.n * 4, hipMemcpyHostToDevice, ctx->stream));
    // the kernel below applies the actions and advances the counters: from here until the step is open (a HIP failure
    // returns early) only a reset makes the environment usable again -- a retried step would apply the actions twice
    ctx->env_needs_reset = true;
    hipLaunchKernelGGL(rf::env_reset_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->env_cfg, ctx->env,
                       (const float *)nullptr, rf::kEnvResetRank, (const int *)ctx->d_actions);
    RF_HIP(hipGetLastError());
    int k = 0;
    RF_HIP(hipMemcpyAsync(&k, ctx->env.done_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    RF_HIP(hipStreamSynchronize(ctx->stream));
    ctx->env_pending = k;
    ctx->env_planned = true;
    ctx->env_needs_reset = false;
    *host_n_reset = k;
    return RF_OK;
}

int rf_env_step_run(rf_ctx *ctx, const float *host_pool, float *host_obs, double *host_rewards, uint8_t *host_truncated)
{
    RF_REQUIRE(ctx != nullptr && host_obs && host_rewards && host_truncated, "rf_env_step_run: NULL argument");
    RF_REQUIRE(ctx->env_ready && ctx->env_pending >= 0 && ctx->env_planned, "rf_env_step_run: rf_env_step_plan first");
    RF_REQUIRE(ctx->env_pending == 0 || host_pool != nullptr, "rf_env_step_run: %d environments ended but host_pool is NULL",
               ctx->env_pending);
    RF_HIP(hipSetDevice(ctx->device));
    const int k = ctx->env_pending;
    ctx->env_pending = -1;
    ctx->env_planned = false;
    ctx->env_needs_reset = true; // until the step has finished (a failure below returns early)
    const rf_env_config &h = ctx->env_host;
    const int n = h.n, fh = h.frame_height;
    const dim3 grid((n + 255) / 256), block(256);
    int rc = RF_OK;
    if (k > 0)
        RF_HIP(hipMemcpyAsync(ctx->d_pool, host_pool, (size_t)k * 8, hipMemcpyHostToDevice, ctx->stream));
    if (fused_step_possible(ctx)) {
        if (k > 0)
            hipLaunchKernelGGL(rf::env_reset_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->env_cfg, ctx->env,
                               (const float *)ctx->d_pool, rf::kEnvResetPack, (const int *)nullptr);
        const SecondPass second{ctx->env.done_count, ctx->env.cam_dyn2, ctx->env.rect2};
        rc = launch_render(ctx, n, fh, fh, h.spp, ctx->env.cam_dyn, ctx->env.rect, ctx->env_axis, false, &second);
        if (rc == RF_OK)
            rc = launch_focus(ctx, n, fh, fh, h.gray_mode, nullptr, true, ctx->env.done_count);
        if (rc != RF_OK)
            return rc;
        hipLaunchKernelGGL(rf::env_finish_kernel, grid, block, 0, ctx->stream, ctx->env_cfg, ctx->env,
                           (const float *)ctx->d_pool);
    } else {
        rc = launch_render(ctx, n, fh, fh, h.spp, ctx->env.cam_dyn, ctx->env.rect, ctx->env_axis, false);
        if (rc == RF_OK)
            rc = launch_focus(ctx, n, fh, fh, h.gray_mode, nullptr, true);
        if (rc != RF_OK)
            return rc;
        hipLaunchKernelGGL(rf::env_post_kernel, grid, block, 0, ctx->stream, ctx->env_cfg, ctx->env,
                           (const double *)nullptr, 0);
        if (k > 0) {
            hipLaunchKernelGGL(rf::env_reset_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->env_cfg, ctx->env,
                               (const float *)ctx->d_pool, rf::kEnvResetApply, (const int *)nullptr);
            rc = launch_render(ctx, k, fh, fh, h.spp, ctx->env.cam_dyn2, ctx->env.rect2, ctx->env_axis, false);
            if (rc == RF_OK)
                rc = launch_focus(ctx, k, fh, fh, h.gray_mode, nullptr, true);
            if (rc != RF_OK)
                return rc;
            hipLaunchKernelGGL(rf::env_reset_post_kernel, dim3((k + 255) / 256), dim3(256), 0, ctx->stream, ctx->env_cfg,
                               ctx->env, (const double *)nullptr, (const float *)nullptr);
        }
    }
    RF_HIP(hipGetLastError());
    RF_HIP(hipMemcpyAsync(host_rewards, ctx->env.reward, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    RF_HIP(hipMemcpyAsync(host_truncated, ctx->env.truncated, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    RF_HIP(hipMemcpyAsync(host_obs, ctx->env.obs, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
    RF_HIP(hipStreamSynchronize(ctx->stream));
    rfh::count_pixels((unsigned long long)(n + k) * (unsigned long long)fh * fh);
    ctx->env_needs_reset = false;
    ctx->env_steps += 1;
    ctx->env_scene_len = k > 0 ? k : n;
    ctx->env_last_partial = k > 0;
    return RF_OK;
}

int rf_env_render_states(rf_ctx *ctx, int k, const float *host_states, double *host_focus)
{
    RF_REQUIRE(ctx != nullptr && host_states != nullptr && host_focus != nullptr, "rf_env_render_states: NULL argument");
    RF_REQUIRE(ctx->env_ready, "rf_env_render_states: rf_env_configure first");
    const rf_env_config &h = ctx->env_host;
    RF_REQUIRE(k > 0 && k <= h.n, "rf_env_render_states: k=%d outside [1, %d]", k, h.n);
    RF_HIP(hipSetDevice(ctx->device));
    drop_env_graph(ctx);
    const int fh = h.frame_height;
    RF_HIP(hipMemcpyAsync(ctx->d_pool, host_states, (size_t)k * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(rf::env_pack_rows_kernel, dim3((k + 255) / 256), dim3(256), 0, ctx->stream, ctx->env_cfg, ctx->env,
                       (const float *)ctx->d_pool, k);
    int rc = launch_render(ctx, k, fh, fh, h.spp, ctx->env.cam_dyn2, ctx->env.rect2, ctx->env_axis);
    if (rc == RF_OK)
        rc = launch_focus(ctx, k, fh, fh, h.gray_mode);
    if (rc != RF_OK)
        return rc;
    RF_HIP(hipGetLastError());
    RF_HIP(hipMemcpyAsync(host_focus, ctx->d_var, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    RF_HIP(hipStreamSynchronize(ctx->stream));
    ctx->env_scene_len = k; // the renderer now holds this compacted set (what rf_env_render would draw)
    ctx->env_last_partial = true;
    return RF_OK;
}

int rf_env_step_end_given(rf_ctx *ctx, const float *host_pool, const double *host_focus, float *host_obs)
{
    RF_REQUIRE(ctx != nullptr && host_obs != nullptr, "rf_env_step_end_given: NULL argument");
    RF_REQUIRE(ctx->env_ready && ctx->env_pending >= 0 && !ctx->env_planned, "rf_env_step_end_given: rf_env_step_begin first");
    const int k = ctx->env_pending;
    RF_REQUIRE(k == 0 || (host_pool != nullptr && host_focus != nullptr),
               "rf_env_step_end_given: %d environments ended but host_pool / host_focus is NULL", k);
    RF_REQUIRE(k <= ctx->focus_cap, "rf_env_step_end_given: focus buffer smaller than %d", k); // (before anything changes)
    RF_HIP(hipSetDevice(ctx->device));
    ctx->env_pending = -1;
    ctx->env_needs_reset = true; // until the second half has finished (a HIP failure below returns early)
    const int n = ctx->env_host.n;
    if (k > 0) {
        RF_HIP(hipMemcpyAsync(ctx->d_pool, host_pool, (size_t)k * 8, hipMemcpyHostToDevice, ctx->stream));
        // the focus values were measured elsewhere: they take the place launch_focus would have filled
        RF_HIP(hipMemcpyAsync(ctx->d_var, host_focus, (size_t)k * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(rf::env_reset_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->env_cfg, ctx->env,
                           (const float *)ctx->d_pool, rf::kEnvResetApply);
        hipLaunchKernelGGL(rf::env_reset_post_kernel, dim3((k + 255) / 256), dim3(256), 0, ctx->stream, ctx->env_cfg,
                           ctx->env, (const double *)ctx->d_var, (const float *)nullptr);
        RF_HIP(hipGetLastError());
    }
    RF_HIP(hipMemcpyAsync(host_obs, ctx->env.obs, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
    RF_HIP(hipStreamSynchronize(ctx->stream));
    ctx->env_needs_reset = false;
    ctx->env_steps += 1;
    return RF_OK;
}

int rf_env_step_abort(rf_ctx *ctx)
{
    RF_REQUIRE(ctx != nullptr, "rf_env_step_abort: ctx is NULL");
    RF_REQUIRE(ctx->env_ready, "rf_env_step_abort: rf_env_configure first");
    if (ctx->env_pending >= 0) {
        // the episode bookkeeping of the environments that ended is half way through a step: only a
        // reset makes the environment usable again, and rf_env_step / _begin say so until then
        ctx->env_pending = -1;
        ctx->env_planned = false;
        ctx->env_needs_reset = true;
    }
    return RF_OK;
}

int rf_env_scene_len(rf_ctx *ctx, int *n_envs)
{
    RF_REQUIRE(ctx != nullptr && n_envs != nullptr, "rf_env_scene_len: NULL argument");
    RF_REQUIRE(ctx->env_ready, "rf_env_scene_len: rf_env_configure first");
    *n_envs = ctx->env_scene_len;
    return RF_OK;
}

int rf_env_render(rf_ctx *ctx, int frame_height, int spp, uint8_t *host_out)
{
    RF_REQUIRE(ctx != nullptr, "rf_env_render: ctx is NULL");
    RF_REQUIRE(ctx->env_ready && ctx->env_scene_len > 0, "rf_env_render: rf_env_reset first");
    RF_REQUIRE(ctx->env_pending < 0, "rf_env_render: a two-phase step is open (rf_env_step_end first)");
    RF_REQUIRE(frame_height > 0 && spp > 0, "rf_env_render: frame_height, spp must be positive");
    const int n = ctx->env_scene_len;
    const uint64_t need = (uint64_t)n * frame_height * frame_height;
    RF_REQUIRE(need <= ctx->n_states, "rf_env_render: %llu pixels but only %llu RNG states (rf_seed first)",
               (unsigned long long)need, (unsigned long long)ctx->n_states);
    RF_HIP(hipSetDevice(ctx->device));
    drop_env_graph(ctx);
    const bool partial = ctx->env_last_partial;
    int rc = launch_render(ctx, n, frame_height, frame_height, spp, partial ? ctx->env.cam_dyn2 : ctx->env.cam_dyn,
                           partial ? ctx->env.rect2 : ctx->env.rect, ctx->env_axis);
    if (rc != RF_OK)
        return rc;
    if (host_out)
        return rf_get_frames(ctx, 0, n, host_out);
    return RF_OK;
}

int rf_env_get_counters(rf_ctx *ctx, int32_t *host_steps, int32_t *host_diverging)
{
    RF_REQUIRE(ctx != nullptr && host_steps != nullptr && host_diverging != nullptr, "rf_env_get_counters: NULL argument");
    RF_REQUIRE(ctx->env_ready, "rf_env_get_counters: rf_env_configure first");
    RF_HIP(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)ctx->env_host.n * 4;
    RF_HIP(hipMemcpyAsync(host_steps, ctx->env.steps, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RF_HIP(hipMemcpyAsync(host_diverging, ctx->env.diverging, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RF_HIP(hipStreamSynchronize(ctx->stream));
    return RF_OK;
}

int rf_env_get_states(rf_ctx *ctx, float *host_states)
{
    RF_REQUIRE(ctx != nullptr && host_states != nullptr, "rf_env_get_states: NULL argument");
    RF_REQUIRE(ctx->env_ready, "rf_env_get_states: rf_env_configure first");
    RF_HIP(hipSetDevice(ctx->device));
    RF_HIP(hipMemcpyAsync(host_states, ctx->env.state, (size_t)ctx->env_host.n * 8, hipMemcpyDeviceToHost, ctx->stream));
    RF_HIP(hipStreamSynchronize(ctx->stream));
    return RF_OK;
}

} // extern "C"
