// rf_env.h -- device-resident DiscreteSteps-v0 step.
//
// The O(N) numpy glue the reference runs on the host every step --
//   DiscreteMoveTransformer.transform      environments/state_transformer.py:248-266
//   TimeLimitEnder | DivergingEnder        environments/episode_ender.py:137-170, :602-628
//   FastCameras / FastWorlds packing       graphics/camera.py:144-179, graphics/world.py:110-123
//   NormalizedObserver(DeltaObserver(..))  environments/state_observer.py:232-292, :472-517
//   Delta + Observation + OnTarget reward  environments/episode_rewarder.py:130-155, :226-292
//   same-step auto-reset                   environments/vector_environment.py:137-151
// -- as two small kernels around the render and focus kernels, so that a step moves only the
// actions and a pool of candidate reset states to the GPU and the observations / rewards /
// flags back.  Arithmetic follows the numpy expressions operation by operation (float32
// arrays with Python-float scalars stay float32, the action set is float64, rewards end up
// float64), so results equal reinfocus_amd/environments/harness.py bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rf_env_types.h"
#include "rf_math.h"

namespace rf {

// ndarray.var() of a frame's Laplacian from its exact integer sums -- focus_finalize's expression (rf_focus.h); the
// environment kernels take the variance from the sums themselves, which saves the replayed step a launch per measure
__device__ __forceinline__ double env_variance(const EnvConfig &c, const unsigned long long *sums, int slot)
{
    return variance_from_sums(c.frame_pixels, sums[2 * slot], sums[2 * slot + 1]);
}


// camera.py:144-179 + world.py:110-123 for one environment
__device__ __forceinline__ void pack_scene(const EnvConfig &c, float target, float fp, float *dyn, float *rc)
{
    const float a = (float)(c.half_width * (double)fp);   // f32(hw * fp)
    const float b = (float)(c.half_height * (double)fp);
    const float h2 = (float)((2.0 * c.half_width) * (double)fp);
    const float v2 = (float)((2.0 * c.half_height) * (double)fp);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float sum = (a * c.cam_u[k] + b * c.cam_v[k]) + fp * c.cam_w[k]; // numpy.sum, left to right
        dyn[k] = c.look_from[k] - sum;
        dyn[3 + k] = h2 * c.cam_u[k];
        dyn[6 + k] = v2 * c.cam_v[k];
    }
    rc[0] = (float)((double)target * c.tan_half_r);
    rc[1] = -target;
}

// transformer -> ender.step -> scene of every env (vector_environment.py:124-126 + the
// update_targets / update_focus_planes of FocusObserver.observe)
__device__ __forceinline__ void env_pre_one(const EnvConfig &c, const EnvState &s, const int *actions, int e)
{
    float target = s.state[2 * e], focus = s.state[2 * e + 1];
    if (actions) { // step: new = clip(f32(f64(old) + move), lo, hi) on BOTH columns
        focus = (float)((double)focus + c.action_set[actions[e]]);
        focus = fminf(fmaxf(focus, c.limit_lo), c.limit_hi);
        target = fminf(fmaxf(target, c.limit_lo), c.limit_hi);
        s.state[2 * e] = target;
        s.state[2 * e + 1] = focus;
        // enders (episode_ender.py:137-148, :602-607)
        s.steps[e] += 1;
        const float diff = fabsf(target - focus);
        if (diff > s.last_diff[e] + c.diverge_threshold)
            s.diverging[e] += 1;
        s.last_diff[e] = diff;
        // which environments end is settled here already (env_post_kernel repeats it): the flags depend on the counters
        // alone, not on what the step observes -- the fused step ranks the ended ones BEFORE its render
        bool trunc = s.diverging[e] >= c.early_end_steps;
        if (c.max_steps > 0)
            trunc = (s.steps[e] >= c.max_steps) || trunc;
        s.done[e] = trunc ? 1 : 0;
    } else { // reset of every env
        s.steps[e] = 0;
        s.diverging[e] = 0;
        s.last_diff[e] = fabsf(target - focus);
    }
    pack_scene(c, target, focus, s.cam_dyn + 9 * e, s.rect + 2 * e);
    s.sums[2 * e] = 0; // (the focus measure of the render that follows accumulates into them)
    s.sums[2 * e + 1] = 0;
}

__global__ void env_pre_kernel(EnvConfig c, EnvState s, const int *actions)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < c.n)
        env_pre_one(c, s, actions, e);
}

__device__ __forceinline__ float normalize1(const EnvConfig &c, int k, float v)
{
    return fminf(fmaxf((v - c.mid[k]) / c.scale[k], -1.0f), 1.0f);
}

// observe -> reward -> done flags (vector_environment.py:128-135); `first` = reset() call
// focus_values == nullptr: the variance comes from the sums the focus kernel left (env_variance)
__device__ __forceinline__ void env_post_one(const EnvConfig &c, const EnvState &s, const double *focus_values, int first, int e)
{
    const float target = s.state[2 * e], focus = s.state[2 * e + 1];
    const float w0 = focus, w1 = (float)(focus_values ? focus_values[e] : env_variance(c, s.sums, e));
    float d0 = 0.0f, d1 = 0.0f;
    if (!first) {
        d0 = w0 - s.old_wrapped[2 * e];
        d1 = w1 - s.old_wrapped[2 * e + 1];
    }
    s.old_wrapped[2 * e] = w0;
    s.old_wrapped[2 * e + 1] = w1;
    const float o0 = normalize1(c, 0, w0), o1 = normalize1(c, 1, w1), o2 = normalize1(c, 2, d0),
                o3 = normalize1(c, 3, d1);
    s.obs[4 * e] = o0;
    s.obs[4 * e + 1] = o1;
    s.obs[4 * e + 2] = o2;
    s.obs[4 * e + 3] = o3;
    if (first) {
        s.old_focus[e] = focus;
        s.truncated[e] = 0;
        s.done[e] = 0;
        s.reward[e] = 0.0;
        return;
    }
    // (abs(focus - old) * -1.0 / scale + obs[:, 1]) + ((abs(target - focus) < span) * 1.0 + 0.0)
    const float moved = fabsf(focus - s.old_focus[e]) * -1.0f / c.reward_scale;
    s.old_focus[e] = focus;
    const double on_target = (fabsf(target - focus) < c.on_target_span ? 1.0 : 0.0) * 1.0 + 0.0;
    s.reward[e] = (double)(moved + o1) + on_target;
    bool trunc = s.diverging[e] >= c.early_end_steps;
    if (c.max_steps > 0)
        trunc = (s.steps[e] >= c.max_steps) || trunc;
    s.truncated[e] = trunc ? 1 : 0;
    s.done[e] = trunc ? 1 : 0;
}

__global__ void env_post_kernel(EnvConfig c, EnvState s, const double *focus_values, int first)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < c.n)
        env_post_one(c, s, focus_values, first, e);
}

// Ranks the done envs in index order (single block, running offset) and applies the
// initializer's r-th candidate state to the r-th done env (vector_environment.py:138-142),
// resets its enders, and packs the compacted scene of the partial render.
// mode kEnvResetBoth does all of it in one launch (rf_env_step); the two-phase step of a sharded
// environment (rf_env_step_begin / rf_env_step_end) ranks first (kEnvResetRank: done_index,
// done_count; no pool yet -- which rows of the initializer's pool a shard takes depends on how
// many environments ended in the shards before it) and applies later (kEnvResetApply).
// kEnvResetPlan (the fused step, before its render): the work of env_pre_kernel first (`actions`), then ranks and packs
// the compacted scene, but leaves the environments' state alone -- the step's observations and rewards still have to be
// taken from what the step left; env_finish_kernel applies the initializer's states afterwards.
// kEnvResetPack: kEnvResetPlan's packing half for a ranking that exists already (rf_env_step_plan / rf_env_step_run).
// `actions` != null in a ranking mode: the transformer / ender work of env_pre_kernel first.
constexpr int kEnvResetBoth = 0, kEnvResetRank = 1, kEnvResetApply = 2, kEnvResetPlan = 3, kEnvResetPack = 4;

__device__ __forceinline__ void env_apply_state(const EnvState &s, const float *pool, int r, int e)
{
    const float target = pool[2 * r], focus = pool[2 * r + 1];
    s.state[2 * e] = target;
    s.state[2 * e + 1] = focus;
    s.steps[e] = 0;
    s.diverging[e] = 0;
    s.last_diff[e] = fabsf(target - focus);
}

__device__ __forceinline__ void env_apply_reset(const EnvConfig &c, const EnvState &s, const float *pool, int r, int e)
{
    env_apply_state(s, pool, r, e);
    pack_scene(c, pool[2 * r], pool[2 * r + 1], s.cam_dyn2 + 9 * r, s.rect2 + 2 * r);
}

__global__ __launch_bounds__(1024) void env_reset_kernel(EnvConfig c, EnvState s, const float *pool, int mode,
                                                         const int *actions = nullptr)
{
    __shared__ int wave_sum[16];
    __shared__ int running;
    if (mode == kEnvResetApply) {
        const int count = *s.done_count;
        for (int r = (int)threadIdx.x; r < count; r += 1024) {
            env_apply_reset(c, s, pool, r, s.done_index[r]);
            s.sums[2 * r] = 0; // (the auto-reset's focus measure accumulates into slot r)
            s.sums[2 * r + 1] = 0;
        }
        return;
    }
    if (mode == kEnvResetPack) {
        const int count = *s.done_count;
        for (int r = (int)threadIdx.x; r < count; r += 1024) {
            s.done_rank[s.done_index[r]] = r;
            pack_scene(c, pool[2 * r], pool[2 * r + 1], s.cam_dyn2 + 9 * r, s.rect2 + 2 * r);
            s.sums2[2 * r] = 0;
            s.sums2[2 * r + 1] = 0;
        }
        return;
    }
    if (threadIdx.x == 0)
        running = 0;
    __syncthreads();
    for (int base = 0; base < c.n; base += 1024) {
        const int e = base + threadIdx.x;
        if (actions != nullptr && e < c.n)
            env_pre_one(c, s, actions, e); // (sets done[e], read by the same thread below)
        const bool d = e < c.n && s.done[e];
        const unsigned long long ballot = __ballot(d);
        const int lane_rank = __builtin_amdgcn_mbcnt_hi((unsigned)(ballot >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((unsigned)ballot, 0));
        if ((threadIdx.x & 63) == 0)
            wave_sum[threadIdx.x >> 6] = __popcll(ballot);
        __syncthreads();
        int before = running, total = 0;
        for (int i = 0; i < 16; ++i) {
            before += (i < (int)(threadIdx.x >> 6)) ? wave_sum[i] : 0;
            total += wave_sum[i];
        }
        if (d) {
            const int r = before + lane_rank;
            s.done_index[r] = e;
            if (mode == kEnvResetBoth)
                env_apply_reset(c, s, pool, r, e);
            if (mode == kEnvResetPlan) {
                s.done_rank[e] = r;
                pack_scene(c, pool[2 * r], pool[2 * r + 1], s.cam_dyn2 + 9 * r, s.rect2 + 2 * r);
                s.sums2[2 * r] = 0;
                s.sums2[2 * r + 1] = 0;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0)
            running += total;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        *s.done_count = running;
    // the auto-reset render is enqueued for all n slots: mark the ones it has to skip
    if (mode == kEnvResetBoth) {
        for (int r = running + (int)threadIdx.x; r < c.n; r += 1024)
            s.rect2[2 * r] = __builtin_bit_cast(float, kSkipEnvBits);
        for (int r = (int)threadIdx.x; r < running; r += 1024) { // (the auto-reset's focus measure accumulates into slot r)
            s.sums[2 * r] = 0;
            s.sums[2 * r + 1] = 0;
        }
    }
}

// Scene of k given states as rows 0..k-1 of the compacted set (cam_dyn2 / rect2): the packing half of
// FocusObserver.observe(new_state, indices) (state_observer.py:377-378) for rows that belong to
// environments of OTHER contexts -- the exact mode of a sharded environment renders compacted row r
// on the context that owns the RNG states of pixel indices [r h w, (r + 1) h w) (render.py:217).
__global__ void env_pack_rows_kernel(EnvConfig c, EnvState s, const float *states, int k)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < k)
        pack_scene(c, states[2 * r], states[2 * r + 1], s.cam_dyn2 + 9 * r, s.rect2 + 2 * r);
}

// observations of the freshly reset envs (DeltaObserver.reset: zero deltas) and the
// rewarder's reset (vector_environment.py:144-148); planned_pool != null: the fused step -- the initializer's states
// are applied only now (kEnvResetPlan), and the re-rendered frames' sums are in sums2
__device__ __forceinline__ void env_reset_post_one(const EnvConfig &c, const EnvState &s, const double *focus_values,
                                                   const float *planned_pool, int r, int e)
{
    if (planned_pool)
        env_apply_state(s, planned_pool, r, e);
    const float focus = s.state[2 * e + 1];
    const float w0 = focus,
                w1 = (float)(focus_values ? focus_values[r] : env_variance(c, planned_pool ? s.sums2 : s.sums, r));
    s.old_wrapped[2 * e] = w0;
    s.old_wrapped[2 * e + 1] = w1;
    s.obs[4 * e] = normalize1(c, 0, w0);
    s.obs[4 * e + 1] = normalize1(c, 1, w1);
    s.obs[4 * e + 2] = normalize1(c, 2, 0.0f);
    s.obs[4 * e + 3] = normalize1(c, 3, 0.0f);
    s.old_focus[e] = focus;
}

__global__ void env_reset_post_kernel(EnvConfig c, EnvState s, const double *focus_values, const float *planned_pool)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < *s.done_count)
        env_reset_post_one(c, s, focus_values, planned_pool, r, s.done_index[r]);
}

// the fused step's last kernel: env_post_kernel and env_reset_post_kernel as one launch (both measures are done by then)
__global__ void env_finish_kernel(EnvConfig c, EnvState s, const float *planned_pool)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.n)
        return;
    env_post_one(c, s, nullptr, 0, e);
    if (s.done[e])
        env_reset_post_one(c, s, nullptr, planned_pool, s.done_rank[e], e);
}

} // namespace rf
