// rf_env_types.h -- configuration and device arrays of the device-resident environment step (kernels: rf_env.h)
#pragma once

#include <stdint.h>

namespace rf {

struct EnvConfig {
    int n;                    // environments
    int n_actions;
    double action_set[32];    // float64 moves (state_transformer.py:246 numpy.asarray(action_set))
    float limit_lo, limit_hi; // clip limits
    int max_steps;            // <= 0: no time limit (single-env DiscreteSteps)
    float diverge_threshold;  // target_radius / 2
    int early_end_steps;
    float mid[4], scale[4];   // NormalizedObserver (float32)
    float reward_scale;       // DeltaRewarder scale (target_radius * 2)
    float on_target_span;     // OnTargetRewarder span
    // camera / world packing
    double half_width, half_height, tan_half_r; // Python floats
    float look_from[3], cam_u[3], cam_v[3], cam_w[3];
    unsigned long long frame_pixels; // h * w of the frames the focus measure reduces
};

struct EnvState {            // all device arrays, length n unless noted
    float *state;            // [n][2] target, focus
    int *steps;              // TimeLimitEnder._steps
    int *diverging;          // DivergingEnder._diverging_steps
    float *last_diff;        // DivergingEnder._last_diff
    float *old_wrapped;      // [n][2] DeltaObserver._old_wrapped_observations
    float *old_focus;        // DeltaRewarder._old_states
    // per-step scratch / outputs
    float *cam_dyn, *rect;   // scene of all n envs
    float *cam_dyn2, *rect2; // compacted scene of the envs that reset this step
    int *done_index;         // [n] env index of the r-th reset env
    int *done_rank;          // [n] the inverse for the envs that ended (fused step)
    int *done_count;         // [1]
    float *obs;              // [n][4]
    double *reward;          // [n]
    uint8_t *truncated;      // [n]
    uint8_t *done;           // [n] scratch
    unsigned long long *sums; // [n][2] the focus measure's (sum, sum of squares): zeroed here before a measure, read after
    unsigned long long *sums2; // [n][2] the same for the re-rendered frames of a fused step (both measures are one launch there)
};

} // namespace rf
