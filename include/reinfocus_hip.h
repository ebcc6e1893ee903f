/*
 * reinfocus_hip.h -- C ABI of libreinfocus_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the render-and-measure hot path of jeffwhunter/reinfocus.
 * The reference has no FFI: its device side is numba @cuda.jit Python.  Each entry
 * point below names the reference interface it replaces (file:line relative to the
 * reference checkout); INTEGRATION.md shows the ctypes stub a reference maintainer
 * would add to bind them.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no torch / numpy types cross the boundary.
 *  - every function returns 0 on success, a negative rf_status otherwise;
 *    rf_last_error() returns a thread-local message (HIP errors carry file:line).
 *  - an rf_ctx owns ALL device memory (RNG states, scene parameters, frames,
 *    reduction partials) of one renderer on one GPU.  Host buffers are caller-owned.
 *  - one ctx is not thread-safe; different ctxs (e.g. one per GPU) are independent.
 *  - calls are synchronous with respect to the host buffers they are handed;
 *    rf_render(..., NULL) only enqueues (frames stay in HBM).
 *  - there is NO CPU fallback: without a usable gfx950 device rf_create fails.
 */
#ifndef REINFOCUS_HIP_H
#define REINFOCUS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rf_ctx rf_ctx;

typedef enum rf_status {
    RF_OK = 0,
    RF_ERR_INVALID = -1,   /* bad argument / call order (maps to AssertionError) */
    RF_ERR_HIP = -2,       /* HIP runtime error (maps to RuntimeError)           */
    RF_ERR_NO_DEVICE = -3, /* no usable GPU                                      */
    RF_ERR_OOM = -4        /* device or pinned-host allocation failed            */
} rf_status;

/* gray_mode for rf_focus: fixed-point RGB->gray coefficients (vision.py:24,
 * cv2.cvtColor COLOR_RGB2GRAY).  15 = OpenCV >= 4 (9798/19235/3735 >> 15, the
 * reference's pinned opencv-python 4.9); 14 = OpenCV 2/3 (4899/9617/1868 >> 14). */
#define RF_GRAY_15BIT 15
#define RF_GRAY_14BIT 14

const char *rf_last_error(void);
int rf_abi_version(void);

/* Number of visible HIP devices (0 on a CPU-only host; never fails the process). */
int rf_device_count(int *count);

/* Which physical GPU `device` is: its PCI bus id ("0000:c1:00.0", NUL-terminated, into bus_id[len], len >= 16)
 * and the NUMA node the host reports for it (/sys/bus/pci/devices/<id>/numa_node; -1 if unknown).
 * No reference counterpart (the reference is single-device): bench.py prints one row per rank so that an
 * N-GPU line proves it ran on N distinct GPUs, and ranks / shard threads pin themselves to their GPU's node. */
int rf_device_info(int device, char *bus_id, int len, int *numa_node);

/* Creates the per-renderer context on `device`.
 * Replaces: FastRenderer.__init__ device-side state (graphics/render.py:127-145). */
int rf_create(int device, rf_ctx **out);
int rf_destroy(rf_ctx *ctx);

/* (Re)creates `n_states` xoroshiro128+ states on the device:
 *   state[i] = jump_2^64 ^ (first_state_index + i) ( splitmix64(seed) ),
 * bit-identical to numba's sequential host seeding, computed in parallel with a
 * GF(2) jump-ahead.  first_state_index lets GPU g of a sharded run own the states a
 * single-device run would have used for its env slice.
 * Replaces: random.make_random_states (graphics/random.py:8-18) as called from
 * FastRenderer._make_random_states (graphics/render.py:248-257). */
int rf_seed(rf_ctx *ctx, uint64_t n_states, uint64_t seed, uint64_t first_state_index);
int rf_num_states(rf_ctx *ctx, uint64_t *n_states);

/* Checkpoint / test access to the RNG states; layout uint64[count][2] = (s0, s1),
 * numba's xoroshiro128p_dtype.  No reference counterpart (states are not reachable
 * through any reference API). */
int rf_get_states(rf_ctx *ctx, uint64_t first, uint64_t count, uint64_t *host_out);
int rf_set_states(rf_ctx *ctx, uint64_t first, uint64_t count, const uint64_t *host_in);

/* Uploads the scene of `n` environments.
 *   cam_dyn  float32[n][3][3]  rows: lower_left, horizontal, vertical
 *   rect     float32[n][2]     (half_side, z_pos); a half_side with the bit pattern 0x7FC0DEAD (a
 *                              NaN no arithmetic produces) marks a slot that render and focus skip
 *   origin, u, v  float32[3]   shared camera frame;  lens_radius float64
 * Replaces: the two cuda.to_device uploads in FastCameras._make_device_data
 * (graphics/camera.py:144-179) and FastWorlds._make_device_data
 * (graphics/world.py:110-123); the tuple layout is camera.py:39-56. */
int rf_set_scene(rf_ctx *ctx, int n, const float *cam_dyn, const float *rect,
                 const float origin[3], const float u[3], const float v[3],
                 double lens_radius);

/* Renders n frames of h x w pixels with spp samples per pixel into the ctx's frame
 * buffer (uint8[n][h][w][3], row 0 = bottom of the scene), advancing RNG states
 * [0, n*h*w).  If host_out is non-NULL the frames are also copied to it.
 * n must equal the n of the last rf_set_scene; n*h*w must not exceed rf_num_states.
 * Replaces: FastRenderer.render's launch + copy_to_host (graphics/render.py:165-188)
 * and the kernel FastRenderer._device_render (graphics/render.py:190-246). */
int rf_render(rf_ctx *ctx, int n, int h, int w, int spp, uint8_t *host_out);

/* Copies frames [first_env, first_env + n_envs) of the last render to the host. */
int rf_get_frames(rf_ctx *ctx, int first_env, int n_envs, uint8_t *host_out);

/* Replaces the ctx's frame buffer by caller-supplied images uint8[n][h][w][3], so
 * that rf_focus can score frames that did not come from rf_render
 * (vision.focus_values called on a host array, vision.py:28-39). */
int rf_upload_frames(rf_ctx *ctx, int n, int h, int w, const uint8_t *host_in);

/* Focus score of every frame in the ctx's frame buffer:
 * RGB->gray, 3x3 median (replicate border), 3x3 Laplacian (reflect-101 border,
 * saturated to uint8), population variance; host_var float64[n].
 * Replaces: vision.focus_values / focus_value (vision.py:11-39), i.e. the
 * cv2.cvtColor + cv2.medianBlur + cv2.Laplacian + ndarray.var chain. */
int rf_focus(rf_ctx *ctx, int n, int h, int w, int gray_mode, double *host_var);

/* rf_render(host_out = NULL) followed by rf_focus: what FocusObserver.observe does
 * per step (environments/state_observer.py:377-381) without the frame D2H. */
int rf_step(rf_ctx *ctx, int n, int h, int w, int spp, int gray_mode, double *host_var);

/* Name of the render kernel (template instance included) the ctx's last render launch used,
 * e.g. "render_kernel_coop2<true, 1, 4, 32>"; "none" before the first render.  The string is static.
 * No reference counterpart (bench.py reports it next to the roofline figures). */
const char *rf_render_kernel_name(rf_ctx *ctx);

/* Pixels every render launch of this process (all contexts) was made for, n * h * w each, since the
 * library was loaded.  No reference counterpart: bench.py divides the PMC totals of a profiled run by
 * it, so that per-pixel figures count the pixels really rendered (not waves x pixels per wave). */
unsigned long long rf_pixels_rendered(void);

/* 1 when the process started with REINFOCUS_POISON_ALLOC in its environment: every device / pinned-host allocation of the
 * library is then filled with 0xA5 bytes before use (csrc/rf_host.h dev_malloc) -- a debugging aid under which the GPU test
 * suite runs, so that no result can depend on what fresh or recycled memory holds.  No reference counterpart. */
int rf_allocations_poisoned(void);

/* Pixels the launches of the ctx's last rf_render_general call left to the fix-up kernel, summed over the call's launches:
 * the pixels the one-shape or the dense kernel could not decide in float32 and the literal code rendered again
 * (csrc/rf_general_one.h, rf_general_dense.h); 0 for calls the literal kernel served and before the first call.
 * No reference counterpart (tools/bench_general.py reports the share). */
unsigned rf_general_redo_pixels(rf_ctx *ctx);

/* Blocks until everything enqueued on the ctx's stream has finished. */
int rf_synchronize(rf_ctx *ctx);

/* Per-kernel timing with HIP events recorded on the ctx's own stream (bench.py's
 * roofline figure).  rf_timing(ctx, 1) enables and resets the accumulators;
 * rf_timing_read synchronizes and returns total milliseconds and launch counts. */
int rf_timing(rf_ctx *ctx, int enable);
int rf_timing_read(rf_ctx *ctx, double *render_ms, uint64_t *render_launches,
                   double *focus_ms, uint64_t *focus_launches);

/* ---- general renderer ------------------------------------
 * Renders n environments of spheres and z-aligned rectangles seen through per-environment
 * cameras, with up to 50 diffuse bounces per sample, into the ctx's frame buffer
 * (uint8[n][h][w][3]) and optionally to host_out.  As the reference does for every call
 * (render.py:115) the RNG states [0, n*h*w) are re-created from seed 0 first.
 *   cameras float64[n][19]: lower_left, horizontal, vertical, origin, u, v (3 each), lens
 *           radius -- the numpy.hstack row of camera.Cameras (camera.py:63-83)
 *   params  float32[n][most][width], types int32[n][most] (0 sphere, 1 rectangle),
 *           sizes int32[n]: world.Worlds (world.py:30-65); a sphere is {x, y, z, r, fx, fy}
 *           (sphere.py:14-19), a rectangle {x_min, x_max, y_min, y_max, z, fx, fy}
 *           (rectangle.py:12-19)
 * Replaces: render.render (graphics/render.py:88-119) and kernel device_render (:31-85). */
int rf_render_general(rf_ctx *ctx, int n, int h, int w, int spp, const double *cameras,
                      const float *params, const int32_t *types, const int32_t *sizes, int most,
                      int width, uint8_t *host_out);

/* ---- device-resident DiscreteSteps-v0 step ------------
 * The per-step numpy glue of the reference's vector environment runs on the GPU around
 * the render and focus kernels; a step uploads the actions and a pool of candidate reset
 * states and downloads observations, rewards and flags.
 * Replaces, for the environment assembled in examples/custom_environments.py:114-241:
 *   VectorEnvironment.reset / step        environments/vector_environment.py:75-164
 *   DiscreteMoveTransformer.transform     environments/state_transformer.py:248-266
 *   TimeLimitEnder | DivergingEnder       environments/episode_ender.py:106-207, :580-656
 *   Normalized(Delta([Indexed, Focus]))   environments/state_observer.py:232-292, :472-517
 *   Delta + Observation + OnTarget reward environments/episode_rewarder.py:86-155, :210-292
 *   FastCameras / FastWorlds packing      graphics/camera.py:144-179, graphics/world.py:110-123
 */
typedef struct rf_env_config {
    int n;                    /* environments */
    int n_actions;            /* <= 32 */
    double action_set[32];    /* moves of the focus plane (float64, as the reference) */
    float limit_lo, limit_hi; /* clip limits of the state */
    int max_steps;            /* TimeLimitEnder; <= 0 disables it */
    float diverge_threshold;  /* DivergingEnder threshold */
    int early_end_steps;      /* DivergingEnder early_end_steps */
    float mid[4], scale[4];   /* NormalizedObserver mid / scale (float32) */
    float reward_scale;       /* DeltaRewarder scale */
    float on_target_span;     /* OnTargetRewarder span */
    double half_width, half_height; /* FastCameras: aspect * tan(vfov/2), tan(vfov/2) */
    double tan_half_r;        /* FastWorlds: tan(radians(r_size / 2)) */
    float look_from[3], cam_u[3], cam_v[3], cam_w[3];
    double lens_radius;
    int frame_height, spp, gray_mode;
} rf_env_config;

/* Allocates the per-env device state for cfg->n environments (RNG states must already
 * cover n * frame_height^2 pixels: rf_seed first). */
int rf_env_configure(rf_ctx *ctx, const rf_env_config *cfg);

/* vector_environment.py:75-102: installs host_states float32[n][2] = [target, focus plane],
 * renders and scores every environment, returns observations float32[n][4]. */
int rf_env_reset(rf_ctx *ctx, const float *host_states, float *host_obs);

/* vector_environment.py:104-164: one step.  host_actions int32[n]; host_pool float32[n][2]
 * holds the initializer's candidate states, the r-th done environment (in index order)
 * takes row r; *host_n_reset returns how many rows were consumed.  Outputs:
 * observations float32[n][4], rewards float64[n], truncated uint8[n] (terminated is always
 * false for this environment). */
int rf_env_step(rf_ctx *ctx, const int32_t *host_actions, const float *host_pool, float *host_obs,
                double *host_rewards, uint8_t *host_truncated, int *host_n_reset);

/* The same step in two halves, for an environment sharded over several contexts / GPUs
 * (harness.ShardedVectorDiscreteSteps): which rows of the initializer's pool a shard takes depends
 * on how many environments ended in the shards before it (vector_environment.py:138-142 draws
 * done.sum() states for the done environments in index order).
 *   rf_env_step_begin   transform, enders, full render + focus, observations, rewards, flags
 *                       (vector_environment.py:124-135); *host_n_reset = k environments ended
 *   rf_env_step_end     those k environments take host_pool float32[k][2] in index order and are
 *                       rendered and scored again (vector_environment.py:137-151); observations
 *                       float32[n][4] of all environments.  host_pool may be NULL when k == 0.
 * rf_env_step == begin + end with the pool's first k rows -- on ONE context.  Several contexts that
 * share an environment range (harness.ShardedVectorDiscreteSteps) are not bit-equal to one context
 * holding all of it after the first auto-reset: each context's partial render indexes RNG states from
 * its own base, where a single context (like the reference, graphics/render.py:217 on
 * vector_environment.py:144's compacted rows) indexes the compacted set of ALL ended environments
 * from state 0 (the sharded environment's exact mode gathers them instead).
 *   rf_env_step_abort   drops an open two-phase step (after a failure on another shard): the
 *                       environments that ended stay un-reset and rf_env_reset must come next. */
int rf_env_step_begin(rf_ctx *ctx, const int32_t *host_actions, double *host_rewards, uint8_t *host_truncated,
                      int *host_n_reset);
int rf_env_step_end(rf_ctx *ctx, const float *host_pool, float *host_obs);
int rf_env_step_abort(rf_ctx *ctx);

/* The two halves a sharded environment uses by default (harness.ShardedVectorDiscreteSteps), cut where the fused
 * step (RF_ENV_BRANCH_FUSED below) allows -- BEFORE the render, because which environments end depends on their
 * counters alone (episode_ender.py:137-148, :602-607), not on what the step observes:
 *   rf_env_step_plan    transform + enders + ranking of the environments that end; *host_n_reset = k.  Cheap: no render.
 *   rf_env_step_run     the rest of the step with host_pool float32[k][2] for those k (NULL when k == 0): ONE render
 *                       launch (two-pass blocks for slots 0 .. k-1) and one focus launch where the two-pass kernel
 *                       exists, the separate launches of rf_env_step_begin / _end otherwise -- same results either way,
 *                       and the same as begin + end.  Outputs as rf_env_step.
 * rf_env_step_abort drops a planned step as it drops a begun one.  No reference counterpart
 * (vector_environment.py:104-164 is one synchronous schedule on one device). */
int rf_env_step_plan(rf_ctx *ctx, const int32_t *host_actions, int *host_n_reset);
int rf_env_step_run(rf_ctx *ctx, const float *host_pool, float *host_obs, double *host_rewards, uint8_t *host_truncated);

/* Exact mode of an environment sharded over several contexts (opt-in; harness.ShardedVectorDiscreteSteps
 * exact=True).  On one device the partial render of an auto-reset indexes RNG states from 0 over the
 * compacted rows of ALL environments that ended (vector_environment.py:144 -> state_observer.py:377-381
 * -> render.py:217), i.e. compacted row r draws from the states of environment slot r.  To reproduce
 * that, row r is rendered by the context that owns slot r, whichever context the ended environment
 * lives on, and the focus value travels back through the host:
 *   rf_env_render_states   packs host_states float32[k][2] (target, focus plane) as compacted rows
 *                          0..k-1, renders them at the environment's frame size / spp from THIS context's
 *                          RNG state 0 and scores them: host_focus float64[k].  Afterwards the context's
 *                          renderer holds that set (rf_env_render draws it), as the reference's does.
 *   rf_env_step_end_given  rf_env_step_end without the render: this context's k ended environments take
 *                          host_pool float32[k][2] in index order, and their reset observations are built
 *                          from host_focus float64[k]. */
int rf_env_render_states(rf_ctx *ctx, int k, const float *host_states, double *host_focus);
int rf_env_step_end_given(rf_ctx *ctx, const float *host_pool, const double *host_focus, float *host_obs);

/* Renders the scene set the environment uploaded last -- all n environments after a step in which
 * none ended, otherwise only the k that were reset, exactly what the reference's shared
 * FastRenderer holds at that point -- at frame_height x frame_height with spp samples, advancing
 * RNG states [0, len * frame_height^2); host_out uint8[len][frame_height][frame_height][3] may be
 * NULL (frames stay in the ctx's buffer).  rf_env_scene_len returns len.
 * Replaces: HistoryVisualizer.visualize's renderer.render(600)
 * (environments/episode_visualizer.py:197) on the renderer FocusObserver shares with it. */
int rf_env_scene_len(rf_ctx *ctx, int *n_envs);
int rf_env_render(rf_ctx *ctx, int frame_height, int spp, uint8_t *host_out);

/* TimeLimitEnder._steps and DivergingEnder._diverging_steps, int32[n] each (what the visualiser
 * prints: episode_ender.py:191-207, :646-656). */
int rf_env_get_counters(rf_ctx *ctx, int32_t *host_steps, int32_t *host_diverging);

/* How the last rf_env_step ran -- different schedules of the same arithmetic with the same results
 * (tests/test_gpu_environment.py runs each against the reference's numpy glue):
 *   RF_ENV_BRANCH_FUSED       the default for the canonical camera: ONE render launch and one focus launch per
 *                             step.  Which environments end depends on their counters alone, so they are ranked
 *                             before the render, and the blocks of slots 0 .. k-1 -- whose RNG streams the k
 *                             re-rendered frames continue (render.py:217) -- make two passes; one host
 *                             synchronisation at any size;
 *   RF_ENV_BRANCH_FUSED_GRAPH the same, replayed as one hipGraph from the second step on;
 * and, with REINFOCUS_ENV_FUSED=0 (or a camera / kernel choice without a two-pass instance), two launches each:
 *   RF_ENV_BRANCH_ONE_SYNC    whole step enqueued at once, auto-reset launch sized for all n slots,
 *                             one host synchronisation (small configurations);
 *   RF_ENV_BRANCH_GRAPH       the same, replayed as one hipGraph (from such a configuration's second
 *                             step on; REINFOCUS_ENV_GRAPH=0 disables);
 *   RF_ENV_BRANCH_COUNT_SIZED rf_env_step_begin + rf_env_step_end: one host round trip mid-step, the
 *                             auto-reset launch sized by the count (configurations whose full render
 *                             has more than REINFOCUS_ENV_ONE_SYNC_MAX = 65536 blocks, e.g. the
 *                             benchmarked 4096 x 256 x 256).
 * No reference counterpart (vector_environment.py:104-164 is one synchronous schedule); diagnostic. */
#define RF_ENV_BRANCH_NONE 0
#define RF_ENV_BRANCH_ONE_SYNC 1
#define RF_ENV_BRANCH_GRAPH 2
#define RF_ENV_BRANCH_COUNT_SIZED 3
#define RF_ENV_BRANCH_FUSED 4
#define RF_ENV_BRANCH_FUSED_GRAPH 5
int rf_env_last_step_branch(rf_ctx *ctx, int *branch);

/* Current states float32[n][2] (tests / checkpoint). */
int rf_env_get_states(rf_ctx *ctx, float *host_states);

#ifdef __cplusplus
}
#endif
#endif /* REINFOCUS_HIP_H */
