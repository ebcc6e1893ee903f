// rf_host.h -- what the translation units of libreinfocus_hip.so share on the host side: the context, the error
// plumbing and the helpers one unit exports to the others.
//
//   rf_abi_ctx.hip      contexts, RNG states (seed_kernel), scene upload, timing, device table
//   rf_abi_render.hip   which render kernel a launch takes (render_form, pick_tile_layout, launch_render), rf_render,
//                       the frame buffer, the focus measure (launch_focus, rf_focus, rf_step)
//   rf_abi_general.hip  rf_render_general
//   rf_abi_env.hip      the device-resident environment step and its schedules
//
// Every kernel is defined in exactly one unit (the one that launches it); the units share only host functions.
#pragma once
#include <cstring>

#include "../../include/reinfocus_hip.h"

#include <hip/hip_runtime.h>

#include <stdint.h>

#include <string>
#include <utility>
#include <vector>

#include "rf_env_types.h" // EnvConfig / EnvState (plain structs of device pointers)
#include "rf_math.h" // CamStatic, CheckerTable

typedef std::pair<hipEvent_t, hipEvent_t> EventPair;

struct rf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;

    ulonglong2 *d_states = nullptr;
    uint64_t n_states = 0;
    ulonglong2 *d_mats = nullptr;
    int *d_zero = nullptr; // a device int that stays 0: the second-pass count of a two-pass kernel launched for one pass
    // rf_render_general re-creates seed-0 states for every call, as the reference does (render.py:115): a copy of
    // the freshly seeded array turns all but the first seeding of a size into a device-to-device copy
    ulonglong2 *d_seed_cache = nullptr;
    uint64_t seed_cache_n = 0;

    float *d_cam = nullptr;
    float *d_rect = nullptr;
    int scene_n = 0;
    int scene_cap = 0;
    rf::CamStatic cs{};
    bool axis = false;
    bool coop = true; // cooperative rejection tails (REINFOCUS_RENDER_COOP=0: render_kernel for every launch)
    bool auto_form = true; // launches of few blocks take the kernel without cooperative tails, mid-size launches the
                           // wave-cooperative one (render_form; REINFOCUS_RENDER_SETS=3: three pixels per thread with
                           // block-cooperative tails for launches of every size)
    int wave_sets = 0;     // REINFOCUS_RENDER_SETS=w1 / w2 / w3: render_kernel_wave<K> for launches of every size
                           // (1 ... 3), REINFOCUS_RENDER_SETS=1: render_kernel (0: by launch size)
    bool one_px = false;
    bool strip = true; // a frame's last w % 64 <= 48 columns as tiles of 48 x 16 (REINFOCUS_RENDER_STRIP=0: one tile shape)
    double hit_fraction = 0.658; // target width / frame width of the current scene (tan 10 / tan 15 deg by default)
    bool general_one = true; // general renderer: cooperative kernel for one-shape worlds (REINFOCUS_GENERAL_ONE=0: never)
    bool general_one_always = false; // ... for launches of every size (REINFOCUS_GENERAL_ONE=1; default: large launches only)
    bool general_dense = true; // general renderer: the float32 kernel with abstentions for worlds of up to three shapes
                               // (REINFOCUS_GENERAL_DENSE=0: never)
    bool general_dense_always = false; // ... for launches of every size (REINFOCUS_GENERAL_DENSE=1; default: large launches only)

    uint8_t *d_frames = nullptr;
    size_t frames_cap = 0;
    int fn = 0, fh = 0, fw = 0;
    uint8_t *d_frames2 = nullptr; // the fused environment step: the step's frames of the environments whose slot renders twice
    size_t frames2_cap = 0;

    unsigned long long *d_sums = nullptr;
    double *d_var = nullptr;
    int focus_cap = 0;
    int focus_choice = 0; // REINFOCUS_FOCUS_KERNEL=quad / byte: the round-2 kernel where its rows fit into LDS / the
                          // byte-per-thread kernel, instead of focus_kernel_roll (A/B runs and tests)
    int focus_band = 0;   // REINFOCUS_FOCUS_BAND=r: rows per band of focus_kernel_roll (default: by launch size)

    rf::CheckerTable tab{};

    // device-resident env step (rf_env_*)
    bool env_ready = false;
    rf::EnvConfig env_cfg{};
    rf::EnvState env{};
    rf_env_config env_host{};
    void *env_block = nullptr; // one allocation holding every EnvState array
    int *d_actions = nullptr;
    float *d_pool = nullptr;
    // small configurations replay their (host-independent) step as one hipGraph
    bool env_graph_enabled = true; // REINFOCUS_ENV_GRAPH=0 disables
    hipGraphExec_t env_graph = nullptr;
    uint8_t *h_stage = nullptr;    // pinned: the host image of the io block (EnvIo) of the replayed step
    size_t h_stage_bytes = 0;
    uint64_t env_steps = 0;
    bool env_axis = false;
    bool env_last_partial = false; // that set is the compacted one of an auto-reset (cam_dyn2 / rect2)
    int env_scene_len = 0; // environments of the scene set uploaded last: n after a full render, k after a partial one
    int env_pending = -1; // >= 0: rf_env_step_begin ran and that many environments wait for rf_env_step_end
    bool env_planned = false; // the open step is rf_env_step_plan's (rf_env_step_run finishes it)
    bool env_graph_fail_once = false; // REINFOCUS_ENV_GRAPH_FAIL=1 (tests): the first instantiation "fails"
    int env_last_branch = RF_ENV_BRANCH_NONE; // rf_env_last_step_branch
    bool env_needs_reset = false; // rf_env_step_abort dropped a half-finished step
    bool env_fused = true; // the step's two renders and two focus measures as one launch each (REINFOCUS_ENV_FUSED=0: the
                           // three schedules of separate launches)
    long env_one_sync_max = 65536; // blocks of a full render up to which rf_env_step runs without the mid-step round
                                   // trip (REINFOCUS_ENV_ONE_SYNC_MAX; tests set 0 to reach the count-sized branch at small sizes)
    const char *render_kernel = "none"; // the render kernel the last launch used (rf_render_kernel_name)
    void *general_scratch = nullptr;    // scene arrays of rf_render_general (grown on demand)
    unsigned general_redo_last = 0;     // pixels the launches of the last rf_render_general left to the fix-up kernel
    size_t general_scratch_bytes = 0;

    bool timing = false;
    std::vector<EventPair> ev_render, ev_focus;
    double render_ms = 0.0, focus_ms = 0.0;
    uint64_t render_n = 0, focus_n = 0;
};

namespace rfh {

// the calling thread's last error message (rf_last_error)
void set_err(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define RF_HIP(expr)                                                                           \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            rfh::set_err("%s: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
            return _e == hipErrorOutOfMemory ? RF_ERR_OOM : RF_ERR_HIP;                        \
        }                                                                                      \
    } while (0)

#define RF_REQUIRE(cond, ...)                                                                  \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            rfh::set_err(__VA_ARGS__);                                                         \
            return RF_ERR_INVALID;                                                             \
        }                                                                                      \
    } while (0)

inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// every pixel any render kernel of this process was launched for (rf_pixels_rendered)
void count_pixels(unsigned long long pixels);

int drain_events(std::vector<EventPair> &evs, double &ms, uint64_t &count);

// HIP events around a kernel's launches while rf_timing is on (the launch's own stream)
struct Timed {
    rf_ctx *ctx;
    std::vector<EventPair> *evs;
    hipEvent_t a = nullptr, b = nullptr;
    Timed(rf_ctx *c, std::vector<EventPair> *e) : ctx(c), evs(e)
    {
        if (ctx->timing) {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            (void)hipEventRecord(a, ctx->stream);
        }
    }
    ~Timed()
    {
        if (ctx->timing) {
            (void)hipEventRecord(b, ctx->stream);
            evs->push_back(EventPair(a, b));
        }
    }
};

// The captured env step (rf_env_step) holds device pointers and kernel arguments by value: any
// call that may reallocate a buffer or change the scene / configuration drops it.
void drop_env_graph(rf_ctx *ctx);

// states [first, first + count) of the context's array := the states of `seed` at indices first_state_index ...
// (rf_seed is this for the whole array)
int seed_range(rf_ctx *ctx, uint64_t first, uint64_t count, uint64_t seed, uint64_t first_state_index);

// Splits the lens radius for rf_math.h lens_offset and decides whether the float32 form is exact for it
void lens_split(rf::CamStatic &cs);
// ... the same question without the 60 ms proof for a radius the process has seen in fewer than 64 calls (rf_render_general)
bool lens_exact_if_known(double radius);

int ensure_frames(rf_ctx *ctx, int n, int h, int w);
int ensure_frames2(rf_ctx *ctx, int n, int h, int w);

// the second pass of a fused environment step's render (RenderArgs::count2 ...)
struct SecondPass {
    const int *count;
    const float *cam, *rect;
};
// 3 = three pixels per thread with block-cooperative tails (render_kernel_coop2 and its strip form), 0 = one pixel per
// thread without them (render_kernel), 20 + K = render_kernel_wave<K> (K pixels per thread, wave-cooperative tails)
int render_form(const rf_ctx *ctx, int n, int h, int w);
// enqueues the render of n envs whose scene arrays are cam / rect (device pointers)
int launch_render(rf_ctx *ctx, int n, int h, int w, int spp, const float *cam, const float *rect, bool axis,
                  bool count_pixels = true, const SecondPass *second = nullptr);
int ensure_focus(rf_ctx *ctx, int n);
int launch_focus(rf_ctx *ctx, int n, int h, int w, int gray_mode, const float *skip_rect = nullptr, bool in_env_step = false,
                 const int *fused_count = nullptr);

// hipMalloc / hipHostMalloc.  With REINFOCUS_POISON_ALLOC in the environment (tests/conftest.py sets it for the whole GPU suite)
// every allocation is filled with 0xA5 bytes first: nothing may depend on what fresh -- or recycled -- memory happens to hold.
inline bool poison_allocations()
{
    static const bool poison = getenv("REINFOCUS_POISON_ALLOC") != nullptr;
    return poison;
}
inline hipError_t dev_malloc(void **p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess && bytes && poison_allocations()) {
        e = hipMemset(*p, 0xA5, bytes);
        if (e == hipSuccess)
            e = hipDeviceSynchronize(); // (the ctx's stream does not wait for the null stream)
    }
    return e;
}
inline hipError_t host_malloc(void **p, size_t bytes)
{
    hipError_t e = hipHostMalloc(p, bytes, hipHostMallocDefault);
    if (e == hipSuccess && bytes && poison_allocations())
        memset(*p, 0xA5, bytes);
    return e;
}

} // namespace rfh
