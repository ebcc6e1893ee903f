// rf_abi.hip -- C ABI of libreinfocus_hip.so (see include/reinfocus_hip.h).
//
// Host-side driver of the gfx950 kernels in rf_kernels.h.  One rf_ctx = one
// renderer on one GPU: it owns the RNG states, the scene parameters, the frame
// buffer and the focus partials, all on one HIP stream.  There is no CPU fallback:
// if no device is usable rf_create fails and says so.
#include "../../include/reinfocus_hip.h"

#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "rf_env.h"
#include "rf_jump.h"
#include "rf_kernels.h"
#include "rf_coop2.h"
#include "rf_general_one.h"

namespace {

thread_local std::string g_err;
std::atomic<unsigned long long> g_pixels_rendered{0}; // every pixel any render kernel of this process was launched for

void set_err(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

#define RF_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            set_err("%s: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
            return _e == hipErrorOutOfMemory ? RF_ERR_OOM : RF_ERR_HIP;                   \
        }                                                                                 \
    } while (0)

#define RF_REQUIRE(cond, ...)                                                             \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            set_err(__VA_ARGS__);                                                         \
            return RF_ERR_INVALID;                                                        \
        }                                                                                 \
    } while (0)

typedef std::pair<hipEvent_t, hipEvent_t> EventPair;

} // namespace

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
    bool coop = true; // block-cooperative sphere loop (REINFOCUS_RENDER_COOP=0 disables)
    bool two_sets = true; // several pixels per thread in the cooperative kernel (REINFOCUS_RENDER_SETS=1 disables)
    bool auto_sets = true; // ... except for launches of few blocks, which take the kernel without cooperative tails (few_blocks
                           // below; REINFOCUS_RENDER_SETS=3 / =1: always three / one pixels per thread with them)
    bool strip = true; // a frame's last w % 64 <= 48 columns as tiles of 48 x 16 (REINFOCUS_RENDER_STRIP=0: one tile shape)
    int tile_layout = -1; // REINFOCUS_TILE_LAYOUT=0..3 forces one (experiments), -1: pick_tile_layout
    double hit_fraction = 0.658; // target width / frame width of the current scene (tan 10 / tan 15 deg by default)
    bool focus_quad = true; // 4-pixels-per-thread focus kernel (REINFOCUS_FOCUS_QUAD=0 disables)
    bool general_one = true; // general renderer: cooperative kernel for one-shape worlds (REINFOCUS_GENERAL_ONE=0: literal kernel)
    bool general_one_always = false; // ... for launches of every size (REINFOCUS_GENERAL_ONE=1; default: large launches only)

    uint8_t *d_frames = nullptr;
    size_t frames_cap = 0;
    int fn = 0, fh = 0, fw = 0;
    uint8_t *d_frames2 = nullptr; // the fused environment step: the step's frames of the environments whose slot renders twice
    size_t frames2_cap = 0;

    unsigned long long *d_sums = nullptr;
    double *d_var = nullptr;
    int focus_cap = 0;

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
                           // three schedules of separate launches below)
    long env_one_sync_max = 65536; // blocks of a full render up to which rf_env_step runs without the mid-step round
                                   // trip (REINFOCUS_ENV_ONE_SYNC_MAX; tests set 0 to reach the count-sized branch at small sizes)
    const char *render_kernel = "none"; // the render kernel the last launch used (rf_render_kernel_name)
    void *general_scratch = nullptr;    // scene arrays of rf_render_general (grown on demand)
    unsigned general_redo_last = 0;     // pixels the last launch of the last rf_render_general left to the fix-up kernel
    size_t general_scratch_bytes = 0;

    bool timing = false;
    std::vector<EventPair> ev_render, ev_focus;
    double render_ms = 0.0, focus_ms = 0.0;
    uint64_t render_n = 0, focus_n = 0;
};

namespace {

// physics.py:58-62 with uf = 32: sign of sin(fl64(fl64(32*pi) * k/32)) for the 33
// texture coordinates where 32*u is an integer (see rf_math.h checker_sign).
rf::CheckerTable make_checker_table()
{
    rf::CheckerTable t{0};
    for (int k = 1; k <= 32; ++k) {
        const float u = (float)k / 32.0f;
        const double si = ((double)32.0f * 3.14159265358979323846) * (double)u;
        if (sin(si) < 0.0)
            t.neg_mask |= (1ull << k);
    }
    return t;
}

int drain_events(std::vector<EventPair> &evs, double &ms, uint64_t &count)
{
    for (EventPair &p : evs) {
        float t = 0.0f;
        RF_HIP(hipEventSynchronize(p.second));
        RF_HIP(hipEventElapsedTime(&t, p.first, p.second));
        ms += (double)t;
        count += 1;
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    evs.clear();
    return RF_OK;
}

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

// The captured env step (rf_env_step) holds device pointers and kernel arguments by value: any
// call that may reallocate a buffer or change the scene / configuration drops it.
void drop_env_graph(rf_ctx *ctx)
{
    if (ctx->env_graph)
        (void)hipGraphExecDestroy(ctx->env_graph);
    ctx->env_graph = nullptr;
    ctx->env_steps = 0;
}

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// Tile layout of render_kernel_coop2 for a frame size (index into the table in launch_render):
//   0: 128 x 6 (4 waves of 32 x 2 side by side)   1: 64 x 12 (2 x 2 such waves)
//   2: 256 x 3 (4 waves of 64 x 1)                3: 128 x 6 (2 x 2 waves of 64 x 1)
//   4: 64 x 12 (4 waves of 16 x 4 side by side)   5: 32 x 24 (2 x 2 such waves; experiments only)
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

// Splits the lens radius for rf_math.h lens_offset and decides -- by trying every float32 a disc
// coordinate can be (multiples of 2^-24 in [-1, 0), of 2^-23 in [0, 1]) -- whether the float32
// form reproduces float32(float64(p) * radius) for this radius.  ~60 ms on one core; remembered
// per radius for the life of the process (the reference's FastCameras always uses 0.05).
void lens_split(rf::CamStatic &cs)
{
    static std::mutex guard;
    static std::map<double, bool> known;
    const double radius = cs.lens_radius;
    cs.lens_hi = (float)radius;
    cs.lens_lo = (float)(radius - (double)cs.lens_hi);
    cs.lens_f32 = 0;
    if (!(radius == radius) || radius - radius != 0.0) // NaN / infinity: literal path
        return;
    std::lock_guard<std::mutex> lock(guard);
    auto it = known.find(radius);
    if (it == known.end()) {
        bool exact = true;
        const float hi = cs.lens_hi, lo = cs.lens_lo;
        for (int k = 0; exact && k <= (1 << 24); ++k) {
            const float neg = (float)((double)k * (1.0 / 16777216.0) - 1.0); // k 2^-24 - 1, exact
            const float pos = (float)((double)k * (1.0 / 8388608.0));          // k 2^-23 (k <= 2^23)
            exact = fmaf(neg, hi, neg * lo) == (float)((double)neg * radius) &&
                    (k > (1 << 23) || fmaf(pos, hi, pos * lo) == (float)((double)pos * radius));
        }
        it = known.emplace(radius, exact).first;
    }
    cs.lens_f32 = it->second ? 1 : 0;
}

int ensure_frames(rf_ctx *ctx, int n, int h, int w)
{
    const size_t need = (size_t)n * h * w * 3 + 64; // + slack for dword tails
    if (need > ctx->frames_cap) {
        if (ctx->d_frames)
            RF_HIP(hipFree(ctx->d_frames));
        ctx->d_frames = nullptr;
        ctx->frames_cap = 0;
        RF_HIP(hipMalloc((void **)&ctx->d_frames, need));
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
        RF_HIP(hipMalloc((void **)&ctx->d_frames2, need));
        ctx->frames2_cap = need;
    }
    return RF_OK;
}

} // namespace

extern "C" {

const char *rf_last_error(void) { return g_err.c_str(); }

int rf_abi_version(void) { return 1; }

int rf_device_count(int *count)
{
    RF_REQUIRE(count != nullptr, "rf_device_count: count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return RF_OK;
}

int rf_device_info(int device, char *bus_id, int len, int *numa_node)
{
    RF_REQUIRE(bus_id != nullptr && numa_node != nullptr && len >= 16, "rf_device_info: bus_id[>= 16] and numa_node wanted");
    int n = 0;
    RF_HIP(hipGetDeviceCount(&n));
    RF_REQUIRE(device >= 0 && device < n, "rf_device_info: device %d of %d", device, n);
    RF_HIP(hipDeviceGetPCIBusId(bus_id, len, device));
    for (char *c = bus_id; *c; ++c) // sysfs spells the id in lower case
        *c = (*c >= 'A' && *c <= 'F') ? (char)(*c - 'A' + 'a') : *c;
    *numa_node = -1;
    char path[128];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus_id);
    if (FILE *f = fopen(path, "r")) {
        int node = -1;
        if (fscanf(f, "%d", &node) == 1)
            *numa_node = node;
        fclose(f);
    }
    return RF_OK;
}

int rf_create(int device, rf_ctx **out)
{
    RF_REQUIRE(out != nullptr, "rf_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        set_err("rf_create: no HIP device is visible (%s); libreinfocus_hip has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return RF_ERR_NO_DEVICE;
    }
    RF_REQUIRE(device >= 0 && device < count, "rf_create: device %d out of range [0,%d)", device, count);
    RF_HIP(hipSetDevice(device));

    rf_ctx *ctx = new (std::nothrow) rf_ctx();
    RF_REQUIRE(ctx != nullptr, "rf_create: out of host memory");
    ctx->device = device;
    ctx->tab = make_checker_table();
    if (const char *v = getenv("REINFOCUS_RENDER_COOP"))
        ctx->coop = v[0] != '0';
    if (const char *v = getenv("REINFOCUS_RENDER_STRIP"))
        ctx->strip = strcmp(v, "0") != 0;
    if (const char *v = getenv("REINFOCUS_RENDER_SETS"))
        ctx->two_sets = v[0] != '1', ctx->auto_sets = false; // (1: always one pixel per thread, anything else: always three)
    if (const char *v = getenv("REINFOCUS_ENV_FUSED"))
        ctx->env_fused = strcmp(v, "0") != 0;
    if (const char *v = getenv("REINFOCUS_ENV_GRAPH"))
        ctx->env_graph_enabled = v[0] != '0';
    if (const char *v = getenv("REINFOCUS_TILE_LAYOUT"))
        ctx->tile_layout = (v[0] >= '0' && v[0] <= '5') ? v[0] - '0' : -1;
    if (const char *v = getenv("REINFOCUS_GENERAL_ONE"))
        ctx->general_one = v[0] != '0', ctx->general_one_always = v[0] == '1';
    if (const char *v = getenv("REINFOCUS_FOCUS_QUAD"))
        ctx->focus_quad = v[0] != '0';
    if (const char *v = getenv("REINFOCUS_ENV_GRAPH_FAIL"))
        ctx->env_graph_fail_once = v[0] == '1';
    if (const char *v = getenv("REINFOCUS_ENV_ONE_SYNC_MAX")) {
        char *end = nullptr;
        const long limit = strtol(v, &end, 10);
        if (end != v && limit >= 0)
            ctx->env_one_sync_max = limit;
    }

    std::vector<rf::Mat128> tables;
    if (!rf::h_build_jump_tables(rf::kSeedMats, tables)) {
        delete ctx;
        set_err("rf_create: GF(2) jump matrix disagrees with numba's jump polynomial");
        return RF_ERR_INVALID;
    }
    hipError_t he = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (he == hipSuccess)
        he = hipMalloc((void **)&ctx->d_mats, sizeof(rf::Mat128) * rf::kSeedMats);
    if (he == hipSuccess)
        he = hipMemcpy(ctx->d_mats, tables.data(), sizeof(rf::Mat128) * rf::kSeedMats,
                       hipMemcpyHostToDevice);
    if (he == hipSuccess)
        he = hipMalloc((void **)&ctx->d_zero, 256);
    if (he == hipSuccess)
        he = hipMemset(ctx->d_zero, 0, 256);
    if (he != hipSuccess) {
        set_err("rf_create: %s", hipGetErrorString(he));
        rf_destroy(ctx);
        return he == hipErrorOutOfMemory ? RF_ERR_OOM : RF_ERR_HIP;
    }
    *out = ctx;
    return RF_OK;
}

int rf_destroy(rf_ctx *ctx)
{
    if (!ctx)
        return RF_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    for (EventPair &p : ctx->ev_render) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    for (EventPair &p : ctx->ev_focus) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    if (ctx->d_states) (void)hipFree(ctx->d_states);
    if (ctx->d_mats) (void)hipFree(ctx->d_mats);
    if (ctx->d_zero) (void)hipFree(ctx->d_zero);
    if (ctx->d_seed_cache) (void)hipFree(ctx->d_seed_cache);
    if (ctx->d_cam) (void)hipFree(ctx->d_cam);
    if (ctx->d_rect) (void)hipFree(ctx->d_rect);
    if (ctx->d_frames) (void)hipFree(ctx->d_frames);
    if (ctx->d_frames2) (void)hipFree(ctx->d_frames2);
    if (ctx->d_sums) (void)hipFree(ctx->d_sums);
    if (ctx->d_var) (void)hipFree(ctx->d_var);
    if (ctx->env_graph) (void)hipGraphExecDestroy(ctx->env_graph);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->env_block) (void)hipFree(ctx->env_block);
    if (ctx->general_scratch) (void)hipFree(ctx->general_scratch);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RF_OK;
}

int rf_seed(rf_ctx *ctx, uint64_t n_states, uint64_t seed, uint64_t first_state_index)
{
    RF_REQUIRE(ctx != nullptr, "rf_seed: ctx is NULL");
    RF_REQUIRE(n_states > 0, "rf_seed: n_states must be positive");
    RF_REQUIRE(first_state_index + n_states < (1ull << rf::kSeedMats),
               "rf_seed: state index exceeds 2^%d", rf::kSeedMats);
    RF_HIP(hipSetDevice(ctx->device));
    drop_env_graph(ctx);
    if (n_states != ctx->n_states) {
        RF_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->d_states)
            RF_HIP(hipFree(ctx->d_states));
        ctx->d_states = nullptr;
        ctx->n_states = 0;
        RF_HIP(hipMalloc((void **)&ctx->d_states, n_states * sizeof(ulonglong2)));
        ctx->n_states = n_states;
    }
    const rf::S128 s0 = rf::h_splitmix(seed);
    const uint64_t per_wave = 64ull * rf::kSeedRun;
    const uint64_t waves = (n_states + per_wave - 1) / per_wave;
    const uint64_t blocks = (waves * 64 + rf::kBlock - 1) / rf::kBlock;
    RF_REQUIRE(blocks < (1ull << 31), "rf_seed: too many states for one launch");
    hipLaunchKernelGGL(rf::seed_kernel, dim3((unsigned)blocks), dim3(rf::kBlock), 0, ctx->stream,
                       ctx->d_states, (unsigned long long)n_states,
                       (unsigned long long)first_state_index, make_ulonglong2(s0.s0, s0.s1),
                       ctx->d_mats);
    RF_HIP(hipGetLastError());
    return RF_OK;
}

int rf_num_states(rf_ctx *ctx, uint64_t *n_states)
{
    RF_REQUIRE(ctx != nullptr && n_states != nullptr, "rf_num_states: NULL argument");
    *n_states = ctx->n_states;
    return RF_OK;
}

int rf_get_states(rf_ctx *ctx, uint64_t first, uint64_t count, uint64_t *host_out)
{
    RF_REQUIRE(ctx != nullptr && host_out != nullptr, "rf_get_states: NULL argument");
    RF_REQUIRE(first + count <= ctx->n_states, "rf_get_states: range [%llu,%llu) exceeds %llu states",
               (unsigned long long)first, (unsigned long long)(first + count),
               (unsigned long long)ctx->n_states);
    RF_HIP(hipSetDevice(ctx->device));
    RF_HIP(hipMemcpyAsync(host_out, ctx->d_states + first, count * sizeof(ulonglong2),
                          hipMemcpyDeviceToHost, ctx->stream));
    RF_HIP(hipStreamSynchronize(ctx->stream));
    return RF_OK;
}

int rf_set_states(rf_ctx *ctx, uint64_t first, uint64_t count, const uint64_t *host_in)
{
    RF_REQUIRE(ctx != nullptr && host_in != nullptr, "rf_set_states: NULL argument");
    RF_REQUIRE(first + count <= ctx->n_states, "rf_set_states: range exceeds %llu states",
               (unsigned long long)ctx->n_states);
    RF_HIP(hipSetDevice(ctx->device));
    drop_env_graph(ctx);
    RF_HIP(hipMemcpyAsync(ctx->d_states + first, host_in, count * sizeof(ulonglong2),
                          hipMemcpyHostToDevice, ctx->stream));
    RF_HIP(hipStreamSynchronize(ctx->stream));
    return RF_OK;
}

int rf_set_scene(rf_ctx *ctx, int n, const float *cam_dyn, const float *rect,
                 const float origin[3], const float u[3], const float v[3], double lens_radius)
{
    RF_REQUIRE(ctx != nullptr, "rf_set_scene: ctx is NULL");
    RF_REQUIRE(n > 0, "rf_set_scene: n must be positive");
    RF_REQUIRE(cam_dyn && rect && origin && u && v, "rf_set_scene: NULL argument");
    RF_HIP(hipSetDevice(ctx->device));
    drop_env_graph(ctx);
    if (n > ctx->scene_cap) {
        RF_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->d_cam) RF_HIP(hipFree(ctx->d_cam));
        if (ctx->d_rect) RF_HIP(hipFree(ctx->d_rect));
        ctx->d_cam = ctx->d_rect = nullptr;
        ctx->scene_cap = 0;
        RF_HIP(hipMalloc((void **)&ctx->d_cam, (size_t)n * 9 * sizeof(float)));
        RF_HIP(hipMalloc((void **)&ctx->d_rect, (size_t)n * 2 * sizeof(float)));
        ctx->scene_cap = n;
    }
    RF_HIP(hipMemcpyAsync(ctx->d_cam, cam_dyn, (size_t)n * 9 * sizeof(float), hipMemcpyHostToDevice,
                          ctx->stream));
    RF_HIP(hipMemcpyAsync(ctx->d_rect, rect, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice,
                          ctx->stream));
    RF_HIP(hipStreamSynchronize(ctx->stream)); // host buffers are free again on return

    ctx->cs = rf::CamStatic{origin[0], origin[1], origin[2], u[0], u[1], u[2],
                            v[0],      v[1],      v[2],      lens_radius, 0.0f, 0.0f, 0};
    lens_split(ctx->cs);
    // canonical frame of FastCameras() (camera.py:100-130): enables the AXIS kernel
    bool axis = origin[0] == 0.0f && origin[1] == 0.0f && origin[2] == 0.0f && u[0] == 1.0f &&
                u[1] == 0.0f && u[2] == 0.0f && v[0] == 0.0f && v[1] == 1.0f && v[2] == 0.0f;
    for (int e = 0; axis && e < n; ++e) {
        const float *c = cam_dyn + (size_t)e * 9;
        // horizontal = (hx, +0, +0), vertical = (+0, vy, +0)
        axis = c[4] == 0.0f && c[5] == 0.0f && c[6] == 0.0f && c[8] == 0.0f && !signbit(c[4]) &&
               !signbit(c[5]) && !signbit(c[6]) && !signbit(c[8]);
    }
    ctx->axis = axis;
    ctx->scene_n = n;
    // width of the target in the frame (for the tile layout): |p.x| <= half at the rectangle's
    // plane <=> s within half * |ll.z| / (|z| * horizontal.x) of the centre
    {
        const double half = rect[0], z = rect[1], llz = cam_dyn[2], hx = cam_dyn[3];
        const double f = (z != 0.0 && hx != 0.0) ? fabs(2.0 * half * llz / (z * hx)) : 0.658;
        ctx->hit_fraction = (f == f && f > 0.0 && f < 1.0) ? f : (f >= 1.0 ? 1.0 : 0.658);
    }
    return RF_OK;
}

namespace {

// the second pass of a fused environment step's render (RenderArgs::count2 ...)
struct SecondPass {
    const int *count;
    const float *cam, *rect;
};

// Launches of few blocks -- the reference's own default is ONE environment of 300 x 300 pixels at 100 samples: 119
// blocks of three pixels per thread on 256 CUs, each running its samples one after the other -- are bound by the
// latency of a sample, not by issue slots: with one or two waves per SIMD nothing hides the cooperative tails' barriers,
// and the kernel without them (render_kernel<AXIS, POW2>: one pixel per thread, rejection loops inside the wave) is the
// fastest form -- 1 x 300^2 x 100: 687 us per step with three pixels per thread, 419 with one and cooperative tails, 289
// without them; 4 environments 863 / 646 / 563.  Three pixels per thread win from about 650 000 pixels per launch on (10
// environments of 300^2 or of 256^2, 40 of 128^2: profiles/r04_ab.txt section 18), at any number of samples (the
// kernel has a two-pass form of its own, so the fused step's few launches serve both).
bool few_blocks(uint64_t n, uint64_t h, uint64_t w, int spp) { return n * h * w <= 650000; }

// which render kernel a launch takes: 3 = three pixels per thread with cooperative tails (render_kernel_coop2 and its
// strip form), 1 = one pixel per thread with them (render_kernel_coop), 0 = without them (render_kernel)
int render_form(const rf_ctx *ctx, int n, int h, int w, int spp)
{
    if (!ctx->coop || (ctx->auto_sets && few_blocks((uint64_t)n, (uint64_t)h, (uint64_t)w, spp)))
        return 0;
    return ctx->two_sets ? 3 : 1;
}

bool fused_step_possible(const rf_ctx *ctx)
{
    const rf_env_config &h = ctx->env_host;
    return ctx->env_fused && ctx->env_axis && render_form(ctx, h.n, h.frame_height, h.frame_height, h.spp) != 1;
}

// enqueues the render of n envs whose scene arrays are cam / rect (device pointers)
int launch_render(rf_ctx *ctx, int n, int h, int w, int spp, const float *cam, const float *rect, bool axis,
                  bool count_pixels = true, const SecondPass *second = nullptr)
{
    int rc = ensure_frames(ctx, n, h, w);
    if (rc == RF_OK && second)
        rc = ensure_frames2(ctx, n, h, w);
    if (rc != RF_OK)
        return rc;
    const int form = render_form(ctx, n, h, w, spp);
    RF_REQUIRE(!second || (axis && form != 1), "launch_render: no two-pass instance of this kernel");
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
    a.inv_w = 1.0f / (float)w;
    a.inv_h = 1.0f / (float)h;
    a.rw64 = 1.0 / (double)w;
    a.rh64 = 1.0 / (double)h;
    a.w64 = (double)w;
    a.h64 = (double)h;
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
            const dim3 tiles(((w + rf::kTileW - 1) / rf::kTileW) * ((h + rf::kTileH - 1) / rf::kTileH), ne);
            // A block's tile is WX waves of WW x 64/WW pixels side by side, 4/WX down, kSets sets:
            //   A 128 x 6  (WX 4, WW 32)   B 64 x 12 (2, 32)   C 256 x 3 (4, 64)   D 128 x 6 (2, 64)
            // see pick_tile_layout.
            const int layout = ctx->tile_layout >= 0 ? ctx->tile_layout : pick_tile_layout(h, w, ctx->hit_fraction);
            static const int kLayoutWX[6] = {4, 2, 4, 2, 4, 2};
            static const int kLayoutWW[6] = {32, 32, 64, 64, 16, 16};
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
            if (axis && form == 3 && ctx->strip && ctx->tile_layout < 0 && !pow2 && w > 128 && rem > 0 &&
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
        case 4: RF_LAUNCH2_ONE(P, L, 4, 16); break;                                                        \
        default: RF_LAUNCH2_ONE(P, L, 2, 16); break;                                                       \
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
            else if (axis && form == 1 && pow2) {
                hipLaunchKernelGGL((rf::render_kernel_coop<true>), tiles, block, 0, ctx->stream, b);
                ctx->render_kernel = "render_kernel_coop<true>";
            } else if (axis && form == 1) {
                hipLaunchKernelGGL((rf::render_kernel_coop<false>), tiles, block, 0, ctx->stream, b);
                ctx->render_kernel = "render_kernel_coop<false>";
            } else if (axis && pow2 && second) {
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
        g_pixels_rendered += (unsigned long long)n * (unsigned long long)a.hw;
    if (ctx->ev_render.size() > 512)
        return drain_events(ctx->ev_render, ctx->render_ms, ctx->render_n);
    return RF_OK;
}

} // namespace

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

namespace {

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
    RF_HIP(hipMalloc((void **)&ctx->d_sums, (size_t)n * 2 * sizeof(unsigned long long)));
    RF_HIP(hipMalloc((void **)&ctx->d_var, (size_t)n * sizeof(double)));
    ctx->focus_cap = n;
    ctx->env.sums = ctx->d_sums;
    return RF_OK;
}

// enqueues the focus measure of the first n frames: sums into ctx->d_sums, variances into ctx->d_var (device).
// in_env_step: the sums were zeroed by the environment kernel before (env_pre_kernel / env_reset_kernel) and the
// variance is taken from them by the one after (env_post_kernel / env_reset_post_kernel, the same expression as
// focus_finalize): no memset and no finalize launch -- two nodes less per focus measure of a replayed step.
// fused_count != null: both measures of a fused environment step as one launch of 2 n rows (FocusArgs::count2)
int launch_focus(rf_ctx *ctx, int n, int h, int w, int gray_mode, const float *skip_rect = nullptr, bool in_env_step = false,
                 const int *fused_count = nullptr)
{
    // widths that are a multiple of 4 (and >= 4): four pixels per thread, 32-row bands
    const size_t lds_quad = (((size_t)(2 * rf::kBandQ + 6) * w) + 15) & ~(size_t)15;
    const bool quad = (w & 3) == 0 && w >= 4 && lds_quad <= 64 * 1024 && ctx->focus_quad;
    const int band = quad ? rf::kBandQ : rf::kBand;
    const size_t lds = quad ? lds_quad : ((((size_t)(2 * rf::kBand + 6) * w) + 15) & ~(size_t)15);
    RF_REQUIRE(lds <= 64 * 1024, "rf_focus: frame width %d needs %zu B of LDS (max 65536)", w, lds);
    int rc = ensure_focus(ctx, n);
    if (rc != RF_OK)
        return rc;
    if (!in_env_step)
        RF_HIP(hipMemsetAsync(ctx->d_sums, 0, (size_t)n * 2 * sizeof(unsigned long long), ctx->stream));
    {
        Timed timed(ctx, &ctx->ev_focus);
        const int gx = (h + band - 1) / band;
        const int rows = fused_count ? 2 * n : n;
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
            if (quad)
                hipLaunchKernelGGL(rf::focus_kernel_quad, dim3(gx, ne), dim3(rf::kBlock), lds, ctx->stream, a);
            else
                hipLaunchKernelGGL(rf::focus_kernel, dim3(gx, ne), dim3(rf::kBlock), lds, ctx->stream, a);
        }
        if (!in_env_step)
            hipLaunchKernelGGL(rf::focus_finalize, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_sums,
                               ctx->d_var, n, (unsigned long long)h * (unsigned long long)w);
    }
    RF_HIP(hipGetLastError());
    return RF_OK;
}

} // namespace

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

unsigned long long rf_pixels_rendered(void) { return g_pixels_rendered.load(); }
unsigned rf_general_redo_pixels(rf_ctx *ctx) { return ctx ? ctx->general_redo_last : 0u; }

int rf_synchronize(rf_ctx *ctx)
{
    RF_REQUIRE(ctx != nullptr, "rf_synchronize: ctx is NULL");
    RF_HIP(hipSetDevice(ctx->device));
    drop_env_graph(ctx);
    RF_HIP(hipStreamSynchronize(ctx->stream));
    return RF_OK;
}

int rf_timing(rf_ctx *ctx, int enable)
{
    RF_REQUIRE(ctx != nullptr, "rf_timing: ctx is NULL");
    RF_HIP(hipSetDevice(ctx->device));
    drop_env_graph(ctx);
    RF_HIP(hipStreamSynchronize(ctx->stream));
    double dummy_ms = 0.0;
    uint64_t dummy_n = 0;
    int rc = drain_events(ctx->ev_render, dummy_ms, dummy_n);
    if (rc == RF_OK)
        rc = drain_events(ctx->ev_focus, dummy_ms, dummy_n);
    ctx->render_ms = ctx->focus_ms = 0.0;
    ctx->render_n = ctx->focus_n = 0;
    ctx->timing = enable != 0;
    return rc;
}

int rf_timing_read(rf_ctx *ctx, double *render_ms, uint64_t *render_launches, double *focus_ms,
                   uint64_t *focus_launches)
{
    RF_REQUIRE(ctx != nullptr, "rf_timing_read: ctx is NULL");
    RF_HIP(hipSetDevice(ctx->device));
    RF_HIP(hipStreamSynchronize(ctx->stream));
    int rc = drain_events(ctx->ev_render, ctx->render_ms, ctx->render_n);
    if (rc == RF_OK)
        rc = drain_events(ctx->ev_focus, ctx->focus_ms, ctx->focus_n);
    if (render_ms) *render_ms = ctx->render_ms;
    if (render_launches) *render_launches = ctx->render_n;
    if (focus_ms) *focus_ms = ctx->focus_ms;
    if (focus_launches) *focus_launches = ctx->focus_n;
    return rc;
}

/* ---- general renderer ---------------------------------------------------------------------- */

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

/* ---- device-resident env step ------------------------------------------------------------ */

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
    RF_HIP(hipMalloc(&ctx->env_block, off));
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

namespace {

bool env_one_sync(const rf_ctx *ctx)
{
    const int n = ctx->env_host.n, fh = ctx->env_host.frame_height;
    const long tiles = (long)((fh + rf::kTileW - 1) / rf::kTileW) * ((fh + rf::kTileH2 - 1) / rf::kTileH2);
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
                RF_HIP(hipHostMalloc((void **)&ctx->h_stage, bytes, hipHostMallocDefault));
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
                    g_pixels_rendered += (unsigned long long)(n + k) * (unsigned long long)h.frame_height * h.frame_height;
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
        g_pixels_rendered += (unsigned long long)(n + k) * (unsigned long long)h.frame_height * h.frame_height;
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
    hipLaunchKernelGGL(rf::env_reset_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->env_cfg, ctx->env,
                       (const float *)nullptr, rf::kEnvResetRank, (const int *)ctx->d_actions);
    RF_HIP(hipGetLastError());
    int k = 0;
    RF_HIP(hipMemcpyAsync(&k, ctx->env.done_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    RF_HIP(hipStreamSynchronize(ctx->stream));
    ctx->env_pending = k;
    ctx->env_planned = true;
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
    g_pixels_rendered += (unsigned long long)(n + k) * (unsigned long long)fh * fh;
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
