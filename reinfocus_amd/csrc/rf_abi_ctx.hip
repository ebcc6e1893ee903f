// rf_abi_ctx.hip -- C ABI of libreinfocus_hip.so (include/reinfocus_hip.h): contexts, RNG states, scene upload,
// timing, the device table.  One rf_ctx = one renderer on one GPU: it owns the RNG states, the scene parameters, the
// frame buffer and the focus partials, all on one HIP stream.  There is no CPU fallback: if no device is usable
// rf_create fails and says so.
#include "rf_host.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <new>

#include "rf_jump.h"
#include "rf_seed.h"

using namespace rfh;

namespace {

thread_local std::string g_err;
std::atomic<unsigned long long> g_pixels_rendered{0};

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

} // namespace

namespace rfh {

void set_err(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

void count_pixels(unsigned long long pixels) { g_pixels_rendered += pixels; }

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

// The captured env step (rf_env_step) holds device pointers and kernel arguments by value: any
// call that may reallocate a buffer or change the scene / configuration drops it.
void drop_env_graph(rf_ctx *ctx)
{
    if (ctx->env_graph)
        (void)hipGraphExecDestroy(ctx->env_graph);
    ctx->env_graph = nullptr;
    ctx->env_steps = 0;
}

// Whether the float32 form of rf_math.h lens_offset reproduces float32(float64(p) * radius) for EVERY float32 p a disc
// coordinate can be (multiples of 2^-24 in [-1, 0), of 2^-23 in [0, 1]): tried one by one, ~60 ms on one core.
static bool lens_f32_exact(double radius, float hi, float lo)
{
    bool exact = true;
    for (int k = 0; exact && k <= (1 << 24); ++k) {
        const float neg = (float)((double)k * (1.0 / 16777216.0) - 1.0); // k 2^-24 - 1, exact
        const float pos = (float)((double)k * (1.0 / 8388608.0));          // k 2^-23 (k <= 2^23)
        exact = fmaf(neg, hi, neg * lo) == (float)((double)neg * radius) &&
                (k > (1 << 23) || fmaf(pos, hi, pos * lo) == (float)((double)pos * radius));
    }
    return exact;
}

// What the process knows about lens radii.  The reference's aperture 0.1 (camera.py:106: FastCameras and make_gpu_camera's
// default) is entered as proven -- the CPU test suite runs the proof on the same arithmetic (test_lens_offset_float32_form)
// -- so that no process pays for it; everything else is proven on demand, OUTSIDE the lock (two threads may prove the
// same radius at the same time: same answer), so that contexts on other threads never wait 60 ms for a mutex.
namespace {
struct LensRecord {
    int exact = -1;    // -1: not proven yet
    unsigned asked = 0; // calls of the general renderer that would have liked to know
};
std::mutex lens_guard;
std::map<double, LensRecord> lens_known = {{0.05, LensRecord{1, 0}}};
} // namespace

// Splits the lens radius for rf_math.h lens_offset and decides whether the float32 form is exact for it (the fast path's
// question: ONE radius per context, used by every sample of every render -- always worth the proof).  Remembered per radius
// for the life of the process.
void lens_split(rf::CamStatic &cs)
{
    const double radius = cs.lens_radius;
    cs.lens_hi = (float)radius;
    cs.lens_lo = (float)(radius - (double)cs.lens_hi);
    cs.lens_f32 = 0;
    if (!(radius == radius) || radius - radius != 0.0) // NaN / infinity: literal path
        return;
    {
        std::lock_guard<std::mutex> lock(lens_guard);
        auto it = lens_known.find(radius);
        if (it != lens_known.end() && it->second.exact >= 0) {
            cs.lens_f32 = it->second.exact;
            return;
        }
    }
    const bool exact = lens_f32_exact(radius, cs.lens_hi, cs.lens_lo);
    std::lock_guard<std::mutex> lock(lens_guard);
    lens_known[radius].exact = exact ? 1 : 0;
    cs.lens_f32 = exact ? 1 : 0;
}

// The general renderer's question (rf_render_general, per camera of a launch): is the float32 lens offset KNOWN to be exact
// for this radius?  Its SIMPLE kernel instances are ~3 % faster than the ones with the reference's float64 lens products --
// of a render of about a millisecond -- and a proof costs 60 ms, so a radius is only proven once it has come back in
// kLensProveAfter calls (an aperture sweep, one new radius per call, never pays; REINFOCUS_LENS_PROVE_AFTER=n changes the
// count, tests set 1).  Until then: false, the float64 instances -- the same frames either way.
bool lens_exact_if_known(double radius)
{
    static const unsigned prove_after = [] {
        const char *v = getenv("REINFOCUS_LENS_PROVE_AFTER");
        const long n = v ? strtol(v, nullptr, 10) : 0;
        return (unsigned)(n >= 1 ? n : 64);
    }();
    if (!(radius == radius) || radius - radius != 0.0)
        return false;
    {
        std::lock_guard<std::mutex> lock(lens_guard);
        LensRecord &record = lens_known[radius];
        if (record.exact >= 0)
            return record.exact == 1;
        if (++record.asked < prove_after)
            return false;
    }
    rf::CamStatic probe{};
    probe.lens_radius = radius;
    lens_split(probe);
    return probe.lens_f32 != 0;
}

int seed_range(rf_ctx *ctx, uint64_t first, uint64_t count, uint64_t seed, uint64_t first_state_index)
{
    RF_REQUIRE(first + count <= ctx->n_states, "seed_range: [%llu, %llu) exceeds %llu states", (unsigned long long)first,
               (unsigned long long)(first + count), (unsigned long long)ctx->n_states);
    const rf::S128 s0 = rf::h_splitmix(seed);
    const uint64_t per_wave = 64ull * rf::kSeedRun;
    const uint64_t waves = (count + per_wave - 1) / per_wave;
    const uint64_t blocks = (waves * 64 + rf::kBlock - 1) / rf::kBlock;
    RF_REQUIRE(blocks < (1ull << 31), "rf_seed: too many states for one launch");
    hipLaunchKernelGGL(rf::seed_kernel, dim3((unsigned)blocks), dim3(rf::kBlock), 0, ctx->stream, ctx->d_states + first,
                       (unsigned long long)count, (unsigned long long)first_state_index, make_ulonglong2(s0.s0, s0.s1),
                       ctx->d_mats);
    RF_HIP(hipGetLastError());
    return RF_OK;
}

} // namespace rfh

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
    {
        ctx->auto_form = v[0] != '3'; // (3: three pixels per thread with cooperative tails for launches of every size)
        if (v[0] == 'w' && v[1] >= '1' && v[1] <= '3')
            ctx->wave_sets = v[1] - '0';
        ctx->one_px = v[0] == '1';
    }
    if (const char *v = getenv("REINFOCUS_GENERAL_DENSE"))
        ctx->general_dense = v[0] != '0', ctx->general_dense_always = v[0] == '1';
    if (const char *v = getenv("REINFOCUS_ENV_FUSED"))
        ctx->env_fused = strcmp(v, "0") != 0;
    if (const char *v = getenv("REINFOCUS_ENV_GRAPH"))
        ctx->env_graph_enabled = v[0] != '0';
    if (const char *v = getenv("REINFOCUS_GENERAL_ONE"))
        ctx->general_one = v[0] != '0', ctx->general_one_always = v[0] == '1';
    if (const char *v = getenv("REINFOCUS_ENV_GRAPH_FAIL"))
        ctx->env_graph_fail_once = v[0] == '1';
    if (const char *v = getenv("REINFOCUS_FOCUS_KERNEL"))
        ctx->focus_choice = !strcmp(v, "quad") ? 1 : !strcmp(v, "byte") ? 2 : 0;
    if (const char *v = getenv("REINFOCUS_FOCUS_BAND"))
        ctx->focus_band = atoi(v);
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
        he = dev_malloc((void **)&ctx->d_mats, sizeof(rf::Mat128) * rf::kSeedMats);
    if (he == hipSuccess)
        he = hipMemcpy(ctx->d_mats, tables.data(), sizeof(rf::Mat128) * rf::kSeedMats,
                       hipMemcpyHostToDevice);
    if (he == hipSuccess)
        he = dev_malloc((void **)&ctx->d_zero, 256);
    if (he == hipSuccess)
        he = hipMemset(ctx->d_zero, 0, 256);
    if (he == hipSuccess)
        he = hipDeviceSynchronize(); // (a memset on the null stream, which the ctx's non-blocking stream does not wait for)
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
        RF_HIP(dev_malloc((void **)&ctx->d_states, n_states * sizeof(ulonglong2)));
        ctx->n_states = n_states;
    }
    return rfh::seed_range(ctx, 0, n_states, seed, first_state_index);
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
        RF_HIP(dev_malloc((void **)&ctx->d_cam, (size_t)n * 9 * sizeof(float)));
        RF_HIP(dev_malloc((void **)&ctx->d_rect, (size_t)n * 2 * sizeof(float)));
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

unsigned long long rf_pixels_rendered(void) { return g_pixels_rendered.load(); }
int rf_allocations_poisoned(void) { return poison_allocations() ? 1 : 0; }

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

} // extern "C"
