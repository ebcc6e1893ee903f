// rf_general_coop.h -- the general renderer organised for gfx950:
//
//   render_general_dense_kernel<POW2>   the dense pass: every pixel, block-cooperative rejection tails,
//                                       block-uniform bounce loop, no float64 texture code
//   render_general_fixup_kernel<POW2>   the literal per-pixel code (rf_general.h) for the pixels the dense
//                                       pass could not decide in float32, from a compacted list
//
// Why two kernels.  The literal kernel (one thread per pixel, render_pixel_general) ran at 95 VGPRs = 5
// waves per SIMD with a lane utilisation of 0.49-0.58: every rejection loop and the <= 50-bounce loop ran
// in-wave at max-over-lanes trips, and the out-of-line float64 atan2 / acos / sin of the texture
// coordinates set the register budget although about one decision in 10^3 needs them.  Two facts of the
// reference's arithmetic make the split exact:
//   * the checker colour of a hit (physics.py:47-64) only ever scales the attenuation by 0 or 1
//     (physics.py:67-92): it changes neither the path of the ray nor how many draws the pixel's stream
//     makes.  So a pixel whose colour decision is doubtful in float32 can be abandoned without harm --
//   * and re-rendered from its untouched RNG state by the literal code: the dense pass stores neither the
//     state nor (lastingly) the bytes of such a pixel and appends its index to a list; the fix-up kernel
//     renders the listed pixels exactly as the old kernel rendered every pixel.
// The float32 decisions are the ones the literal code itself takes first (sphere_red / checker_sign_general:
// same expressions, same margins), so the two kernels agree wherever the dense pass does not abstain, and
// frames and final RNG states are what the literal kernel alone produces (tests/test_gpu_general.py, no pixel
// budget).
//
// The dense pass.  256 threads = 256 consecutive pixels of one environment (lanes along x), all of them in
// lockstep through the sample loop so that every barrier is reached by every thread:
//   per sample   coordinates (2 draws) | one in-wave disc attempt, stragglers finish on packed waves (call) |
//                ray | bounces: closest hit -> two in-wave sphere attempts, stragglers on packed waves (call)
//                -> scatter; the loop ends for the block when no lane hit anything (or after 50 bounces)
//   one call     park | B1 | packed waves finish | B2 | collect    (B2 only when somebody parked)
// Which lane executes an attempt is irrelevant: a pixel's stream is advanced by exactly the same draws.
//
// SYNCHRONISATION of a call.  cnt[parity] is never reset: a wave adds (stragglers | any-hit << 16) with one
// LDS atomic, every thread keeps the value it read after the previous call of that parity (block-uniform) and
// works with the difference.  Calls alternate parity strictly (one running index for disc and sphere calls
// alike), and every call executes B1, so
//   - cnt[p] / state[p] written by the park of call c are next written by the park of call c + 2, which lies
//     behind B1 of call c + 1 -- and no wave reaches that barrier before it has finished call c (its read of
//     cnt[p] after B1, its collect reads of state[p]);
//   - words4 / words2 are written by the workers of call c + 1 only after B1 of call c + 1: same argument.
#pragma once

#include "rf_coop2.h" // lanemask, lane_in, lanes_where
#include "rf_kernels.h"

namespace rf {

struct GeneralDenseArgs {
    GeneralArgs g;
    unsigned *redo_count; // [1], zeroed before the launch
    unsigned *redo_list;  // [n * hw]: pixel indices (e * hw + p within the launch) for the fix-up kernel
};

// --- float32 texture decisions that abstain instead of falling back to float64 --------------------------
// (the quick paths of rf_general.h's sphere_red / checker_sign_general, expression by expression)
__device__ __forceinline__ bool sphere_red_quick(const float n[3], float fu, float fv, bool &doubt)
{
    float u, v;
    sphere_uv_approx(n, u, v);
    const float mu = fu * u, mv = fv * v;
    int odd_u, odd_v;
    const bool quick = safe_parity(mu, (__builtin_fabsf(fu) + __builtin_fabsf(mu) + 1.0f) * 2e-6f, odd_u) &&
                       safe_parity(mv, (__builtin_fabsf(fv) + __builtin_fabsf(mv) + 1.0f) * 2e-6f, odd_v);
    doubt = doubt || !quick;
    return odd_u == odd_v;
}

__device__ __forceinline__ int checker_sign_quick(float f, float u, bool &doubt)
{
    const float m = f * u;
    const float fl = __builtin_floorf(m);
    const float fr = m - fl;
    const float am = __builtin_fabsf(m);
    const float margin = (am > 1.0f ? am : 1.0f) * 9.5367431640625e-07f; // 2^-20
    const bool quick = am < 65536.0f && fr > margin && fr < 1.0f - margin; // false for NaN
    doubt = doubt || !quick;
    return ((int)fl & 1) ? -1 : 1;
}

__device__ __forceinline__ bool rectangle_red_quick(const float *rp, const HitRec &r, bool &doubt)
{
    const float u = (r.p[0] - rp[0]) / (rp[1] - rp[0]);
    const float v = (r.p[1] - rp[2]) / (rp[3] - rp[2]);
    return checker_sign_quick(rp[5], u, doubt) * checker_sign_quick(rp[6], v, doubt) > 0;
}

// The scene arrays (cameras, shape parameters, types, sizes) are written by the host before the launch and never by a
// kernel: read through the constant address space, a block-uniform address becomes an s_load into scalar registers.
// (Through a plain pointer the compiler has to assume that the kernel's own stores and atomics may have changed them:
// it then re-reads them after every barrier with one vector load per lane.)
template <class T>
using const_as = const __attribute__((address_space(4))) T;
template <class T>
__device__ __forceinline__ const_as<T> *as_const(const T *p)
{
    return (const_as<T> *)(unsigned long long)p;
}
constexpr int kShapeWords = 7; // parameters of a shape the kernels read (sphere: 6, rectangle: 7)

// world.hit (world.py:126-167) with the closest hit's colour decided in float32 or not at all
__device__ __forceinline__ bool world_hit_quick(const_as<float> *params, const_as<int32_t> *types, int n_shapes, int width,
                                                const float o[3], const float d[3], HitRec &rec, bool &doubt)
{
    int which = -1;
    float closest = 1000000.0f;
    for (int i = 0; i < n_shapes; ++i) {
        HitRec tmp;
        float row[kShapeWords];
#pragma unroll
        for (int k = 0; k < kShapeWords; ++k)
            row[k] = params[i * width + k];
        const bool h = types[i] == 0 ? sphere_hit(row, o, d, 0.001f, closest, tmp) : rectangle_hit(row, o, d, 0.001f, closest, tmp);
        if (h) {
            which = i;
            closest = tmp.t;
            rec = tmp;
        }
    }
    // texture coordinates / checker colour of the closest hit only, shape by shape so that the shape's parameters
    // stay scalar (the lanes of a wave mostly agree on the shape)
    for (int i = 0; i < n_shapes; ++i) {
        if (lanes_where(which == i) == 0) // wave-uniform
            continue;
        float row[kShapeWords];
#pragma unroll
        for (int k = 0; k < kShapeWords; ++k)
            row[k] = params[i * width + k];
        if (which == i) {
            if (types[i] == 0) {
                // (opaque: or the normal's texture coordinates -- independent of i -- are hoisted out of the loop and
                // computed for every hit, rectangles included)
                float n[3] = {rec.n[0], rec.n[1], rec.n[2]};
                asm volatile("" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]));
                rec.red = sphere_red_quick(n, row[4], row[5], doubt);
            } else {
                rec.red = rectangle_red_quick(row, rec, doubt);
            }
        }
    }
    return which >= 0;
}

// camera.get_ray (camera.py:307-350), general_ray of rf_general.h with the per-environment constants from the host
__device__ __forceinline__ void general_ray_dense(const_as<GeneralCamera> &cam, float p0, float p1, float s, float t, float o[3],
                                                  float d[3])
{
    const double rd0 = (double)p0 * cam.lens_radius, rd1 = (double)p1 * cam.lens_radius;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = (cam.origin0[k] + (float)(cam.u64[k] * rd0)) + (float)(cam.v64[k] * rd1);
        d[k] = ((cam.lower_left0[k] + cam.f[3 + k] * s) + cam.f[6 + k] * t) - o[k];
    }
}

// One cooperative call (see SYNCHRONISATION above).  `need`: this lane's rejection loop has not accepted yet; `flag`:
// a per-lane boolean whose block-wide OR is returned (the bounce loop's "somebody hit").  On return every lane that
// needed holds the advanced state and the accepted raw draws in w[0 .. 2 DIM).
struct GeneralCoopLds {
    uint4 state[2][kBlock];
    uint4 words4[kBlock];
    uint2 words2[kBlock];
    int cnt[2];
};

template <int DIM>
__device__ __forceinline__ bool coop_call(GeneralCoopLds &lds, int parity, int &seen0, int &seen1, lanemask need, bool any_flag,
                                          Rng &g, uint32_t *w)
{
    const int before = parity ? seen1 : seen0; // (block-uniform: scalar selects)
    const int tid = threadIdx.x;
    uint4 *const state = lds.state[parity];
    const int add = (int)__builtin_popcountll(need) | (any_flag ? 0x10000 : 0);
    int old = 0;
    if (add != 0) { // wave-uniform
        if ((tid & 63) == 0)
            old = atomicAdd(&lds.cnt[parity], add);
        old = __builtin_amdgcn_readfirstlane(old);
    }
    const int slot = ((old - before) & 0xFFFF) +
                     (int)__builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0));
    if (lane_in(need))
        state[slot] = make_uint4(g.a_lo, g.a_hi, g.b_lo, g.b_hi);
    __syncthreads(); // B1
    const int now = __builtin_amdgcn_readfirstlane(lds.cnt[parity]);
    const int delta = now - before;
    seen0 = parity ? seen0 : now;
    seen1 = parity ? now : seen1;
    const int total = delta & 0xFFFF; // <= kBlock: one pixel per thread
    if (total != 0) { // block-uniform
        if (tid < total) {
            __builtin_amdgcn_s_setprio(1); // serial work the rest of the block waits for
            const uint4 ps = state[tid];
            Rng wg{ps.x, ps.y, ps.z, ps.w};
            uint32_t ww[6] = {0, 0, 0, 0, 0, 0};
            if (DIM == 2) {
                while (!disc_attempt(wg, ww)) {
                }
            } else {
                while (!sphere_attempt(wg, ww)) {
                }
            }
            state[tid] = make_uint4(wg.a_lo, wg.a_hi, wg.b_lo, wg.b_hi);
            lds.words4[tid] = make_uint4(ww[0], ww[1], ww[2], ww[3]);
            if (DIM == 3)
                lds.words2[tid] = make_uint2(ww[4], ww[5]);
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads(); // B2
        if (lane_in(need)) {
            const uint4 ps = state[slot];
            g = Rng{ps.x, ps.y, ps.z, ps.w};
            const uint4 w4 = lds.words4[slot];
            w[0] = w4.x; w[1] = w4.y; w[2] = w4.z; w[3] = w4.w;
            if (DIM == 3) {
                const uint2 w2 = lds.words2[slot];
                w[4] = w2.x; w[5] = w2.y;
            }
        }
    }
    return (delta >> 16) != 0;
}

// waves per SIMD the register allocator is held to: 70 registers (7 waves) without the float64 pixel coordinates of
// frames that are not powers of two, 78 (6 waves) with them
#ifndef RF_GENERAL_DENSE_OCC
#define RF_GENERAL_DENSE_OCC (POW2 ? 7 : 6)
#endif
template <bool POW2, bool COOP>
__global__ __launch_bounds__(kBlock, RF_GENERAL_DENSE_OCC) void render_general_dense_kernel(GeneralDenseArgs da)
{
    const GeneralArgs &a = da.g;
    __shared__ GeneralCoopLds lds;
    __shared__ uint32_t stage[kBlock * 3 / 4]; // the block's 256 pixels x 3 B, stored as 192 coalesced dwords
    const int tid = threadIdx.x;
    if (tid < 2)
        lds.cnt[tid] = 0;
    __syncthreads();
    int seen0 = 0, seen1 = 0;
    int call = 0; // running index of the cooperative calls (block-uniform); parity = call & 1

    const int e = blockIdx.y;
    const int p0 = blockIdx.x * kBlock;
    const int p = p0 + tid;
    const bool live = p < a.hw;
    const size_t pix = (size_t)e * a.hw + (live ? p : 0);
    const int y = p / a.w, x = p - y * a.w;

    Rng g = rng_load(0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull); // dead lanes: any state
    if (live) {
        const ulonglong2 st = a.states[pix];
        g = rng_load(st.x, st.y);
    }
    const_as<GeneralCamera> &cam = *as_const(a.cameras + e);
    const_as<float> *const params = as_const(a.params + ((size_t)e * a.most) * a.width);
    const_as<int32_t> *const types = as_const(a.types + (size_t)e * a.most);
    const int n_shapes = *as_const(a.sizes + e);
    const GeneralFrame frame = general_frame(a.h, a.w);

    // lane predicates as 64-bit masks in scalar registers (rf_coop2.h: a bool that crosses control flow costs a
    // vector register and slow-path compares)
    const lanemask live_m = lanes_where(p < a.hw);
    lanemask doubt_m = 0;
    float cr = 0.0f, cg = 0.0f, cb = 0.0f;
    for (int k = 0; k < a.spp; ++k) {
        float s, t;
        general_coords<POW2>(g, x, y, frame, s, t);
        float q0, q1, q2;
        float o[3], d[3];
        lanemask need_m;
        {
            // (the raw-draw words are written by the first attempt of every lane that reads them: rf_coop2.h any_u32)
            uint32_t w[4] = {any_u32(), any_u32(), any_u32(), any_u32()};
            // (every lane makes the attempt: a dead lane's state is never stored)
            need_m = live_m & ~lanes_where(disc_attempt_sq(g, w) < 1.0f);
            if (COOP) {
                coop_call<2>(lds, call++ & 1, seen0, seen1, need_m, false, g, w);
            } else {
#ifdef XP_TAILCAP // TIMING EXPERIMENT (wrong frames)
                for (int trip_ = 0; trip_ < XP_TAILCAP && need_m != 0; ++trip_) {
#else
                while (need_m != 0) { // wave-uniform
#endif
                    float sq = 2.0f;
                    if (lane_in(need_m))
                        sq = disc_attempt_sq(g, w);
                    need_m &= ~lanes_where(sq < 1.0f);
                }
            }
            disc_finish(w, q0, q1);
        }
        general_ray_dense(cam, q0, q1, s, t, o, d);

        // physics.py:67-92: a hit multiplies the attenuation by (1, 0, 0) or (0, 1, 0), so all that matters is
        // whether the ray has met red, green, or anything at all
        lanemask active_m = live_m, missed_m = 0, red_m = 0, green_m = 0;
        for (int bounce = 0; bounce < kMaxBounces; ++bounce) { // block-uniform trip count
            HitRec rec;
            rec.red = false;
            lanemask hit_m = 0;
            if (active_m != 0) { // wave-uniform
                bool hit = false, doubt = false;
                if (lane_in(active_m))
                    hit = world_hit_quick(params, types, n_shapes, a.width, o, d, rec, doubt);
                hit_m = active_m & lanes_where(hit);
                doubt_m |= lanes_where(doubt);
                missed_m |= active_m & ~hit_m;
                active_m = hit_m;
                const lanemask r = lanes_where(rec.red);
                red_m |= hit_m & r;
                green_m |= hit_m & ~r;
            }
            uint32_t w[6] = {any_u32(), any_u32(), any_u32(), any_u32(), any_u32(), any_u32()};
            need_m = hit_m;
            if (COOP) {
#pragma unroll
                for (int trip = 0; trip < 2; ++trip) {
                    if (need_m != 0) { // wave-uniform
                        float sq = 2.0f;
                        if (lane_in(need_m))
                            sq = sphere_attempt_sq(g, w);
                        need_m &= ~lanes_where(sq < 1.0f);
                    }
                }
                if (!coop_call<3>(lds, call++ & 1, seen0, seen1, need_m, hit_m != 0, g, w))
                    break; // nobody in the block hit anything: no draws were made, every ray has left the scene
            } else {
                if (hit_m == 0) // nobody in the wave hit anything
                    break;
#ifdef XP_TAILCAP // TIMING EXPERIMENT (wrong frames)
                for (int trip_ = 0; trip_ < XP_TAILCAP + 1 && need_m != 0; ++trip_) {
#else
                while (need_m != 0) { // wave-uniform
#endif
                    float sq = 2.0f;
                    if (lane_in(need_m))
                        sq = sphere_attempt_sq(g, w);
                    need_m &= ~lanes_where(sq < 1.0f);
                }
            }
            if (lane_in(hit_m)) { // physics.py:81-87: the scattered ray
                sphere_finish(w, q0, q1, q2);
                o[0] = rec.p[0];
                o[1] = rec.p[1];
                o[2] = rec.p[2];
                d[0] = add2(rec.n[0], q0);
                d[1] = add2(rec.n[1], q1);
                d[2] = add2(rec.n[2], q2);
            }
        }
        // physics.py:137-145: the sky along the ray that left; a ray still bouncing after 50 hits is black
        Colour c{0.0f, 0.0f, 0.0f};
        if (lane_in(missed_m))
            c = sky_colour(d, lane_in(green_m) ? 0.0f : 1.0f, lane_in(red_m) ? 0.0f : 1.0f,
                           lane_in(red_m | green_m) ? 0.0f : 1.0f);
        cr = add2(cr, c.r);
        cg = add2(cg, c.g);
        cb = add2(cb, c.b);
    }
    const bool doubt = lane_in(doubt_m);

    uint8_t r8 = 0, g8 = 0, b8 = 0;
    if (live) {
        if (doubt) { // abstain: state untouched, pixel listed for render_general_fixup_kernel
            da.redo_list[atomicAdd(da.redo_count, 1u)] = (unsigned)pix;
        } else {
            a.states[pix] = make_ulonglong2(rng_s0(g), rng_s1(g));
            r8 = (uint8_t)(cr * a.scale);
            g8 = (uint8_t)(cg * a.scale);
            b8 = (uint8_t)(cb * a.scale);
        }
    }
    // a full block whose first byte is dword-aligned goes through LDS; anything else stores bytes
    // (the ADDRESS decides: a.frames is the chunk's base, see render_general_kernel)
    const size_t first_byte = ((size_t)e * a.hw + p0) * 3;
    const bool staged = p0 + kBlock <= a.hw && (reinterpret_cast<uintptr_t>(a.frames + first_byte) & 3) == 0; // block-uniform
    if (staged) {
        uint8_t *sb = reinterpret_cast<uint8_t *>(stage);
        sb[tid * 3 + 0] = r8;
        sb[tid * 3 + 1] = g8;
        sb[tid * 3 + 2] = b8;
        __syncthreads();
        if (tid < kBlock * 3 / 4)
            reinterpret_cast<uint32_t *>(a.frames + first_byte)[tid] = stage[tid];
    } else if (live) {
        uint8_t *dst = a.frames + pix * 3;
        dst[0] = r8;
        dst[1] = g8;
        dst[2] = b8;
    }
}

// The listed pixels, literally (rf_general.h render_pixel_general: float64 texture coordinates where float32 cannot
// decide).  Runs after the dense pass on the same stream; grid-stride over the list, whose length it reads itself.
template <bool POW2>
__global__ __launch_bounds__(kBlock, RF_GENERAL_OCC) void render_general_fixup_kernel(GeneralDenseArgs da)
{
    const GeneralArgs &a = da.g;
    const unsigned total = *da.redo_count;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
        const unsigned pix = da.redo_list[i];
        const int e = (int)(pix / (unsigned)a.hw), p = (int)(pix - (unsigned)e * (unsigned)a.hw);
        const int y = p / a.w, x = p - y * a.w;
        const ulonglong2 st = a.states[pix];
        Rng g = rng_load(st.x, st.y);
        float cr, cg, cb;
        render_pixel_general<POW2>(g, x, y, a.h, a.w, a.spp, a.cameras[e], a.params + ((size_t)e * a.most) * a.width,
                                   a.types + (size_t)e * a.most, a.sizes[e], a.width, cr, cg, cb);
        a.states[pix] = make_ulonglong2(rng_s0(g), rng_s1(g));
        uint8_t *dst = a.frames + (size_t)pix * 3;
        dst[0] = (uint8_t)(cr * a.scale);
        dst[1] = (uint8_t)(cg * a.scale);
        dst[2] = (uint8_t)(cb * a.scale);
    }
}

} // namespace rf
