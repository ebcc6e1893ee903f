// rf_general_coop.h -- the general renderer organised like the fast path:
//
//   render_general_coop_kernel<POW2>    the dense pass: every pixel, three pixels per thread, block-cooperative
//                                       rejection tails (rf_coop2.h's machinery), block-uniform bounce loop, float32
//                                       texture decisions only, the scene in scalar registers
//   render_general_fixup_kernel<POW2>   the literal per-pixel code (rf_general.h) for the pixels the dense pass could
//                                       not decide in float32, from a compacted list
//
// Why two kernels.  The literal kernel (one thread per pixel, render_pixel_general) runs at 95 VGPRs with every
// rejection loop and the <= 50-bounce loop in-wave at max-over-lanes trips (lane utilisation 0.49-0.58): on
// one-rectangle scenes 37 % of its time is the sparse ends of those loops (profiles/r04_ab.txt).  Two facts of the
// reference's arithmetic make a dense pass without float64 texture code exact:
//   * the checker colour of a hit (physics.py:47-64) only ever scales the attenuation by 0 or 1 (physics.py:67-92): it
//     changes neither the path of the ray nor how many draws the pixel's stream makes.  So a pixel whose colour
//     decision is doubtful in float32 can be abandoned without harm --
//   * and re-rendered from its untouched RNG state by the literal code: the dense pass stores neither the state nor
//     (lastingly) the bytes of such a pixel and appends its index to a list; the fix-up kernel renders the listed
//     pixels exactly as the literal kernel renders every pixel.
// The float32 decisions are the ones the literal code itself takes first (sphere_red / checker_sign_general: same
// expressions, same margins), so the two agree wherever the dense pass does not abstain, and frames and final RNG
// states are what the literal kernel alone produces (tests/test_gpu_general.py, no pixel budget).
//
// The dense pass.  256 threads x kSets = 768 consecutive pixels of one environment per block; all threads in lockstep
// through the sample loop so that every barrier is reached by every thread:
//   per sample   coordinates + one in-wave disc attempt per set | ONE cooperative call for the disc stragglers of all
//                sets | rays | bounces: closest hit per set, one in-wave sphere attempt per set, ONE cooperative call
//                for the sphere stragglers of all sets, which also carries the block's "somebody hit" vote | scatter;
//                the bounce loop ends for the block when no lane hit anything (or after 50 bounces)
// Which lane executes an attempt is irrelevant: a pixel's stream is advanced by exactly the same draws.
// SYNCHRONISATION: every call is coop_finish2m<.., FENCED = true> with the counter (and the vote word) alternating with
// a running call index: B1 of the next call orders the empty-list exit, B4 everything else (rf_coop2.h, top).
#pragma once

#include "rf_coop2.h"
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
// (GeneralCamera::u64 ...: loop invariants the kernel would otherwise keep in 18 vector registers)
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

#ifndef RF_GENERAL_SETS
#define RF_GENERAL_SETS 2
#endif
constexpr int kGenSets = RF_GENERAL_SETS; // pixels per thread of the cooperative general kernel
#ifndef RF_GENERAL_COOP_OCC
#define RF_GENERAL_COOP_OCC 5 // waves per SIMD the register allocator is held to
#endif
template <bool POW2>
__global__ __launch_bounds__(kBlock2, RF_GENERAL_COOP_OCC) void render_general_coop_kernel(GeneralDenseArgs da)
{
    const GeneralArgs &a = da.g;
    __shared__ CoopLds2 lds;
    __shared__ int votes[2];
    // the frame staging buffer (kGenSets * 768 B) reuses the words4 array once the sample loop is over
    static_assert(sizeof(lds.words4) >= (size_t)kGenSets * kBlock2 * 3, "stage does not fit");
    uint32_t *const stage = reinterpret_cast<uint32_t *>(lds.words4);
    __shared__ float lds_colour[kGenSets][3][kBlock2]; // the colour sums live in LDS (one read-modify-write per sample and channel)
    const int tid = threadIdx.x;
    if (tid < 2) {
        lds.cnt[tid] = 0;
        votes[tid] = 0;
    }
    if (tid == 2)
        lds.cnt2 = 0;
    int voted[2] = {0, 0}; // votes[i] as this thread saw it last (block-uniform)
    int call = 0;          // running index of the cooperative calls (block-uniform)

    const int e = blockIdx.y;
    const int p0 = blockIdx.x * (kBlock2 * kGenSets);
    const_as<GeneralCamera> &cam = *as_const(a.cameras + e);
    const_as<float> *const params = as_const(a.params + ((size_t)e * a.most) * a.width);
    const_as<int32_t> *const types = as_const(a.types + (size_t)e * a.most);
    const int n_shapes = *as_const(a.sizes + e);
    const GeneralFrame frame = general_frame(a.h, a.w);

    // pixel of set j: p0 + j * 256 + tid; re-derived inside the loops from an index the compiler cannot see through
    // (rf_coop2.h: the loop invariants are what the register allocator spills)
    auto opaque = [](int v) {
        asm volatile("" : "+v"(v));
        return v;
    };
    auto pixel_of = [&](int t, int j) { return p0 + j * kBlock2 + t; };
    Rng g[kGenSets];
    lanemask live_m[kGenSets], doubt_m[kGenSets];
#pragma unroll
    for (int j = 0; j < kGenSets; ++j) {
        const int p = pixel_of(tid, j);
        live_m[j] = lanes_where(p < a.hw);
        doubt_m[j] = 0;
        g[j] = rng_load(0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull); // dead lanes: any state
        if (p < a.hw) {
            const ulonglong2 st = a.states[(size_t)e * a.hw + p];
            g[j] = rng_load(st.x, st.y);
        }
        lds_colour[j][0][tid] = lds_colour[j][1][tid] = lds_colour[j][2][tid] = 0.0f;
    }
    __syncthreads();

    for (int k = 0; k < a.spp; ++k) {
        uint32_t w[kGenSets][6];
        float o[kGenSets][3], d[kGenSets][3];
        lanemask need_m[kGenSets];
        {
            float s[kGenSets], t[kGenSets];
#pragma unroll
            for (int j = 0; j < kGenSets; ++j) {
                const int p = pixel_of(opaque(tid), j);
                const int y = p / a.w, x = p - y * a.w;
                general_coords<POW2>(g[j], x, y, frame, s[j], t[j]);
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    w[j][i] = any_u32();
                // (every lane makes the attempt: a dead lane's state is never stored)
                need_m[j] = live_m[j] & ~lanes_where(disc_attempt_sq(g[j], w[j]) < 1.0f);
            }
            coop_finish2m<2, true, true, kGenSets>(lds, 0, &lds.cnt[call & 1], need_m, g, w, tid);
            ++call;
#pragma unroll
            for (int j = 0; j < kGenSets; ++j) {
                float q0, q1;
                disc_finish(w[j], q0, q1);
                general_ray_dense(cam, q0, q1, s[j], t[j], o[j], d[j]);
            }
        }

        // physics.py:67-92: a hit multiplies the attenuation by (1, 0, 0) or (0, 1, 0), so all that matters is whether
        // the ray has met red, green, or anything at all
        lanemask active_m[kGenSets], missed_m[kGenSets], red_m[kGenSets], green_m[kGenSets];
#pragma unroll
        for (int j = 0; j < kGenSets; ++j) {
            active_m[j] = live_m[j];
            missed_m[j] = red_m[j] = green_m[j] = 0;
        }
        for (int bounce = 0; bounce < kMaxBounces; ++bounce) { // block-uniform trip count
            lanemask hit_m[kGenSets];
            lanemask any_hit = 0;
#pragma unroll
            for (int j = 0; j < kGenSets; ++j) {
                hit_m[j] = 0;
                if (active_m[j] != 0) { // wave-uniform
                    HitRec rec;
                    rec.red = false;
                    bool hit = false, doubt = false;
                    if (lane_in(active_m[j])) {
                        hit = world_hit_quick(params, types, n_shapes, a.width, o[j], d[j], rec, doubt);
                        if (hit) { // physics.py:81-87: the scattered ray starts at the hit; its direction is n + q
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                o[j][i] = rec.p[i];
                                d[j][i] = rec.n[i];
                            }
                        }
                    }
                    hit_m[j] = active_m[j] & lanes_where(hit);
                    doubt_m[j] |= lanes_where(doubt);
                    missed_m[j] |= active_m[j] & ~hit_m[j];
                    active_m[j] = hit_m[j];
                    const lanemask r = lanes_where(rec.red);
                    red_m[j] |= hit_m[j] & r;
                    green_m[j] |= hit_m[j] & ~r;
                }
                any_hit |= hit_m[j];
                need_m[j] = hit_m[j];
                if (need_m[j] != 0) { // one in-wave attempt; the rest on packed waves
                    float sq = 2.0f;
                    if (lane_in(need_m[j]))
                        sq = sphere_attempt_sq(g[j], w[j]);
                    asm volatile("" : "+v"(sq));
                    need_m[j] &= ~lanes_where(sq < 1.0f);
                }
            }
            int seen = 0;
            coop_finish2m<3, true, true, kGenSets>(lds, 1, &lds.cnt[call & 1], need_m, g, w, tid, &votes[call & 1], any_hit != 0, &seen);
            const bool somebody_hit = seen != voted[call & 1];
            voted[call & 1] = seen;
            ++call;
            if (!somebody_hit)
                break; // nobody in the block hit anything: no draws were made, every ray has left the scene
#pragma unroll
            for (int j = 0; j < kGenSets; ++j) {
                if (lane_in(hit_m[j])) {
                    float q0, q1, q2;
                    sphere_finish(w[j], q0, q1, q2);
                    d[j][0] = add2(d[j][0], q0);
                    d[j][1] = add2(d[j][1], q1);
                    d[j][2] = add2(d[j][2], q2);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kGenSets; ++j) {
            // physics.py:137-145: the sky along the ray that left; a ray still bouncing after 50 hits is black
            Colour c{0.0f, 0.0f, 0.0f};
            if (lane_in(missed_m[j]))
                c = sky_colour(d[j], lane_in(green_m[j]) ? 0.0f : 1.0f, lane_in(red_m[j]) ? 0.0f : 1.0f,
                               lane_in(red_m[j] | green_m[j]) ? 0.0f : 1.0f);
            lds_colour[j][0][tid] = add2(lds_colour[j][0][tid], c.r);
            lds_colour[j][1][tid] = add2(lds_colour[j][1][tid], c.g);
            lds_colour[j][2][tid] = add2(lds_colour[j][2][tid], c.b);
        }
    }
    __syncthreads(); // the cooperative arrays are dead from here on: words4 becomes the stage

    // a full block whose first byte is dword-aligned goes through LDS; anything else stores bytes (the ADDRESS decides:
    // a.frames is the chunk's base, see render_general_kernel)
    const size_t first_byte = ((size_t)e * a.hw + p0) * 3;
    const bool staged = p0 + kBlock2 * kGenSets <= a.hw && (reinterpret_cast<uintptr_t>(a.frames + first_byte) & 3) == 0; // block-uniform
    uint8_t *sb = reinterpret_cast<uint8_t *>(stage);
#pragma unroll
    for (int j = 0; j < kGenSets; ++j) {
        const int p = pixel_of(opaque(tid), j);
        const size_t pix = (size_t)e * a.hw + p;
        uint8_t r8 = 0, g8 = 0, b8 = 0;
        if (p < a.hw) {
            if (lane_in(doubt_m[j])) { // abstain: state untouched, pixel listed for render_general_fixup_kernel
                da.redo_list[atomicAdd(da.redo_count, 1u)] = (unsigned)pix;
            } else {
                a.states[pix] = make_ulonglong2(rng_s0(g[j]), rng_s1(g[j]));
                r8 = (uint8_t)(lds_colour[j][0][tid] * a.scale);
                g8 = (uint8_t)(lds_colour[j][1][tid] * a.scale);
                b8 = (uint8_t)(lds_colour[j][2][tid] * a.scale);
            }
        }
        if (staged) {
            const int slot = j * kBlock2 + tid;
            sb[slot * 3 + 0] = r8;
            sb[slot * 3 + 1] = g8;
            sb[slot * 3 + 2] = b8;
        } else if (p < a.hw) {
            uint8_t *dst = a.frames + pix * 3;
            dst[0] = r8;
            dst[1] = g8;
            dst[2] = b8;
        }
    }
    if (staged) {
        __syncthreads();
        for (int i = tid; i < kBlock2 * kGenSets * 3 / 4; i += kBlock2)
            reinterpret_cast<uint32_t *>(a.frames + first_byte)[i] = stage[i];
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
