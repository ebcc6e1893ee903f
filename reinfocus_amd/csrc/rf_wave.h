// rf_wave.h -- render_kernel_wave<POW2, LENS, K, TWO>: the render kernel of launches between "a few blocks" and "fills
// the device several times over" -- the reference's own training shape, 8 environments of 300 x 300 pixels at 100
// samples (examples/ppo_tuned.yml:5, state_observer.py:335, render.py:129), sits there.
//
// Same arithmetic and the same pixel <-> RNG-state mapping as render_kernel (rf_render.h).  A wave owns K * 64
// consecutive pixels of an environment (K "sets": lane l renders pixels base + 64 j + l, j < K) and is a block of its
// own: there is NO barrier in the sample loop, nothing a wave waits for but itself.  The rejection loops of
// camera.py:229-252 and physics.py:20-44 are wave-cooperative: every lane makes its first attempt in place; the
// stragglers of all K sets are packed onto the wave's first lanes through the wave's own LDS slots (ranks from the lane
// masks: v_mbcnt), finished there and handed back -- one sparse tail per wave and phase instead of one per set (in-wave
// loops run max-over-lanes trips: 6.5 for a mean of 1.9).  While the K sets hold more stragglers than the wave has lanes
// (tiles inside the target: 0.48 * 64 K on average), every set makes another attempt in place first.  Which lane
// executes an attempt does not change a pixel's stream, so frames and states are bit-identical to render_kernel.
//
// Why not the block-cooperative render_kernel_coop2 here: its tails pool the stragglers of 4 waves x 3 sets and cost 3-5
// block barriers per sample, which only ~7 resident waves per SIMD hide; a launch of 0.1-3 M pixels has 1-4.  Why not
// render_kernel: one pixel per thread leaves nothing to pool (a wave's packed tail runs as many trips as its in-place
// loop), and 1.5-2x the instructions.
//
// LDS order inside a wave: wave_lds_order (rf_coop2.h) -- the wave's slots are nobody else's.
#pragma once

#include "rf_coop2.h"

namespace rf {

#ifndef RF_WAVE_OCC
#define RF_WAVE_OCC 4
#endif
#ifndef RF_WAVE_COLOUR_LDS
#define RF_WAVE_COLOUR_LDS 0
#endif
#ifndef RF_WAVE_XY_LDS
#define RF_WAVE_XY_LDS 0
#endif
#ifndef RF_WAVE_STATE_OUT
#define RF_WAVE_STATE_OUT 1
#endif
#ifndef RF_WAVE_PRIO
#define RF_WAVE_PRIO 0
#endif
#ifndef RF_WAVE_UNROLL
#define RF_WAVE_UNROLL 1
#endif

// sets whose colour sums live in LDS (the others: registers)
template <int K>
struct WaveTune {
    static constexpr int colour_lds = K >= 3 ? RF_WAVE_COLOUR_LDS : (K == 2 ? (RF_WAVE_COLOUR_LDS > 0 ? 1 : 0) : 0);
    static constexpr int occupancy = K >= 3 ? RF_WAVE_OCC : 8; // waves per SIMD the register allocator is held to
};

template <int K>
struct WaveLds {
    uint4 slot[64];   // a straggler's state to its worker; the accepted draws' first four words, then the advanced state, back
    uint2 words2[64]; // sphere: the draws' last two words
#if RF_WAVE_STATE_OUT
    uint4 state_out[64];
#endif
    float colour[WaveTune<K>::colour_lds > 0 ? WaveTune<K>::colour_lds : 1][3][WaveTune<K>::colour_lds > 0 ? 64 : 1]; // colour sums of the sets that keep them here
    float xy[RF_WAVE_XY_LDS ? K : 1][2][RF_WAVE_XY_LDS ? 64 : 1]; // (RF_WAVE_XY_LDS) frames that are no powers of two: (float)x, (float)y of a lane's pixels
};

template <int DIM>
__device__ __forceinline__ float attempt_sq(Rng &g, uint32_t (&w)[6])
{
    return DIM == 2 ? disc_attempt_sq(g, w) : sphere_attempt_sq(g, w);
}
template <int DIM>
__device__ __forceinline__ bool attempt(Rng &g, uint32_t (&w)[6])
{
    return DIM == 2 ? disc_attempt(g, w) : sphere_attempt(g, w);
}

// The stragglers (lane masks `need`) of the K sets finish their loop: packed onto the wave's first lanes when they fit.
template <int DIM, int K>
__device__ __forceinline__ void tails_wave(WaveLds<K> &lds, lanemask (&need)[K], Rng (&g)[K], uint32_t (&w)[K][6], int lane)
{
    int total = 0, first[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        first[j] = total;
        total += (int)__builtin_popcountll(need[j]);
    }
    // more stragglers than lanes: another attempt of every set in place (K wave-attempts; packing them would take
    // ceil(total / 64) attempts and two hand-overs)
    while (total > 64) { // wave-uniform
        total = 0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (need[j] != 0) {
                float sq = 2.0f;
                if (lane_in(need[j]))
                    sq = attempt_sq<DIM>(g[j], w[j]);
                asm volatile("" : "+v"(sq));
                need[j] &= ~lanes_where(sq < 1.0f);
            }
            first[j] = total;
            total += (int)__builtin_popcountll(need[j]);
        }
    }
    if (total == 0) // wave-uniform
        return;
    int slot[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(need[j] >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((unsigned)need[j], 0));
        slot[j] = (first[j] + rank) * 16; // byte offset of the entry
        if (lane_in(need[j]))
            *entry16(lds.slot, slot[j]) = make_uint4(g[j].a_lo, g[j].a_hi, g[j].b_lo, g[j].b_hi);
    }
    wave_lds_order();
    const bool worker = lane < total;
    Rng wg{0, 0, 0, 0};
    if (RF_WAVE_PRIO)
        __builtin_amdgcn_s_setprio(RF_WAVE_PRIO);
    if (worker) {
        const uint4 ps = lds.slot[lane];
        wg = Rng{ps.x, ps.y, ps.z, ps.w};
        uint32_t ww[6] = {0, 0, 0, 0, 0, 0};
        for (;;) { // (RF_WAVE_UNROLL attempts per trip: see rf_coop2.h RF_COOP_UNROLL)
            if (attempt<DIM>(wg, ww))
                break;
#if RF_WAVE_UNROLL >= 2
            if (attempt<DIM>(wg, ww))
                break;
#endif
        }
        lds.slot[lane] = RF_WORDS4(ww); // the accepted draws first: the worker keeps the four state words meanwhile
        if (DIM == 3)
            lds.words2[lane] = RF_WORDS2(ww);
#if RF_WAVE_STATE_OUT
        lds.state_out[lane] = make_uint4(wg.a_lo, wg.a_hi, wg.b_lo, wg.b_hi);
#endif
    }
    if (RF_WAVE_PRIO)
        __builtin_amdgcn_s_setprio(0);
    wave_lds_order();
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (lane_in(need[j])) {
            const uint4 w4 = *entry16(lds.slot, slot[j]);
            w[j][1] = w4.x; w[j][0] = w4.y; w[j][3] = w4.z; w[j][2] = w4.w;
            if (DIM == 3) {
                const uint2 w2 = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(lds.words2) + (slot[j] >> 1));
                w[j][5] = w2.x; w[j][4] = w2.y;
            }
#if RF_WAVE_STATE_OUT
            uint4 ps = *entry16(lds.state_out, slot[j]);
            asm volatile("" : "+v"(ps.x), "+v"(ps.y), "+v"(ps.z), "+v"(ps.w));
            g[j] = Rng{ps.x, ps.y, ps.z, ps.w};
#endif
        }
    }
    wave_lds_order();
#if !RF_WAVE_STATE_OUT
    if (worker)
        lds.slot[lane] = make_uint4(wg.a_lo, wg.a_hi, wg.b_lo, wg.b_hi);
    wave_lds_order();
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (lane_in(need[j])) {
            uint4 ps = *entry16(lds.slot, slot[j]);
            asm volatile("" : "+v"(ps.x), "+v"(ps.y), "+v"(ps.z), "+v"(ps.w));
            g[j] = Rng{ps.x, ps.y, ps.z, ps.w};
        }
    }
    wave_lds_order(); // (the next phase's stragglers overwrite the slots)
#endif
}

// TWO: the environment step's render as one launch (RenderArgs::count2, see render_tile_coop2): the waves of the
// environments below *count2 make two passes over their pixels.
template <bool POW2, int LENS, int K, bool TWO>
__global__ __launch_bounds__(64, WaveTune<K>::occupancy) void render_kernel_wave(RenderArgs a_in)
{
    static_assert(takes_render_args_only<decltype(&render_kernel_wave<POW2, LENS, K, TWO>)>::value,
                  "the two-pass form reads RenderArgs from offset 0 of the kernarg segment");
    constexpr int kColourLdsW = WaveTune<K>::colour_lds;
    __shared__ WaveLds<K> lds;
    static_assert(sizeof(lds.slot) >= (size_t)K * 64 * 3, "the frame stage does not fit");
    uint32_t *const stage = reinterpret_cast<uint32_t *>(lds.slot);

    if (!TWO && skip_env(a_in.rect, (int)blockIdx.y)) // wave-uniform
        return;
    const int passes = (TWO && a_in.env0 + (int)blockIdx.y < *a_in.count2) ? 2 : 1; // wave-uniform
    int e = (int)blockIdx.y, block_x = (int)blockIdx.x, tid = threadIdx.x;
  for (int pass = 0; pass < passes; ++pass) {
    RenderArgs a_pass;
    if (TWO) { // (every pass reads the arguments afresh: see render_tile_coop2)
        unsigned long long kernarg = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kernarg), "+s"(e), "+s"(block_x));
        asm volatile("" : "+v"(tid));
        __builtin_assume(tid >= 0 && tid < 64);
        a_pass = *(const RenderArgs *)(const __attribute__((address_space(4))) RenderArgs *)kernarg;
    }
    const RenderArgs &a = TWO ? a_pass : a_in;
    const float *const scene_cam = (TWO && pass == 1) ? a.cam_dyn2 : a.cam_dyn;
    const float *const scene_rect = (TWO && pass == 1) ? a.rect2 : a.rect;
    uint8_t *const out_frames = (TWO && pass + 1 < passes) ? a.frames2 : a.frames;
    if (TWO && pass == 1)
        wave_lds_order(); // (the first pass's row stores read the stage)

    // Set j of wave b = the 64 pixels from (b + j B) * 64 on, B = gridDim.x: a wave's K sets lie a K-th of the frame apart.
    // Rows that cross the (centred) target cost 2.5x the rows above and below it (the whole sphere phase), and a launch of
    // this size is ONE round of resident waves: with consecutive sets a SIMD's load would be whatever mix of cheap and
    // expensive waves it was dealt (8 x 300^2 x 100: 770 us per launch against 655 with every wave a like mix of both).
    const int chunks_b = (int)gridDim.x;
    auto pixel_of = [&](int t, int j) { return (block_x + j * chunks_b) * 64 + t; };
    Rng g[K];
    float xreg[K] = {}, yreg[K] = {};
#pragma unroll
    for (int j = 0; j < K; ++j) {
        g[j] = rng_load(0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull); // dead lanes: any state
        if (TWO)
            g[j] = Rng{0x7F4A7C15u ^ (uint32_t)tid, 0x9E3779B9u ^ (uint32_t)tid, 0xD192ED03u ^ (uint32_t)tid,
                       0xD1B54A32u ^ (uint32_t)tid};
        const int p = pixel_of(tid, j);
        if (p < a.hw) {
            const ulonglong2 st = a.states[(size_t)e * a.hw + p];
            g[j] = rng_load(st.x, st.y);
        }
        if (!POW2) {
            const int pc = min(p, a.hw - 1);
            const int y = pc / a.w, x = pc - y * a.w;
            xreg[j] = (float)x;
            yreg[j] = (float)y;
            if (RF_WAVE_XY_LDS) {
                lds.xy[j][0][tid] = (float)x;
                lds.xy[j][1][tid] = (float)y;
            }
        }
    }
    PixelEnv env0;
    if (TWO) {
        float cam9[9], rect2[2];
#pragma unroll
        for (int i = 0; i < 9; ++i)
            cam9[i] = as_const(scene_cam + (size_t)e * 9)[i];
        rect2[0] = as_const(scene_rect + (size_t)e * 2)[0];
        rect2[1] = as_const(scene_rect + (size_t)e * 2)[1];
        env0 = make_pixel_env(cam9, rect2);
    } else {
        env0 = make_pixel_env(scene_cam + (size_t)e * 9, scene_rect + (size_t)e * 2);
    }
    auto uniform = [](float v) {
        int bits = __builtin_bit_cast(int, v);
        asm volatile("" : "+v"(bits));
        return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(bits));
    };
    env0.tt = uniform(env0.tt);
    env0.den = uniform(env0.den);
    env0.rden = uniform(env0.rden);

    float cr[K], cg[K], cb[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        cr[j] = cg[j] = cb[j] = 0.0f;
        if (j < kColourLdsW)
            lds.colour[j][0][tid] = lds.colour[j][1][tid] = lds.colour[j][2][tid] = 0.0f;
    }
    lanemask live_m[K];
#pragma unroll
    for (int j = 0; j < K; ++j)
        live_m[j] = lanes_where(pixel_of(tid, j) < a.hw);
    const int tmiss_s = __builtin_amdgcn_readfirstlane((int)env0.tmiss);
    const int log2w = POW2 ? __builtin_ctz((unsigned)a.w) : 0;
    wave_lds_order();
    for (int k = 0; k < a.spp; ++k) {
        const PixelEnv &env = env0;
        uint32_t w[K][6];
        float s[K], t[K];
        lanemask need_m[K];
        int tk = tid;
        asm volatile("" : "+v"(tk)); // (the pixel geometry is derived afresh every sample: loop invariants spill)
        __builtin_assume(tk >= 0 && tk < 64);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            int x = 0, y = 0;
            float xf, yf;
            if (POW2) {
                const int p = pixel_of(tk, j);
                x = p & (a.w - 1);
                y = p >> log2w;
                xf = (float)x;
                yf = (float)y;
            } else if (RF_WAVE_XY_LDS) {
                xf = lds.xy[j][0][tk];
                yf = lds.xy[j][1][tk];
            } else {
                xf = xreg[j];
                yf = yreg[j];
            }
            {
                uint32_t xh, xl, yh, yl;
                rng_next(g[j], xh, xl);
                rng_next(g[j], yh, yl);
                const float xi = unit_f32_scaled64(xh, xl), yi = unit_f32_scaled64(yh, yl); // 2^64 * uniform
                if (POW2) {
                    s[j] = pixel_coord_pow2_64(xf, xi, a.fc.inv_w);
                    t[j] = pixel_coord_pow2_64(yf, yi, a.fc.inv_h);
                } else {
                    // pixel_coord_div with (double)x taken from the float (exact: x < 2^24)
                    const double ax = (double)xf + (double)(xi * kTwoM64), ay = (double)yf + (double)(yi * kTwoM64);
                    const double qx = ax * a.fc.rw64, qy = ay * a.fc.rh64;
                    s[j] = (float)__builtin_fma(__builtin_fma(-qx, a.fc.w64, ax), a.fc.rw64, qx);
                    t[j] = (float)__builtin_fma(__builtin_fma(-qy, a.fc.h64, ay), a.fc.rh64, qy);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                w[j][i] = any_u32();
            const float sq = disc_attempt_sq(g[j], w[j]);
            need_m[j] = live_m[j] & ~lanes_where(sq < 1.0f);
        }
        tails_wave<2, K>(lds, need_m, g, w, tid);

        float rdx[K], rdy[K], rdz[K];
        lanemask hit_m[K];
        bool red[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            float p0, p1;
            disc_finish(w[j], p0, p1);
            const AxisRay ray = sample_axis_point<LENS>(p0, p1, env, a.cs, s[j], t[j]);
            rdx[j] = ray.dx;
            rdy[j] = ray.dy;
            rdz[j] = ray.dz;
            hit_m[j] = live_m[j] & lanes_where(!(ray.reach > env.rect.half)); // rectangle.py:135
            if (tmiss_s)                                                       // rectangle.py:130
                hit_m[j] = 0;
            red[j] = false;
            if (lane_in(hit_m[j]))
                red[j] = sample_axis_red(ray.px, ray.py, env, a.tab);
            w[j][4] = any_u32();
            w[j][5] = any_u32();
            need_m[j] = hit_m[j];
            if (need_m[j] != 0) { // wave-uniform
                float sq = 2.0f;
                if (lane_in(need_m[j]))
                    sq = sphere_attempt_sq(g[j], w[j]);
                asm volatile("" : "+v"(sq));
                need_m[j] &= ~lanes_where(sq < 1.0f);
            }
        }
        tails_wave<3, K>(lds, need_m, g, w, tid);

#pragma unroll
        for (int j = 0; j < K; ++j) {
            float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f;
            const bool hit = lane_in(hit_m[j]);
            if (hit)
                sphere_finish(w[j], q0, q1, q2);
            const Colour c = sample_axis_shade(hit, red[j], rdx[j], rdy[j], rdz[j], q0, q1, q2);
            if (j < kColourLdsW) {
                lds.colour[j][0][tid] = add2_not_negzero(lds.colour[j][0][tid], c.r);
                lds.colour[j][1][tid] = add2_not_negzero(lds.colour[j][1][tid], c.g);
                lds.colour[j][2][tid] = add2_not_negzero(lds.colour[j][2][tid], c.b);
                continue;
            }
            cr[j] = add2_not_negzero(cr[j], c.r);
            cg[j] = add2_not_negzero(cg[j], c.g);
            cb[j] = add2_not_negzero(cb[j], c.b);
        }
    }
#pragma unroll
    for (int j = 0; j < kColourLdsW && j < K; ++j) {
        cr[j] = lds.colour[j][0][tid];
        cg[j] = lds.colour[j][1][tid];
        cb[j] = lds.colour[j][2][tid];
    }
    wave_lds_order(); // the slots are dead from here on: they become the stage

    uint8_t *sb = reinterpret_cast<uint8_t *>(stage);
    int te = tid;
    asm volatile("" : "+v"(te));
    __builtin_assume(te >= 0 && te < 64);
    const bool dwords = (a.hw & 3) == 0; // every set's 192 bytes start on a dword then
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int p = pixel_of(te, j);
        const bool live = p < a.hw;
        if (live)
            a.states[(size_t)e * a.hw + p] = make_ulonglong2(rng_s0(g[j]), rng_s1(g[j]));
        const uint8_t r8 = (uint8_t)(cr[j] * a.scale);
        const uint8_t g8 = (uint8_t)(cg[j] * a.scale);
        const uint8_t b8 = (uint8_t)(cb[j] * a.scale);
        if (dwords) {
            const int sl = 64 * j + te;
            sb[sl * 3 + 0] = r8;
            sb[sl * 3 + 1] = g8;
            sb[sl * 3 + 2] = b8;
        } else if (live) {
            uint8_t *dst = out_frames + ((size_t)e * a.hw + p) * 3;
            dst[0] = r8;
            dst[1] = g8;
            dst[2] = b8;
        }
    }
    if (dwords) {
        wave_lds_order();
#pragma unroll
        for (int j = 0; j < K; ++j) { // 192 bytes per set: 48 dwords
            const int first = pixel_of(0, j);
            const int ndw = min(64, a.hw - first) * 3 / 4; // (<= 0: the set lies beyond the frame)
            uint32_t *dst = reinterpret_cast<uint32_t *>(out_frames + ((size_t)e * a.hw + first) * 3);
            if (te < ndw)
                dst[te] = stage[48 * j + te];
        }
    }
  } // pass
}

} // namespace rf
