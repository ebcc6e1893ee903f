// rf_coop2.h -- render_kernel_coop2<POW2>: render_kernel_coop with kSets pixels per thread.
//
// Same arithmetic, same pixel <-> RNG-state mapping, same cooperative tails as render_kernel_coop
// (rf_kernels.h).  A block still has four waves, but its tile is kSets times as high
// (128 x 2 kSets pixels): a thread owns the pixels (x, y + 2 j), j < kSets.  Between two barriers
// every wave now does the in-wave work of kSets pixel sets, and one cooperative call finishes
// the stragglers of all of them, so the time a wave spends waiting for a tail -- more than half
// of all wave-cycles in the one-set kernel -- is paid once per kSets samples' worth of work.
#pragma once

#include "rf_kernels.h"

namespace rf {

// Measured at the headline config (G samples/s; 1 set = render_kernel_coop = 122): 2 sets 133,
// 3 sets 138 (at 7 waves per SIMD; 137 at 6, 132 at 8 with heavy spilling), 4 sets 114.
#ifndef RF_SETS
#define RF_SETS 3
#endif
#ifndef RF_SETS_OCC
#define RF_SETS_OCC 7
#endif
constexpr int kSets = RF_SETS;
constexpr int kTileH2 = kTileH * kSets;

// coop_finish for kSets pixel sets at once.  The packed list holds at most kBlock entries (the
// LDS arrays of CoopLds); stragglers that would not fit -- more than a third of all lanes still
// looking, which does not happen in practice -- finish their loop in their own wave instead.
template <int DIM>
__device__ __forceinline__ void coop_finish2(CoopLds &lds, int parity, bool (&need)[kSets], Rng (&g)[kSets],
                                             uint32_t (&w)[kSets][6])
{
    const int tid = threadIdx.x;
    uint4 *const state = lds.state[parity];
    unsigned long long ballot[kSets];
    int pop = 0;
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        ballot[j] = __ballot(need[j]);
        pop += (int)__popcll(ballot[j]);
    }
    int base = 0;
    if (pop != 0) { // wave-uniform
        if ((tid & 63) == 0)
            base = atomicAdd(&lds.cnt[parity], pop);
        base = __builtin_amdgcn_readfirstlane(base);
    }
    int slot[kSets];
    bool parked[kSets];
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        slot[j] = base + __builtin_amdgcn_mbcnt_hi((unsigned)(ballot[j] >> 32),
                                                   __builtin_amdgcn_mbcnt_lo((unsigned)ballot[j], 0));
        base += (int)__popcll(ballot[j]);
        parked[j] = need[j] && slot[j] < kBlock;
        if (parked[j])
            state[slot[j]] = make_uint4(g[j].a_lo, g[j].a_hi, g[j].b_lo, g[j].b_hi);
        if (need[j] && !parked[j]) { // overflow of the packed list: finish in place
            if (DIM == 2) {
                while (!disc_attempt(g[j], w[j])) {
                }
            } else {
                while (!sphere_attempt(g[j], w[j])) {
                }
            }
        }
    }
    __syncthreads();
    const int total = min(lds.cnt[parity], kBlock);
    if (total == 0) // block-uniform
        return;
    if (tid < total) {
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
    }
    __syncthreads();
    if (tid == 0)
        lds.cnt[parity] = 0;
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        if (parked[j]) {
            const uint4 ps = state[slot[j]];
            g[j] = Rng{ps.x, ps.y, ps.z, ps.w};
            const uint4 w4 = lds.words4[slot[j]];
            w[j][0] = w4.x; w[j][1] = w4.y; w[j][2] = w4.z; w[j][3] = w4.w;
            if (DIM == 3) {
                const uint2 w2 = lds.words2[slot[j]];
                w[j][4] = w2.x; w[j][5] = w2.y;
            }
        }
    }
}

template <bool POW2>
__global__ __launch_bounds__(kBlock, RF_SETS_OCC) void render_kernel_coop2(RenderArgs a)
{
    __shared__ uint32_t stage[kSets * kBlock * 3 / 4];
    __shared__ CoopLds lds;

    const int e = blockIdx.y;
    const int tid = threadIdx.x;
    if (tid < 2)
        lds.cnt[tid] = 0;
    __syncthreads();
    const int tiles_x = (a.w + kTileW - 1) / kTileW;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x - tile_y * tiles_x;
    const int wv = tid >> 6, lane = tid & 63;
    const bool mirror = (2 * tile_x + 1) * kTileW > a.w; // see render_kernel_coop
    const int wx = mirror ? (kWavesX - 1 - wv % kWavesX) : (wv % kWavesX);
    const int col = wx * kWaveW + (lane % kWaveW);
    const int row0 = (wv / kWavesX) * kWaveH + (lane / kWaveW); // set j is kTileH rows further down
    const int x = tile_x * kTileW + col;
    const float xf = (float)x;

    // set j covers the rows kTileH * j further down; everything per set that is cheap to
    // recompute (y, pixel index, liveness) is recomputed to keep two pixel states in 64 VGPRs
    const int y0 = tile_y * kTileH2 + row0;
    const bool live_x = x < a.w;
    auto y_of = [&](int j) { return y0 + j * kTileH; };
    auto live_of = [&](int j) { return live_x && y_of(j) < a.h; };
    auto pix_of = [&](int j) { return (size_t)e * a.hw + (live_of(j) ? (size_t)y_of(j) * a.w + x : 0); };
    Rng g[kSets];
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        g[j] = rng_load(0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull); // dead lanes: any state
        if (live_of(j)) {
            const ulonglong2 st = a.states[pix_of(j)];
            g[j] = rng_load(st.x, st.y);
        }
    }
    const PixelEnv env = make_pixel_env(a.cam_dyn + (size_t)e * 9, a.rect + (size_t)e * 2);

    float cr[kSets], cg[kSets], cb[kSets];
#pragma unroll
    for (int j = 0; j < kSets; ++j)
        cr[j] = cg[j] = cb[j] = 0.0f;

    for (int k = 0; k < a.spp; ++k) {
        uint32_t w[kSets][6];
        float s[kSets], t[kSets];
        bool need[kSets];
#pragma unroll
        for (int j = 0; j < kSets; ++j) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
                w[j][i] = 0;
            sample_coords<POW2>(g[j], x, y_of(j), xf, (float)y_of(j), a.h, a.w, a.inv_w, a.inv_h, s[j], t[j]);
            need[j] = live_of(j);
            if (need[j] && disc_attempt(g[j], w[j]))
                need[j] = false;
        }
        coop_finish2<2>(lds, 0, need, g, w);

        AxisPre pre[kSets];
#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            float p0, p1;
            disc_finish(w[j], p0, p1);
            pre[j] = sample_axis_ray(p0, p1, env, a.cs.lens_radius, s[j], t[j], a.tab);
            need[j] = live_of(j) && pre[j].hit;
            for (int trip = 0; trip < kCoopTrips; ++trip) {
                if (__any(need[j])) { // wave-uniform
                    if (need[j] && sphere_attempt(g[j], w[j]))
                        need[j] = false;
                }
            }
        }
        coop_finish2<3>(lds, 1, need, g, w);

#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f;
            if (pre[j].hit)
                sphere_finish(w[j], q0, q1, q2);
            const Colour c = sample_axis_shade(pre[j], q0, q1, q2);
            cr[j] = add2(cr[j], c.r);
            cg[j] = add2(cg[j], c.g);
            cb[j] = add2(cb[j], c.b);
        }
    }

    uint8_t *sb = reinterpret_cast<uint8_t *>(stage);
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        if (live_of(j))
            a.states[pix_of(j)] = make_ulonglong2(rng_s0(g[j]), rng_s1(g[j]));
        const uint8_t r8 = (uint8_t)(cr[j] * a.scale);
        const uint8_t g8 = (uint8_t)(cg[j] * a.scale);
        const uint8_t b8 = (uint8_t)(cb[j] * a.scale);
        if ((a.w & 3) == 0) {
            const int slot = (j * kTileH + row0) * kTileW + col;
            sb[slot * 3 + 0] = r8;
            sb[slot * 3 + 1] = g8;
            sb[slot * 3 + 2] = b8;
        } else if (live_of(j)) {
            uint8_t *dst = a.frames + pix_of(j) * 3;
            dst[0] = r8;
            dst[1] = g8;
            dst[2] = b8;
        }
    }
    if ((a.w & 3) == 0) {
        // the tile's rows (kTileW * 3 B each) -> LDS -> coalesced dword stores per row
        __syncthreads();
        constexpr int kRowDw = kTileW * 3 / 4;
        for (int i = tid; i < kTileH2 * kRowDw; i += kBlock) {
            const int r = i / kRowDw, d = i - r * kRowDw;
            const int yy = tile_y * kTileH2 + r;
            const int valid_dw = min(kTileW, a.w - tile_x * kTileW) * 3 / 4; // w % 4 == 0
            if (yy < a.h && d < valid_dw) {
                uint32_t *dst = reinterpret_cast<uint32_t *>(
                    a.frames + (((size_t)e * a.h + yy) * a.w + (size_t)tile_x * kTileW) * 3);
                dst[d] = stage[r * kRowDw + d];
            }
        }
    }
}

} // namespace rf
