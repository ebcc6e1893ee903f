// rf_kernels.h -- gfx950 kernels of the render-and-measure path.
//
//   render_kernel   FastRenderer._device_render     (graphics/render.py:190-246)
//   focus_kernel    vision.focus_value chain         (vision.py:23-25), per-env sums
//   focus_finalize  ndarray.var()                    (vision.py:25)
//   seed_kernel     make_random_states               (graphics/random.py:8-18)
//
// Data layout in HBM (all owned by rf_ctx):
//   states  uint64x2[n*h*w]   index = e*h*w + y*w + x  (render.py:217) -> a wave's 64
//                             lanes read/write 1 KiB contiguous (global_*_dwordx4)
//   frames  uint8[n][h][w][3] lanes along x; 768 B per 256-thread block are staged in
//                             LDS and leave as 192 coalesced dword stores
//   cam_dyn float[n][9], rect float[n][2]   per-env parameters (block-uniform)
//
// The render kernel is VALU-bound (64-bit integer RNG + rejection loops), not
// HBM-bound: 38 B/pixel of traffic against ~10^4 lane-ops/pixel at 16 spp.  No MFMA:
// nothing here is a contraction.
#pragma once

#include <hip/hip_runtime.h>

#include "rf_general.h"
#include "rf_math.h"

namespace rf {

constexpr int kBlock = 256;

__device__ __forceinline__ bool skip_env(const float *rect, int e)
{
    return __builtin_bit_cast(uint32_t, rect[2 * (size_t)e]) == kSkipEnvBits;
}

// The scene arrays (cameras, shape parameters) are written before the launch (by the host or by an earlier kernel) and never by the kernel that reads them: read
// through the constant address space, a block-uniform address becomes an s_load into scalar registers.  (Through a plain
// pointer the compiler has to assume that the kernel's own stores and atomics may have changed them: it then re-reads
// them after every barrier with one vector load per lane.)
template <class T>
using const_as = const __attribute__((address_space(4))) T;
template <class T>
__device__ __forceinline__ const_as<T> *as_const(const T *p)
{
    return (const_as<T> *)(unsigned long long)p;
}

struct RenderArgs {
    uint8_t *frames;
    ulonglong2 *states;
    const float *cam_dyn; // [n][9]
    const float *rect;    // [n][2]
    CamStatic cs;
    CheckerTable tab;
    int n, h, w, spp;
    int hw;          // h*w
    float scale;     // float32(255.0 / spp)   (render.py:244-246)
    float inv_w, inv_h; // exact reciprocals when w / h are powers of two
    double rw64, rh64;  // RN64(1 / w), RN64(1 / h) for pixel_coord_div
    double w64, h64;    // (double)w, (double)h: scalar operands, no per-lane conversions
    // render_kernel_coop2<..., TWO = true> (the environment step as one launch, rf_abi.hip enqueue_env_step_fused): the
    // blocks of the environments below *count2 render their tile twice -- the step's frame into frames2, then the scene
    // cam_dyn2 / rect2 of the same slot into frames, continuing the pixels' RNG streams (vector_environment.py:137-151:
    // the r-th environment that ended is rendered again as row r of a compacted set, render.py:217)
    const int *count2;
    const float *cam_dyn2, *rect2;
    uint8_t *frames2;
    int env0; // index of the launch's first environment (launches hold at most 65535)
    // render_kernel_coop2_strip: blocks [0, main_tiles) of a grid row render columns [0, strip_x0), the others the rest
    int main_tiles, strip_x0;
};

// AXIS / POW2: exact specialisations, see rf_math.h render_pixel.
// TWO: the fused environment step's form (RenderArgs::count2): the threads of the environments below *count2 render their
// pixel twice -- the step's frame into frames2, then the scene cam_dyn2 / rect2 of the same slot into frames -- with the
// RNG state staying in registers in between.
template <bool AXIS, bool POW2, bool TWO = false>
__global__ __launch_bounds__(kBlock) void render_kernel(RenderArgs a)
{
    __shared__ uint32_t stage[kBlock * 3 / 4];

    const int e = blockIdx.y;
    if (!TWO && skip_env(a.rect, e)) // block-uniform, before any barrier
        return;
    const int passes = (TWO && a.env0 + e < *a.count2) ? 2 : 1; // block-uniform
    const int p = blockIdx.x * kBlock + threadIdx.x; // pixel within the env
    const bool live = p < a.hw;
    const int y = p / a.w;
    const int x = p - y * a.w;
    const size_t pix = (size_t)e * a.hw + (live ? p : 0);
    Rng g = rng_load(0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull);
    if (live) {
        const ulonglong2 st = a.states[pix];
        g = rng_load(st.x, st.y);
    }
    for (int pass = 0; pass < passes; ++pass) {
        const float *const cam = (TWO && pass == 1) ? a.cam_dyn2 : a.cam_dyn;
        const float *const rect = (TWO && pass == 1) ? a.rect2 : a.rect;
        uint8_t *const frames = (TWO && pass + 1 < passes) ? a.frames2 : a.frames;
        float cr = 0.0f, cg = 0.0f, cb = 0.0f;
        if (live) {
            const PixelEnv env = make_pixel_env(cam + (size_t)e * 9, rect + (size_t)e * 2);
            render_pixel<AXIS, POW2>(g, x, y, a.h, a.w, a.spp, a.inv_w, a.inv_h, a.rw64, a.rh64, env, a.cs, a.tab, cr, cg, cb);
        }

        // uint8 truncation of float32(colour * scale)   (render.py:244-246)
        const uint8_t r8 = (uint8_t)(cr * a.scale);
        const uint8_t g8 = (uint8_t)(cg * a.scale);
        const uint8_t b8 = (uint8_t)(cb * a.scale);

        const size_t block_px = (size_t)e * a.hw + (size_t)blockIdx.x * kBlock;
        if ((a.hw & 3) == 0) {
            // 768 B per block -> LDS -> 192 coalesced dword stores (block base is 4-aligned)
            uint8_t *sb = reinterpret_cast<uint8_t *>(stage);
            sb[threadIdx.x * 3 + 0] = r8;
            sb[threadIdx.x * 3 + 1] = g8;
            sb[threadIdx.x * 3 + 2] = b8;
            __syncthreads();
            const int count = min(kBlock, a.hw - (int)blockIdx.x * kBlock); // multiple of 4
            const int ndw = count * 3 / 4;
            if ((int)threadIdx.x < ndw) {
                uint32_t *dst = reinterpret_cast<uint32_t *>(frames + block_px * 3);
                dst[threadIdx.x] = stage[threadIdx.x];
            }
            if (TWO && pass + 1 < passes)
                __syncthreads(); // (the next pass writes the stage again)
        } else if (live) {
            uint8_t *dst = frames + (block_px + threadIdx.x) * 3;
            dst[0] = r8;
            dst[1] = g8;
            dst[2] = b8;
        }
    }
    if (live)
        a.states[pix] = make_ulonglong2(rng_s0(g), rng_s1(g));
}

// ---------------------------------------------------------------------------
// render_kernel_coop<POW2>: same arithmetic, canonical camera only, with the sphere
// rejection loop made block-cooperative.
//
// In a wave whose 64 pixels all hit the target the loop of physics.py:31-44 runs for
// max-over-lanes = ~6.5 trips although a lane needs 1.9 on average: after two trips 23 %
// of the lanes are still looking for a candidate and they hold the whole wave.  Here every
// lane makes at most kCoopTrips attempts in its own wave; the stragglers of all four waves
// then park their RNG state in LDS, are packed densely onto the first lanes of the block
// (one wave is usually enough for the ~58 of them), finish their loops there, and hand the
// accepted draws and the advanced state back.  Which lane executes an attempt does not
// matter -- the stream of a pixel is advanced by exactly the same draws -- so results are
// bit-identical to render_kernel (and to the oracle).
//
// All 256 threads of a block run the sample loop in lockstep (dead threads of a partial
// block included) so that every barrier is reached by every thread.
// ---------------------------------------------------------------------------

// Block-cooperative tail of a rejection loop.  `need` marks the lanes whose loop has not
// accepted yet after their in-wave attempts; on return every such lane holds the advanced
// RNG state and the accepted raw draws in w[0 .. 2*DIM).  DIM = 2: disc, DIM = 3: sphere.
// `parity` selects the counter and the state buffer; callers alternate it between consecutive
// calls.  Two barriers per call (one when nobody is left):
//   park:    every wave takes slots with one LDS atomic (the order of the packed entries is
//            irrelevant) and its stragglers write their RNG state            -- barrier --
//   finish:  the first `total` threads of the block run the loops of the packed entries to
//            the end and publish state + accepted draws                       -- barrier --
//   collect: the owners read their entry back.
// The state buffer is doubled because a fast wave parks the stragglers of the next call while
// a slow one is still collecting; the draws are only written after the next call's first
// barrier, and the counter of this parity is next touched two calls later.
template <int N>
struct CoopLdsT {
    uint4 state[2][N];
    uint4 words4[N];
    uint2 words2[N];
    uint16_t owner[N]; // render_kernel_coop2: original slot of a re-packed entry
    int cnt[2];
    int cnt2;          // render_kernel_coop2: entries in the second round
};
using CoopLds = CoopLdsT<kBlock>;

// RF_TEST_SKEW (tests/gpucheck/libreinfocus_skew.so, tests/test_gpu_parity.py; used by coop_finish here and by
// rf_coop2.h): one wave of every block -- a different
// one from call to call -- sleeps ~8 000 cycles at each point where a cooperative call is ordered against the next
// one by a barrier alone: before it reads the counter after B1, before thread 0's resets, before the collect reads.
// With the ordering right the sleeps change nothing (frames and RNG states stay bit-identical to the oracle); the
// round-3 form of the call -- no B4, one counter -- produces wrong frames under them (profiles/r04_ab.txt section 7).
#ifndef RF_TEST_SKEW
#define RF_TEST_SKEW 0
#endif
__device__ __forceinline__ void test_skew(int tid, int turn)
{
#if RF_TEST_SKEW
    if (((tid >> 6) & 3) == (turn & 3)) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_s_sleep(127); // 8 x 127 x 64 cycles
    }
#else
    (void)tid;
    (void)turn;
#endif
}

template <int DIM>
__device__ __forceinline__ void coop_finish(CoopLds &lds, int parity, bool need, Rng &g, uint32_t *w)
{
    const int tid = threadIdx.x;
    uint4 *const state = lds.state[parity];
    const unsigned long long ballot = __ballot(need);
    int base = 0;
    if (ballot != 0) { // wave-uniform
        if ((tid & 63) == 0)
            base = atomicAdd(&lds.cnt[parity], __popcll(ballot));
        base = __builtin_amdgcn_readfirstlane(base);
    }
    const int lane_rank = __builtin_amdgcn_mbcnt_hi((unsigned)(ballot >> 32),
                                                    __builtin_amdgcn_mbcnt_lo((unsigned)ballot, 0));
    const int slot = base + lane_rank;
    if (need)
        state[slot] = make_uint4(g.a_lo, g.a_hi, g.b_lo, g.b_hi);
    __syncthreads();
    test_skew(tid, parity + 1); // (a wave late with its read of the counter)
    const int total = lds.cnt[parity];
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
    test_skew(tid, 0); // (wave 0, late with its reset)
    if (tid == 0)
        lds.cnt[parity] = 0;
    test_skew(tid, total + 2); // (a wave late with its collect reads)
    if (need) {
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

constexpr int kCoopTrips = 2; // in-wave sphere attempts before the cooperative tail (the disc loop makes one)

// Pixel <-> lane mapping of the cooperative kernel: a wave owns kWaveW x kWaveH pixels and a
// block kWavesX x (4 / kWavesX) waves.  The mapping only changes which thread owns a pixel,
// not the pixel's RNG stream, so it is a pure scheduling knob (a wave always reads / writes
// whole 128-B lines of RNG state as long as kWaveW >= 8).  Measured at the headline config
// (tools sweep, round 1): 32 x 2 pixel waves side by side (128 x 2 blocks) are 8 % faster
// than 64 x 1 rows and than squarer tiles.
constexpr int kWaveW = 32, kWaveH = 64 / kWaveW;
constexpr int kWavesX = 4, kWavesY = (kBlock / 64) / kWavesX;
constexpr int kTileW = kWavesX * kWaveW, kTileH = kWavesY * kWaveH;

template <bool POW2>
__global__ __launch_bounds__(kBlock) void render_kernel_coop(RenderArgs a)
{
    __shared__ uint32_t stage[kBlock * 3 / 4];
    __shared__ CoopLds lds;

    const int e = blockIdx.y;
    if (skip_env(a.rect, e)) // block-uniform, before any barrier
        return;
    const int tid = threadIdx.x;
    if (tid < 2)
        lds.cnt[tid] = 0;
    __syncthreads();
    const int tiles_x = (a.w + kTileW - 1) / kTileW;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x - tile_y * tiles_x;
    const int wv = tid >> 6, lane = tid & 63;
    // Tiles in the right half of the frame place their waves right to left, so that wave 0 --
    // which finishes the cooperative tails -- is the outermost wave on both sides of the
    // (centred) target: the one with the fewest hit lanes of its own (+2.6 % measured).
    const bool mirror = (2 * tile_x + 1) * kTileW > a.w;
    const int wx = mirror ? (kWavesX - 1 - wv % kWavesX) : (wv % kWavesX);
    const int col = wx * kWaveW + (lane % kWaveW); // column within the tile
    const int row = (wv / kWavesX) * kWaveH + (lane / kWaveW); // row within the tile
    const int x = tile_x * kTileW + col;
    const int y = tile_y * kTileH + row;
    const bool live = x < a.w && y < a.h;
    const size_t pix = (size_t)e * a.hw + (live ? (size_t)y * a.w + x : 0);

    Rng g = rng_load(0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull); // dead lanes: any state
    if (live) {
        const ulonglong2 st = a.states[pix];
        g = rng_load(st.x, st.y);
    }
    const PixelEnv env = make_pixel_env(a.cam_dyn + (size_t)e * 9, a.rect + (size_t)e * 2);
    const float xf = (float)x, yf = (float)y;

    float cr = 0.0f, cg = 0.0f, cb = 0.0f;
    for (int k = 0; k < a.spp; ++k) {
        float s, t;
        sample_coords<POW2>(g, x, y, xf, yf, a.h64, a.w64, a.inv_w, a.inv_h, a.rw64, a.rh64, s, t);
        uint32_t w[6] = {0, 0, 0, 0, 0, 0};
        bool dneed = live;
        if (__any(dneed)) {
            if (dneed && disc_attempt(g, w))
                dneed = false;
        }
        coop_finish<2>(lds, 0, dneed, g, w);
        float p0, p1;
        disc_finish(w, p0, p1);
        const AxisPre pre = sample_axis_ray(p0, p1, env, a.cs, s, t, a.tab);

        bool need = live && pre.hit;
        for (int trip = 0; trip < kCoopTrips; ++trip) {
            if (__any(need)) { // wave-uniform
                if (need && sphere_attempt(g, w))
                    need = false;
            }
        }
        coop_finish<3>(lds, 1, need, g, w);

        float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f;
        if (pre.hit)
            sphere_finish(w, q0, q1, q2);
        const Colour c = sample_axis_shade(pre, q0, q1, q2);
        cr = add2(cr, c.r);
        cg = add2(cg, c.g);
        cb = add2(cb, c.b);
    }
    if (live)
        a.states[pix] = make_ulonglong2(rng_s0(g), rng_s1(g));

    const uint8_t r8 = (uint8_t)(cr * a.scale);
    const uint8_t g8 = (uint8_t)(cg * a.scale);
    const uint8_t b8 = (uint8_t)(cb * a.scale);
    if ((a.w & 3) == 0) {
        // the tile's rows (kTileW * 3 B each) -> LDS -> coalesced dword stores per row
        uint8_t *sb = reinterpret_cast<uint8_t *>(stage);
        const int slot = row * kTileW + col;
        sb[slot * 3 + 0] = r8;
        sb[slot * 3 + 1] = g8;
        sb[slot * 3 + 2] = b8;
        __syncthreads();
        constexpr int kRowDw = kTileW * 3 / 4; // dwords per tile row; kTileH * kRowDw == 192
        if (tid < kTileH * kRowDw) {
            const int r = tid / kRowDw, d = tid - r * kRowDw;
            const int yy = tile_y * kTileH + r;
            const int valid_dw = min(kTileW, a.w - tile_x * kTileW) * 3 / 4; // w % 4 == 0
            if (yy < a.h && d < valid_dw) {
                uint32_t *dst = reinterpret_cast<uint32_t *>(
                    a.frames + (((size_t)e * a.h + yy) * a.w + (size_t)tile_x * kTileW) * 3);
                dst[d] = stage[r * kRowDw + d];
            }
        }
    } else if (live) {
        uint8_t *dst = a.frames + pix * 3;
        dst[0] = r8;
        dst[1] = g8;
        dst[2] = b8;
    }
}

// ---------------------------------------------------------------------------
// render_general_kernel: device_render (graphics/render.py:31-85) for worlds of spheres and
// rectangles with per-environment cameras; arithmetic in rf_general.h, one thread per pixel,
// lanes along x, frame bytes staged through LDS.  Held to 5 waves per SIMD: with the float64
// library calls inlined the kernel needed 208 VGPRs (2 waves per SIMD, 42.5 G samples/s on
// one-rectangle scenes); with them out of line 112 (4 waves: 55.6), and at 96 registers with six
// spilled (5 waves) 58.0 -- tools/bench_general.py, profiles/README.md.  (Sphere scenes gained another
// 13 % from deciding a hit's checker colour in float32 where that is safe: rf_general.h sphere_red.)
// ---------------------------------------------------------------------------
struct GeneralArgs {
    uint8_t *frames;
    ulonglong2 *states;
    const GeneralCamera *cameras; // [n], cast from float64[n][19] on the host
    const float *params;    // [n][most][width]
    const int32_t *types;   // [n][most]
    const int32_t *sizes;   // [n]
    int n, h, w, spp, hw, most, width;
    float scale;
};

constexpr int kGeneralOcc = 5; // waves per SIMD the literal kernel's register allocation is held to (6: spills, no faster)
template <bool POW2>
__global__ __launch_bounds__(kBlock, kGeneralOcc) void render_general_kernel(GeneralArgs a)
{
    __shared__ uint32_t stage[kBlock * 3 / 4]; // the block's 256 pixels x 3 B, stored as 192 coalesced dwords
    const int e = blockIdx.y;
    const int p0 = blockIdx.x * kBlock;
    const int p = p0 + threadIdx.x;
    const bool live = p < a.hw;
    const size_t pix = (size_t)e * a.hw + (live ? p : 0);
    uint8_t r8 = 0, g8 = 0, b8 = 0;
    if (live) {
        const int y = p / a.w, x = p - y * a.w;
        const ulonglong2 st = a.states[pix];
        Rng g = rng_load(st.x, st.y);
        float cr, cg, cb;
        render_pixel_general<POW2>(g, x, y, a.h, a.w, a.spp, a.cameras[e],
                             a.params + ((size_t)e * a.most) * a.width, a.types + (size_t)e * a.most, a.sizes[e],
                             a.width, cr, cg, cb);
        a.states[pix] = make_ulonglong2(rng_s0(g), rng_s1(g));
        r8 = (uint8_t)(cr * a.scale);
        g8 = (uint8_t)(cg * a.scale);
        b8 = (uint8_t)(cb * a.scale);
    }
    // a full block whose first byte is dword-aligned goes through LDS; anything else stores bytes
    // (the ADDRESS decides: a.frames is the chunk's base, which for the chunks after the first --
    // 65535 environments each -- is itself only 4-byte aligned when h * w * 3 is a multiple of 4)
    const size_t first_byte = ((size_t)e * a.hw + p0) * 3;
    const bool staged = p0 + kBlock <= a.hw && (reinterpret_cast<uintptr_t>(a.frames + first_byte) & 3) == 0; // block-uniform
    if (staged) {
        uint8_t *sb = reinterpret_cast<uint8_t *>(stage);
        sb[threadIdx.x * 3 + 0] = r8;
        sb[threadIdx.x * 3 + 1] = g8;
        sb[threadIdx.x * 3 + 2] = b8;
        __syncthreads();
        if (threadIdx.x < kBlock * 3 / 4)
            reinterpret_cast<uint32_t *>(a.frames + first_byte)[threadIdx.x] = stage[threadIdx.x];
    } else if (live) {
        uint8_t *dst = a.frames + pix * 3;
        dst[0] = r8;
        dst[1] = g8;
        dst[2] = b8;
    }
}

// ---------------------------------------------------------------------------
// focus: gray -> median3x3 (replicate) -> Laplacian (reflect-101, sat u8) -> sums
// One block per (row band, env).  Integer/byte work, HBM-bound: 3 B/pixel read.
// ---------------------------------------------------------------------------
constexpr int kBand = 16; // output rows per block

struct FocusArgs {
    const uint8_t *frames;
    unsigned long long *sums; // [n][2] = (sum, sum of squares), zeroed before launch
    int n, h, w;
    int gray15; // 1: 15-bit coefficients, 0: 14-bit
    const float *skip_rect; // scene rectangles when slots may be marked kSkipEnvBits, else null
    // both measures of a fused environment step in one launch (count2 != null; frames, sums, sums2 are then the arrays'
    // bases and row0 the launch's first row): rows [0, n_step) are the step's frames -- frames2 for the environments
    // below *count2, frames for the others -- into sums; rows n_step + r, r < *count2, the re-rendered frames[r] into sums2
    const int *count2;
    const uint8_t *frames2;
    unsigned long long *sums2;
    int n_step, row0;
};

// which frame a block of the focus kernels reads and where its sums go; false: nothing to do
__device__ __forceinline__ bool focus_row(const FocusArgs &a, int row, const uint8_t *&img, unsigned long long *&sums)
{
    const size_t frame = (size_t)a.h * a.w * 3;
    if (a.count2 == nullptr) {
        if (a.skip_rect != nullptr && skip_env(a.skip_rect, row))
            return false;
        img = a.frames + frame * row;
        sums = a.sums + 2 * (size_t)row;
        return true;
    }
    const int count = *a.count2;
    row += a.row0;
    if (row < a.n_step) {
        img = (row < count ? a.frames2 : a.frames) + frame * row;
        sums = a.sums + 2 * (size_t)row;
        return true;
    }
    row -= a.n_step;
    if (row >= count)
        return false;
    img = a.frames + frame * row;
    sums = a.sums2 + 2 * (size_t)row;
    return true;
}

__device__ __forceinline__ uint32_t gray_of(uint32_t r, uint32_t g, uint32_t b, int gray15)
{
    // vision.py:24 cv2.cvtColor(COLOR_RGB2GRAY), 8-bit fixed point
    return gray15 ? ((r * 9798u + g * 19235u + b * 3735u + 16384u) >> 15)
                  : ((r * 4899u + g * 9617u + b * 1868u + 8192u) >> 14);
}

__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) { return min(min(a, b), c); }
__device__ __forceinline__ uint32_t max3u(uint32_t a, uint32_t b, uint32_t c) { return max(max(a, b), c); }
__device__ __forceinline__ uint32_t med3u(uint32_t a, uint32_t b, uint32_t c)
{
    return max(min(a, b), min(max(a, b), c));
}

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1)
        return 0;
    if (i < 0)
        return -i;
    if (i >= n)
        return 2 * n - 2 - i;
    return i;
}

// dynamic LDS: gray[(kBand+4)][w] then med[(kBand+2)][w]
__global__ __launch_bounds__(kBlock) void focus_kernel(FocusArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint8_t *img;
    unsigned long long *sums;
    if (!focus_row(a, blockIdx.y, img, sums)) // block-uniform, before any barrier
        return;
    const int r0 = blockIdx.x * kBand;               // first output row
    const int r1 = min(r0 + kBand, a.h);             // one past last output row
    const int w = a.w, h = a.h;

    // median rows needed: reflect101 of [r0-1, r1] -> all inside [m0, m1)
    const int m0 = max(r0 - 1, 0);
    const int m1 = min(r1 + 1, h);
    // gray rows needed for those (replicate border): [g0, g1)
    const int g0 = max(m0 - 1, 0);
    const int g1 = min(m1 + 1, h);

    uint8_t *gray = lds;
    uint8_t *med = lds + (size_t)(kBand + 4) * w;

    const int grows = g1 - g0;
    if ((w & 3) == 0) {
        // 4 pixels (12 B = 3 dwords) per thread per step, coalesced
        const int quads = grows * (w >> 2);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(img + (size_t)g0 * w * 3);
        uint32_t *dst = reinterpret_cast<uint32_t *>(gray);
        for (int q = threadIdx.x; q < quads; q += kBlock) {
            uint32_t d0 = src[3 * q + 0], d1 = src[3 * q + 1], d2 = src[3 * q + 2];
            uint32_t ga = gray_of(d0 & 255u, (d0 >> 8) & 255u, (d0 >> 16) & 255u, a.gray15);
            uint32_t gb = gray_of(d0 >> 24, d1 & 255u, (d1 >> 8) & 255u, a.gray15);
            uint32_t gc = gray_of((d1 >> 16) & 255u, d1 >> 24, d2 & 255u, a.gray15);
            uint32_t gd = gray_of((d2 >> 8) & 255u, (d2 >> 16) & 255u, d2 >> 24, a.gray15);
            dst[q] = ga | (gb << 8) | (gc << 16) | (gd << 24);
        }
    } else {
        const int px = grows * w;
        const uint8_t *src = img + (size_t)g0 * w * 3;
        for (int i = threadIdx.x; i < px; i += kBlock)
            gray[i] = (uint8_t)gray_of(src[3 * i], src[3 * i + 1], src[3 * i + 2], a.gray15);
    }
    __syncthreads();

    // median rows [m0, m1): cv2.medianBlur(gray, 3), BORDER_REPLICATE
    const int mrows = m1 - m0;
    for (int i = threadIdx.x; i < mrows * w; i += kBlock) {
        const int my = i / w, x = i - my * w;
        const int y = m0 + my;
        const int ya = max(y - 1, 0) - g0, yb = y - g0, yc = min(y + 1, h - 1) - g0;
        const int xa = max(x - 1, 0), xc = min(x + 1, w - 1);
        uint32_t lo[3], mi[3], hi[3];
        const int xs[3] = {xa, x, xc};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uint32_t v0 = gray[ya * w + xs[c]], v1 = gray[yb * w + xs[c]], v2 = gray[yc * w + xs[c]];
            lo[c] = min3u(v0, v1, v2);
            mi[c] = med3u(v0, v1, v2);
            hi[c] = max3u(v0, v1, v2);
        }
        med[i] = (uint8_t)med3u(max3u(lo[0], lo[1], lo[2]), med3u(mi[0], mi[1], mi[2]),
                                min3u(hi[0], hi[1], hi[2]));
    }
    __syncthreads();

    // Laplacian rows [r0, r1): cv2.Laplacian(m, CV_8U), ksize 1, BORDER_REFLECT_101
    uint32_t s1 = 0;
    unsigned long long s2 = 0;
    const int orows = r1 - r0;
    for (int i = threadIdx.x; i < orows * w; i += kBlock) {
        const int oy = i / w, x = i - oy * w;
        const int y = r0 + oy;
        const int yu = reflect101(y - 1, h) - m0, yd = reflect101(y + 1, h) - m0, yc = y - m0;
        const int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
        int v = (int)med[yu * w + x] + (int)med[yd * w + x] + (int)med[yc * w + xl] +
                (int)med[yc * w + xr] - 4 * (int)med[yc * w + x];
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        s1 += (uint32_t)v;
        s2 += (uint32_t)(v * v);
    }

    // wave reduction (64 lanes) then one atomic pair per wave
    unsigned long long t1 = s1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        t1 += __shfl_down(t1, off, 64);
        s2 += __shfl_down(s2, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&sums[0], t1);
        atomicAdd(&sums[1], s2);
    }
}

// ---------------------------------------------------------------------------
// focus_kernel_quad: the same chain for widths that are a multiple of 4, four pixels per
// thread and dword LDS traffic (the byte-per-thread kernel above spends ~175 lane
// instructions per pixel, mostly LDS byte reads and index arithmetic).
//   stage 1  12 B (4 pixels) per lane from HBM -> 4 gray bytes -> one ds_write_b32
//   stage 2  3 rows x 3 dwords from LDS -> 6 columns sorted once (min3/med3/max3), each of
//            the 4 medians from 3 neighbouring sorted columns -> one ds_write_b32
//   stage 3  up / down dwords + 3 centre dwords -> 4 Laplacians, saturate, sums
// Bands of kBandQ rows per block: halo 4 rows in kBandQ + 4 (12.5 % at 32).
// ---------------------------------------------------------------------------
constexpr int kBandQ = 32;

__device__ __forceinline__ uint32_t byte_of(uint32_t v, int i) { return (v >> (8 * i)) & 255u; }

__global__ __launch_bounds__(kBlock) void focus_kernel_quad(FocusArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint8_t *img;
    unsigned long long *sums;
    if (!focus_row(a, blockIdx.y, img, sums)) // block-uniform, before any barrier
        return;
    const int w = a.w, h = a.h, wq = a.w >> 2;
    const int r0 = blockIdx.x * kBandQ, r1 = min(r0 + kBandQ, h);
    const int m0 = max(r0 - 1, 0), m1 = min(r1 + 1, h); // median rows needed
    const int g0 = max(m0 - 1, 0), g1 = min(m1 + 1, h); // gray rows needed

    uint32_t *gray = reinterpret_cast<uint32_t *>(lds);                              // [(kBandQ+4)][wq]
    uint32_t *med = reinterpret_cast<uint32_t *>(lds + (size_t)(kBandQ + 4) * w);    // [(kBandQ+2)][wq]
    {
        const int quads = (g1 - g0) * wq;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(img + (size_t)g0 * w * 3);
        for (int q = threadIdx.x; q < quads; q += kBlock) {
            const uint32_t d0 = src[3 * q + 0], d1 = src[3 * q + 1], d2 = src[3 * q + 2];
            const uint32_t ga = gray_of(d0 & 255u, (d0 >> 8) & 255u, (d0 >> 16) & 255u, a.gray15);
            const uint32_t gb = gray_of(d0 >> 24, d1 & 255u, (d1 >> 8) & 255u, a.gray15);
            const uint32_t gc = gray_of((d1 >> 16) & 255u, d1 >> 24, d2 & 255u, a.gray15);
            const uint32_t gd = gray_of((d2 >> 8) & 255u, (d2 >> 16) & 255u, d2 >> 24, a.gray15);
            gray[q] = ga | (gb << 8) | (gc << 16) | (gd << 24);
        }
    }
    __syncthreads();

    // median rows [m0, m1): cv2.medianBlur(gray, 3), BORDER_REPLICATE
    {
        const int quads = (m1 - m0) * wq;
        for (int i = threadIdx.x; i < quads; i += kBlock) {
            const int my = i / wq, q = i - my * wq;
            const int y = m0 + my;
            const int rows[3] = {max(y - 1, 0) - g0, y - g0, min(y + 1, h - 1) - g0};
            uint32_t lo[6], mi[6], hi[6];
            uint32_t c[3][6];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const uint32_t *row = gray + rows[r] * wq;
                const uint32_t mid = row[q];
                const uint32_t left = q > 0 ? row[q - 1] >> 24 : (mid & 255u);         // replicate
                const uint32_t right = q < wq - 1 ? (row[q + 1] & 255u) : (mid >> 24); // replicate
                c[r][0] = left;
                c[r][1] = byte_of(mid, 0);
                c[r][2] = byte_of(mid, 1);
                c[r][3] = byte_of(mid, 2);
                c[r][4] = byte_of(mid, 3);
                c[r][5] = right;
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                lo[j] = min3u(c[0][j], c[1][j], c[2][j]);
                mi[j] = med3u(c[0][j], c[1][j], c[2][j]);
                hi[j] = max3u(c[0][j], c[1][j], c[2][j]);
            }
            uint32_t out = 0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint32_t m = med3u(max3u(lo[p], lo[p + 1], lo[p + 2]), med3u(mi[p], mi[p + 1], mi[p + 2]),
                                         min3u(hi[p], hi[p + 1], hi[p + 2]));
                out |= m << (8 * p);
            }
            med[i] = out;
        }
    }
    __syncthreads();

    // Laplacian rows [r0, r1): cv2.Laplacian(m, CV_8U), ksize 1, BORDER_REFLECT_101
    uint32_t s1 = 0;
    unsigned long long s2 = 0;
    {
        const int quads = (r1 - r0) * wq;
        for (int i = threadIdx.x; i < quads; i += kBlock) {
            const int oy = i / wq, q = i - oy * wq;
            const int y = r0 + oy;
            const uint32_t up = med[(reflect101(y - 1, h) - m0) * wq + q];
            const uint32_t dn = med[(reflect101(y + 1, h) - m0) * wq + q];
            const uint32_t *row = med + (y - m0) * wq;
            const uint32_t mid = row[q];
            // reflect-101: x = -1 -> 1, x = w -> w - 2 (w >= 4 here)
            const uint32_t left = q > 0 ? row[q - 1] >> 24 : byte_of(mid, 1);
            const uint32_t right = q < wq - 1 ? (row[q + 1] & 255u) : byte_of(mid, 2);
            const uint32_t cc[6] = {left, byte_of(mid, 0), byte_of(mid, 1), byte_of(mid, 2), byte_of(mid, 3), right};
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                int v = (int)(byte_of(up, p) + byte_of(dn, p) + cc[p] + cc[p + 2]) - 4 * (int)cc[p + 1];
                v = v < 0 ? 0 : (v > 255 ? 255 : v);
                s1 += (uint32_t)v;
                s2 += (uint32_t)(v * v);
            }
        }
    }

    unsigned long long t1 = s1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        t1 += __shfl_down(t1, off, 64);
        s2 += __shfl_down(s2, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&sums[0], t1);
        atomicAdd(&sums[1], s2);
    }
}

// population variance from exact integer sums: (N*S2 - S1^2) / N^2
__global__ void focus_finalize(const unsigned long long *sums, double *var, int n, unsigned long long npix)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n)
        return;
    const unsigned long long s1 = sums[2 * e], s2 = sums[2 * e + 1];
    const unsigned __int128 num = (unsigned __int128)npix * s2 - (unsigned __int128)s1 * s1;
    const double dn = (double)npix;
    var[e] = (double)(unsigned long long)num / (dn * dn);
}

// ---------------------------------------------------------------------------
// seeding: state[i] = J^(first + i) * s_init over GF(2), J = 2^64-step jump matrix.
// mats[k] = J^(2^k) as 128 columns of 128 bits.  A wave owns 64*R consecutive
// states: lane l starts at base + l and strides by 64 (= mats[6]), so every store
// instruction writes 1 KiB contiguous.
// ---------------------------------------------------------------------------
constexpr int kSeedMats = 48;
constexpr int kSeedRun = 32;

__device__ __forceinline__ ulonglong2 gf2_matvec(const ulonglong2 *__restrict__ cols, ulonglong2 v)
{
    unsigned long long r0 = 0, r1 = 0;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {
        const ulonglong2 c = cols[j];
        const unsigned long long m = 0ull - ((v.x >> j) & 1ull);
        r0 ^= c.x & m;
        r1 ^= c.y & m;
    }
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {
        const ulonglong2 c = cols[64 + j];
        const unsigned long long m = 0ull - ((v.y >> j) & 1ull);
        r0 ^= c.x & m;
        r1 ^= c.y & m;
    }
    return make_ulonglong2(r0, r1);
}

__global__ __launch_bounds__(kBlock) void seed_kernel(ulonglong2 *states, unsigned long long n,
                                                     unsigned long long first, ulonglong2 s_init,
                                                     const ulonglong2 *__restrict__ mats)
{
    const unsigned long long gid = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    const unsigned long long wave = gid >> 6;
    const unsigned lane = (unsigned)(gid & 63);
    const unsigned long long base = wave * (64ull * kSeedRun);
    if (base >= n)
        return;
    unsigned long long i = base + lane;
    const unsigned long long gidx = first + i;

    ulonglong2 s = s_init;
    for (int k = 0; k < kSeedMats; ++k) {
        // wave-level skip keeps the matrix loads scalar and skips unused high bits
        const bool bit = (gidx >> k) & 1ull;
        if (__any(bit)) {
            const ulonglong2 t = gf2_matvec(mats + k * 128, s);
            if (bit)
                s = t;
        }
    }
    for (int j = 0; j < kSeedRun; ++j) {
        if (i < n)
            states[i] = s;
        i += 64;
        if (j + 1 < kSeedRun)
            s = gf2_matvec(mats + 6 * 128, s);
    }
}

} // namespace rf
