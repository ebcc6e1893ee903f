// rf_render.h -- render_kernel: FastRenderer._device_render (graphics/render.py:190-246), one pixel per thread,
// rejection loops inside the wave.  The kernel of launches of few blocks (rf_abi_render.hip few_blocks: bound by the
// latency of a sample, where barriers cost more than they save) and of camera frames other than the canonical one.
// The render kernels are VALU-bound (64-bit integer RNG + rejection loops), not HBM-bound: 35 B/pixel of traffic
// against ~10^4 lane-ops/pixel at 16 spp.  No MFMA: nothing here is a contraction.
#pragma once

#include "rf_common.h"

namespace rf {

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
    FrameConst fc;   // frame sizes in the forms the jittered coordinates use (rf_math.h)
    // render_kernel_coop2<..., TWO = true> (the environment step as one launch, rf_abi_env.hip enqueue_env_step): the
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
            render_pixel<AXIS, POW2>(g, x, y, a.spp, a.fc, env, a.cs, a.tab, cr, cg, cb);
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

} // namespace rf
