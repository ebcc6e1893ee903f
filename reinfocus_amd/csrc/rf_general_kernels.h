// rf_general_kernels.h -- render_general_kernel: the literal kernel of the general renderer (arithmetic: rf_general.h)
#pragma once

#include "rf_common.h"
#include "rf_general.h"

namespace rf {

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
    // where a pixel's state is read from: `states` itself, or the context's copy of the freshly seeded array -- every call
    // of the general renderer starts from seed-0 states (render.py:115), and reading them from the copy saves the call a
    // device-to-device copy of 32 bytes per pixel in front of the kernel.  A pixel that abstains writes nothing, and the
    // fix-up kernel reads its state from here as well.
    const ulonglong2 *states_in;
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
        const ulonglong2 st = a.states_in[pix];
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

} // namespace rf
