// rf_focus.h -- the focus measure of vision.focus_value (vision.py:23-25):
//   focus_kernel / focus_kernel_quad   cvtColor -> medianBlur(3) -> Laplacian(CV_8U) -> per-env integer sums
//   focus_finalize                     ndarray.var() from the exact sums
#pragma once

#include "rf_common.h"

namespace rf {

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

} // namespace rf
