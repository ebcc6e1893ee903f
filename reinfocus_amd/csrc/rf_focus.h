// rf_focus.h -- the focus measure of vision.focus_value (vision.py:23-25): cvtColor -> medianBlur(3) -> Laplacian(CV_8U) ->
// per-env integer sums -> ndarray.var() from the exact sums.
//   focus_kernel_roll   widths that are multiples of 4 (>= 8): four pixels per lane, rows rolling through registers (round 6)
//   focus_kernel        any other width: a byte per thread over column tiles of 512, the chain through LDS
//   focus_kernel_quad   the round-2 kernel (whole rows staged in LDS, widths up to 936 px): A/B runs and tests only
//   focus_finalize      the variance (rf_math.h variance_from_sums)
#pragma once

#include "rf_common.h"

namespace rf {

// ---------------------------------------------------------------------------
// focus: gray -> median3x3 (replicate) -> Laplacian (reflect-101, sat u8) -> sums
// One block per (row band, column tile, env).  Integer/byte work: 3 B/pixel read.
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

// Column tiles of the byte-per-thread kernel: any frame width (vision.py:11-39 scores whatever it is handed); a block
// takes kBand rows x kTileB columns, two more gray columns and one more median column on either side.
constexpr int kTileB = 512;

// dynamic LDS: gray[(kBand+4)][tw+4] then med[(kBand+2)][tw+2]  (tw = columns of the block's tile)
__global__ __launch_bounds__(kBlock) void focus_kernel(FocusArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint8_t *img;
    unsigned long long *sums;
    if (!focus_row(a, blockIdx.y, img, sums)) // block-uniform, before any barrier
        return;
    const int w = a.w, h = a.h;
    const int tiles = (w + kTileB - 1) / kTileB;
    const int band = blockIdx.x / tiles, tile = blockIdx.x - band * tiles;
    const int x0 = tile * kTileB, tw = min(kTileB, w - x0); // output columns [x0, x0 + tw)
    const int r0 = band * kBand;                     // first output row
    const int r1 = min(r0 + kBand, h);               // one past last output row

    // median rows needed: reflect101 of [r0-1, r1] -> all inside [m0, m1)
    const int m0 = max(r0 - 1, 0);
    const int m1 = min(r1 + 1, h);
    // gray rows needed for those (replicate border): [g0, g1)
    const int g0 = max(m0 - 1, 0);
    const int g1 = min(m1 + 1, h);
    // gray columns x0 - 2 + j, j < gw (clamped into the frame: BORDER_REPLICATE); median columns x0 - 1 + j, j < mw
    const int gw = tw + 4, mw = tw + 2;

    uint8_t *gray = lds;
    uint8_t *med = lds + (size_t)(kBand + 4) * (kTileB + 4);

    const int grows = g1 - g0;
    for (int i = threadIdx.x; i < grows * gw; i += kBlock) {
        const int gy = i / gw, j = i - gy * gw;
        const int x = min(max(x0 - 2 + j, 0), w - 1);
        const uint8_t *src = img + ((size_t)(g0 + gy) * w + x) * 3;
        gray[i] = (uint8_t)gray_of(src[0], src[1], src[2], a.gray15);
    }
    __syncthreads();

    // median rows [m0, m1), columns x0 - 1 ... x0 + tw: cv2.medianBlur(gray, 3), BORDER_REPLICATE
    const int mrows = m1 - m0;
    for (int i = threadIdx.x; i < mrows * mw; i += kBlock) {
        const int my = i / mw, j = i - my * mw; // gray column of the centre: j + 1
        const int y = m0 + my;
        const int ya = max(y - 1, 0) - g0, yb = y - g0, yc = min(y + 1, h - 1) - g0;
        uint32_t lo[3], mi[3], hi[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uint32_t v0 = gray[ya * gw + j + c], v1 = gray[yb * gw + j + c], v2 = gray[yc * gw + j + c];
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
    for (int i = threadIdx.x; i < orows * tw; i += kBlock) {
        const int oy = i / tw, xl0 = i - oy * tw;
        const int y = r0 + oy, x = x0 + xl0;
        const int yu = reflect101(y - 1, h) - m0, yd = reflect101(y + 1, h) - m0, yc = y - m0;
        const int xc = xl0 + 1, xl = reflect101(x - 1, w) - (x0 - 1), xr = reflect101(x + 1, w) - (x0 - 1);
        int v = (int)med[yu * mw + xc] + (int)med[yd * mw + xc] + (int)med[yc * mw + xl] +
                (int)med[yc * mw + xr] - 4 * (int)med[yc * mw + xc];
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

// ---------------------------------------------------------------------------
// focus_kernel_roll: the chain for widths that are a multiple of 4 (and >= 8) with everything between the load and the
// sums in registers -- no LDS, no barrier, any frame size.
//   * A lane owns a column of four pixels (12 bytes = one global_load_dwordx3 per row) and walks down a band of
//     `band` rows; its horizontal neighbours are the lanes next to it (v_mov_b32_dpp wave_shr / wave_shl), so the
//     lanes of a wave are consecutive column groups of one frame -- wrapping from the end of one band of rows to the
//     start of the next: the groups at a frame's edge take their border value instead of the neighbour lane's.
//     Where a wave boundary does not fall on a frame edge (64 % groups-per-row != 0) the first and the last lane of
//     every wave are halo lanes: they compute, their sums are not counted, and waves overlap by two groups (HALO).
//   * gray: v_dot4_u32_u8 with the 15-bit (14-bit) coefficients split into bytes -- (256 hi.p + lo.p + round) >> shift,
//     four instructions per pixel, the pixel's three bytes picked out of the row's dwords by v_alignbyte_b32.
//   * median: the three gray rows in flight are sorted per column (v_min3 / v_med3 / v_max3_u32), a pixel's median is
//     med3(max3 of the minima, med3 of the medians, min3 of the maxima) of its three columns.
//   * Laplacian of the row above the newest median row, clamped by v_med3_i32, summed per lane; one wave reduction and
//     one atomic pair per wave at the end.
// Rows above / below the frame: gray rows replicate (the row index is clamped: medianBlur's BORDER_REPLICATE), the
// Laplacian's reflect-101 takes the row below for the row above at y = 0 (and the reverse at y = h - 1).
// A step costs 129 vector instructions for 4 x 62 (64) pixels, nearly all of gfx950's slow issue class (0.24 per cycle and
// SIMD): the headline's 4625 frames of 256^2 in 0.24 ms = 3.9 TB/s of the 3 B/pixel it reads (profiles/r06_ab.txt section 1).
// ---------------------------------------------------------------------------
struct GrayDot {                // RGB2GRAY by v_dot4_u32_u8: coefficient bytes for a pixel in bytes 0..2 / 1..3 of a dword
    uint32_t hi_lo3, lo_lo3;    // high / low bytes of the three coefficients at byte positions 0, 1, 2
    uint32_t hi_hi3, lo_hi3;    // ... at byte positions 1, 2, 3
    uint32_t round, shift;
};

inline GrayDot gray_dot(int gray15)
{
    const uint32_t c[3] = {gray15 ? 9798u : 4899u, gray15 ? 19235u : 9617u, gray15 ? 3735u : 1868u};
    GrayDot d;
    d.hi_lo3 = (c[0] >> 8) | ((c[1] >> 8) << 8) | ((c[2] >> 8) << 16);
    d.lo_lo3 = (c[0] & 255u) | ((c[1] & 255u) << 8) | ((c[2] & 255u) << 16);
    d.hi_hi3 = d.hi_lo3 << 8;
    d.lo_hi3 = d.lo_lo3 << 8;
    d.round = gray15 ? 16384u : 8192u;
    d.shift = gray15 ? 15u : 14u;
    return d;
}

__device__ __forceinline__ uint32_t v_min3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t v_med3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t v_max3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// lane i takes lane i - 1's / lane i + 1's value (lane 0 / 63: zero)
__device__ __forceinline__ uint32_t from_left_lane(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t from_right_lane(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}

struct __attribute__((packed, aligned(4))) Rgb4 { // four pixels
    uint32_t d0, d1, d2;
};
struct GrayRow6 { // gray of columns x - 1 ... x + 4
    uint32_t v[6];
};
struct MedRow6 { // medians of columns x - 1 ... x + 4
    uint32_t v[6];
};

constexpr int kRollBandMax = 64; // rows per band: a lane's sum of squares stays below 2^32 / 64

struct FocusRollArgs {
    FocusArgs f;
    GrayDot dot;
    int band;   // rows per band
    int groups; // w / 4
    int bands;  // ceil(h / band)
};

template <bool HALO>
__global__ __launch_bounds__(64) void focus_kernel_roll(FocusRollArgs ra)
{
    const FocusArgs &a = ra.f;
    const uint8_t *img;
    unsigned long long *sums;
    if (!focus_row(a, blockIdx.y, img, sums)) // wave-uniform
        return;
    constexpr int kEff = HALO ? 62 : 64;
    const int w = a.w, h = a.h, G = ra.groups, R = ra.band;
    const int lane = threadIdx.x;
    const int T = ra.bands * G;
    const int t = (int)blockIdx.x * kEff + lane - (HALO ? 1 : 0);
    const bool in_range = t >= 0 && t < T;
    const int tc = min(max(t, 0), T - 1);
    const int band = tc / G, cg = tc - band * G;
    const bool counted = in_range && !(HALO && (lane == 0 || lane == 63));
    const bool left_edge = cg == 0, right_edge = cg == G - 1;
    const int y0 = band * R;
    const uint8_t *const col = img + (size_t)cg * 12;
    const size_t pitch = (size_t)w * 3;
    auto load_row = [&](int s) { // the row of step s: y0 + s - 2, clamped into the frame
        const int y = min(max(y0 + s - 2, 0), h - 1);
        return *reinterpret_cast<const Rgb4 *>(col + pitch * (size_t)y);
    };
    const GrayDot &k = ra.dot;
    auto gray_row = [&](const Rgb4 &p, GrayRow6 &g) {
        const uint32_t p1 = __builtin_amdgcn_alignbyte(p.d1, p.d0, 3), p2 = __builtin_amdgcn_alignbyte(p.d2, p.d1, 2);
        // (256 H + L + round) >> shift == (H + ((L + round) >> 8)) >> (shift - 8): H is an integer, so the fraction the inner
        // shift drops cannot carry into the outer floor -- and the shifts right issue on gfx950's fast class, the shift-add does not
        const uint32_t l0 = __builtin_amdgcn_udot4(p.d0, k.lo_lo3, k.round, false) >> 8;
        const uint32_t l1 = __builtin_amdgcn_udot4(p1, k.lo_lo3, k.round, false) >> 8;
        const uint32_t l2 = __builtin_amdgcn_udot4(p2, k.lo_lo3, k.round, false) >> 8;
        const uint32_t l3 = __builtin_amdgcn_udot4(p.d2, k.lo_hi3, k.round, false) >> 8;
        g.v[1] = __builtin_amdgcn_udot4(p.d0, k.hi_lo3, l0, false) >> (k.shift - 8);
        g.v[2] = __builtin_amdgcn_udot4(p1, k.hi_lo3, l1, false) >> (k.shift - 8);
        g.v[3] = __builtin_amdgcn_udot4(p2, k.hi_lo3, l2, false) >> (k.shift - 8);
        g.v[4] = __builtin_amdgcn_udot4(p.d2, k.hi_hi3, l3, false) >> (k.shift - 8);
        const uint32_t gl = from_left_lane(g.v[4]), gr = from_right_lane(g.v[1]);
        g.v[0] = left_edge ? g.v[1] : gl;  // BORDER_REPLICATE
        g.v[5] = right_edge ? g.v[4] : gr;
    };
    auto median_row = [&](const GrayRow6 &ga, const GrayRow6 &gb, const GrayRow6 &gc, MedRow6 &m) {
        uint32_t lo[6], mi[6], hi[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            lo[j] = v_min3(ga.v[j], gb.v[j], gc.v[j]);
            mi[j] = v_med3(ga.v[j], gb.v[j], gc.v[j]);
            hi[j] = v_max3(ga.v[j], gb.v[j], gc.v[j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            m.v[i + 1] = v_med3(v_max3(lo[i], lo[i + 1], lo[i + 2]), v_med3(mi[i], mi[i + 1], mi[i + 2]),
                                v_min3(hi[i], hi[i + 1], hi[i + 2]));
        const uint32_t ml = from_left_lane(m.v[4]), mr = from_right_lane(m.v[1]);
        m.v[0] = left_edge ? m.v[2] : ml;  // BORDER_REFLECT_101
        m.v[5] = right_edge ? m.v[3] : mr;
    };
    uint32_t s1 = 0, s2 = 0;
    // step s: gray row y0 + s - 2 (gn), median row y0 + s - 3 (mn, from the three newest gray rows), Laplacian of row
    // y0 + s - 4 (median rows mu above, mc, mn below)
    auto step = [&](int s, const Rgb4 &p, const GrayRow6 &g2, const GrayRow6 &g1, GrayRow6 &gn, const MedRow6 &mu,
                    const MedRow6 &mc, MedRow6 &mn) {
        gray_row(p, gn);
        median_row(g2, g1, gn, mn);
        const int y = y0 + s - 4;
        const bool top = y == 0, bottom = y == h - 1;
        const bool valid = counted && s >= 4 && s < R + 4 && y < h;
        // the frame's first and last row (reflect-101: the row below stands in for the row above, and the reverse) concern
        // one lane in hundreds: the selects run only in the steps in which some lane of the wave is there
        uint32_t up[4], dn[4];
        if (__builtin_amdgcn_ballot_w64(top || bottom) != 0) { // wave-uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                up[i] = top ? mn.v[i + 1] : mu.v[i + 1];
                dn[i] = bottom ? mu.v[i + 1] : mn.v[i + 1];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                up[i] = mu.v[i + 1];
                dn[i] = mn.v[i + 1];
            }
        }
        uint32_t t1 = 0, t2 = 0;
#pragma unroll
        for (int i = 1; i <= 4; ++i) {
            int v = (int)(up[i - 1] + dn[i - 1] + mc.v[i - 1] + mc.v[i + 1]) - 4 * (int)mc.v[i];
            v = min(max(v, 0), 255);
            t1 += (uint32_t)v;
            t2 += (uint32_t)(v * v);
        }
        s1 += valid ? t1 : 0u; // (one select per sum and step, not per pixel)
        s2 += valid ? t2 : 0u;
    };
    GrayRow6 ga{}, gb{}, gc{};
    MedRow6 ma{}, mb{}, mc{};
    // three steps per trip (the rings of gray and median rows rotate through their three roles); the R + 4 steps are
    // rounded up to whole trips -- a step beyond the band counts nothing -- so that a trip is one basic block
    const int steps = R + 4;
    Rgb4 cur = load_row(0);
    for (int s = 0; s < steps; s += 3) {
        Rgb4 nxt = load_row(s + 1);
        step(s, cur, ga, gb, gc, ma, mb, mc);
        cur = load_row(s + 2);
        step(s + 1, nxt, gb, gc, ga, mb, mc, ma);
        nxt = load_row(s + 3);
        step(s + 2, cur, gc, ga, gb, mc, ma, mb);
        cur = nxt;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off, 64);
        s2 += __shfl_down(s2, off, 64);
    }
    if (lane == 0) {
        atomicAdd(&sums[0], (unsigned long long)s1);
        atomicAdd(&sums[1], (unsigned long long)s2);
    }
}

// population variance from exact integer sums: (N*S2 - S1^2) / N^2
__global__ void focus_finalize(const unsigned long long *sums, double *var, int n, unsigned long long npix)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n)
        return;
    var[e] = variance_from_sums(npix, sums[2 * e], sums[2 * e + 1]);
}

} // namespace rf
