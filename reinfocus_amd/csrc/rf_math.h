// rf_math.h -- per-sample arithmetic of the render kernel (gfx950).
//
// Every function here is bit-exact with the reference's numba typing
// (numpy-1.26 promotion, IEEE-754, no FMA contraction), see DESIGN.md
// "Arithmetic contract".  The file is plain C++ so that tests/hostsim can compile
// the very same arithmetic for the host and compare it with the oracle on the CPU
// (test infrastructure only -- the product always runs these on the GPU).
//
// Build flags that matter: -ffp-contract=off (HIP defaults to fast contraction)
// and HIP's default correctly-rounded f32 divide/sqrt.  Where an FMA is used it is
// written explicitly and is provably equal to the unfused expression.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define RF_HD __host__ __device__ __forceinline__
#else
#define RF_HD inline
#endif

namespace rf {

// ---------------------------------------------------------------------------
// xoroshiro128+ (numba.cuda.random xoroshiro128p_next; graphics/random.py:33)
// ---------------------------------------------------------------------------
struct Rng {
    uint64_t s0, s1;
};

RF_HD uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

RF_HD uint64_t rng_next(Rng &g)
{
    uint64_t s0 = g.s0, s1 = g.s1;
    uint64_t r = s0 + s1;
    s1 ^= s0;
    g.s0 = rotl64(s0, 55) ^ s1 ^ (s1 << 14);
    g.s1 = rotl64(s1, 36);
    return r;
}

RF_HD int clz32(uint32_t x) { return x ? __builtin_clz(x) : 32; } // v_ffbh_u32

RF_HD float ldexp_pow2(float x, int e)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ldexpf(x, e);
#else
    return __builtin_ldexpf(x, e);
#endif
}

// float32( float64(r >> 11) * 2^-53 )   (numba uint64_to_unit_float32).
// The f64 product is exact, so the only rounding is the final f64->f32 RNE.
// Literal form:
RF_HD float unit_f32_literal(uint64_t r)
{
    return (float)((double)(r >> 11) * (1.0 / 9007199254740992.0));
}

// Integer form, same value for every r: normalise the 53 kept bits, fold the bits
// below the top 32 into a sticky bit (legal because a normalised 32-bit word keeps
// its round bit at position >= 7), let v_cvt_f32_u32 do the single RNE, scale by an
// exact power of two.
RF_HD float unit_f32_int(uint64_t r)
{
    uint32_t hi = (uint32_t)(r >> 32);
    uint32_t lo = (uint32_t)r & 0xFFFFF800u; // the 11 bits numba shifts out never count
    int lz = hi ? clz32(hi) : 32;
    uint64_t y64 = (((uint64_t)hi << 32) | lo) << (lz & 63);
    if (lz == 32)
        y64 = (uint64_t)lo << 32;
    uint32_t y = (uint32_t)(y64 >> 32);
    uint32_t rest = (uint32_t)y64;
    y |= (rest != 0u) ? 1u : 0u;
    return ldexp_pow2((float)y, -32 - lz);
}

#ifndef RF_UNIFORM_LITERAL
RF_HD float rng_uniform(Rng &g) { return unit_f32_int(rng_next(g)); }
#else
RF_HD float rng_uniform(Rng &g) { return unit_f32_literal(rng_next(g)); }
#endif

// ---------------------------------------------------------------------------
// scene parameters
// ---------------------------------------------------------------------------
struct CamStatic { // camera.py:39-52 FastGpuCameras minus the per-env array
    float ox, oy, oz;
    float ux, uy, uz;
    float vx, vy, vz;
    double lens_radius; // numpy.float64
};

struct CamDyn { // camera.py:54-56 FCAM_DYNAMIC_*
    float llx, lly, llz;
    float hx, hy, hz;
    float vx, vy, vz;
};

struct Rect { // rectangle.py:21-23 FH_RADIUS, FH_ZPOS
    float half, z;
};

// physics.py:47-64 colour_checkerboard with uf = (32, 32) (rectangle.py:145).
// Only the sign of sin(32*pi*u) is used.  For u in [0,1] as f32, m = 32*u is exact;
// when m is not an integer sign = (-1)^floor(m) (the f64 rounding of 32*pi*u moves
// the argument by < 5e-15, the nearest integer is >= 1.9e-6 away); when m is an
// integer k the sign depends on how fl64(32*pi_d*u) rounds against k*pi, and is read
// from a 33-entry table the host fills with the real libm sin (neg_mask bit k set
// <=> sin < 0; sin == 0 only for k == 0).
struct CheckerTable {
    uint64_t neg_mask;
};

// returns 0 if sin == 0, else +1 / -1
RF_HD int checker_sign(float u, const CheckerTable &tab)
{
    float m = u * 32.0f; // exact (power of two), u in [0, 1]
    float fl = __builtin_floorf(m);
    int k = (int)fl;
    int neg;
    if (m == fl) {
        if (k == 0)
            return 0;
        neg = (int)((tab.neg_mask >> k) & 1u);
    } else {
        neg = k & 1;
    }
    return neg ? -1 : 1;
}

// ---------------------------------------------------------------------------
// one sample: camera.get_ray (camera.py:307-350) + physics.fast_find_colour
// (physics.py:148-193) with rectangle.fast_hit (rectangle.py:102-148).
// Literal/general version: any camera frame.  Returns the sample colour.
// ---------------------------------------------------------------------------
RF_HD float add2(float a, float b) { return (0.0f + a) + b; }
RF_HD float add3(float a, float b, float c) { return ((0.0f + a) + b) + c; }

RF_HD float sq_len(float a, float b, float c)
{
    // vector.py:300-314: float32(v**2) per component, summed left to right in f32
    float aa = a * a, bb = b * b, cc = c * c;
    return (aa + bb) + cc;
}

struct Colour {
    float r, g, b;
};

RF_HD void disc_sample(Rng &g, float &p0, float &p1)
{
    // camera.py:229-252.  a*2f is exact, so fma(a, 2, -1) == RN(a*2f - 1f).
    for (;;) {
        float a = rng_uniform(g);
        float b = rng_uniform(g);
        p0 = __builtin_fmaf(a, 2.0f, -1.0f);
        p1 = __builtin_fmaf(b, 2.0f, -1.0f);
        float d0 = p0 * p0, d1 = p1 * p1;
        if (d0 + d1 < 1.0f)
            return;
    }
}

RF_HD void sphere_sample(Rng &g, float &q0, float &q1, float &q2)
{
    // physics.py:20-44
    for (;;) {
        float a = rng_uniform(g);
        float b = rng_uniform(g);
        float c = rng_uniform(g);
        q0 = __builtin_fmaf(a, 2.0f, -1.0f);
        q1 = __builtin_fmaf(b, 2.0f, -1.0f);
        q2 = __builtin_fmaf(c, 2.0f, -1.0f);
        if (sq_len(q0, q1, q2) < 1.0f)
            return;
    }
}

// physics.py:183-193: sky colour of direction d times attenuation.
// T = 0.5*(ud.y + 1.0) in f64; 0.5*x is exact so T == fma(ud.y, 0.5, 0.5).
RF_HD double sky_t(float d0, float d1, float d2)
{
    float sq = sq_len(d0, d1, d2);
    // float32(math.sqrt(sq)): an f64 sqrt rounded to f32 equals the correctly rounded
    // f32 sqrt (53 >= 2*24+2).  Needs -fhip-fp32-correctly-rounded-divide-sqrt (HIP's
    // default; __fsqrt_rn would be the *native* 1-ulp sqrt).
    float len = __builtin_sqrtf(sq);
    float inv = 1.0f / len;
    float ud1 = d1 * inv;
    return __builtin_fma((double)ud1, 0.5, 0.5);
}

RF_HD float sky_channel(double t, float white, float k)
{
    float blue = (float)((double)k * t);
    return add2(white, blue);
}

RF_HD Colour sample_general(Rng &g, const CamDyn &cd, const CamStatic &cs, const Rect &rc,
                            float s, float t, const CheckerTable &tab)
{
    float p0, p1;
    disc_sample(g, p0, p1);
    double rd0 = (double)p0 * cs.lens_radius;
    double rd1 = (double)p1 * cs.lens_radius;
    float ox = add3(cs.ox, (float)((double)cs.ux * rd0), (float)((double)cs.vx * rd1));
    float oy = add3(cs.oy, (float)((double)cs.uy * rd0), (float)((double)cs.vy * rd1));
    float oz = add3(cs.oz, (float)((double)cs.uz * rd0), (float)((double)cs.vz * rd1));
    float dx = add3(cd.llx, cd.hx * s, cd.vx * t) - ox;
    float dy = add3(cd.lly, cd.hy * s, cd.vy * t) - oy;
    float dz = add3(cd.llz, cd.hz * s, cd.vz * t) - oz;

    // rectangle.py:125-136
    float tt = (rc.z - oz) / dz;
    bool hit = !(tt < 0.001f || tt > 1000000.0f);
    float px = 0.f, py = 0.f;
    if (hit) {
        px = add2(ox, dx * tt);
        py = add2(oy, dy * tt);
        hit = !(px < -rc.half || px > rc.half || py < -rc.half || py > rc.half);
    }

    float ar = 1.0f, ag = 1.0f, ab = 1.0f;
    if (hit) {
        float den = rc.half - (-rc.half);
        float u = (px - (-rc.half)) / den;
        float v = (py - (-rc.half)) / den;
        float q0, q1, q2;
        sphere_sample(g, q0, q1, q2);
        dx = add2(0.0f, q0);
        dy = add2(0.0f, q1);
        dz = add2(1.0f, q2);
        int su = checker_sign(u, tab), sv = checker_sign(v, tab);
        bool red = (su * sv) > 0;
        ar = red ? 1.0f : 0.0f;
        ag = red ? 0.0f : 1.0f;
        ab = 0.0f;
    }
    double T = sky_t(dx, dy, dz);
    float white = (float)(1.0 - T);
    Colour c;
    c.r = sky_channel(T, white, 0.5f) * ar;
    c.g = sky_channel(T, white, 0.7f) * ag;
    c.b = sky_channel(T, white, 1.0f) * ab;
    return c;
}

// s = float32((x + xi) / w): sum and quotient in f64 (int + f32 -> f64).
RF_HD float pixel_coord_literal(int x, float xi, int w)
{
    return (float)(((double)x + (double)xi) / (double)w);
}

// For w a power of two the same value needs no f64: the f64 sum is either exact or
// the addend is < 2^-28 ulp-wise irrelevant, so RN32(RN64(x+xi)) ==
// RN32(x+xi) == the f32 add, and the division is an exact scaling.
RF_HD float pixel_coord_pow2(int x, float xi, float inv_w)
{
    return ((float)x + xi) * inv_w;
}

// ---------------------------------------------------------------------------
// one pixel of FastRenderer._device_render (render.py:210-246): spp samples
// accumulated in f32.  Shared verbatim by the gfx950 kernel (rf_kernels.h) and the
// CPU-side simulation used only by tests (tests/hostsim).
//
// AXIS: the camera frame is the canonical one FastCameras() always produces
//       (origin 0, u = +x, v = +y, horizontal || +x, vertical || +y).  Every product
//       with a zero component is +-0 and x + (+-0) == x, so those terms are dropped;
//       direction.z == lower_left.z and origin.z == +0, hence the hit parameter
//       t = z_pos / lower_left.z is one value per environment.
// POW2: h and w are powers of two (pixel_coord_pow2).
// ---------------------------------------------------------------------------
struct PixelEnv {
    CamDyn dyn;
    Rect rect;
    // AXIS only:
    float tt;   // rectangle.py:128
    bool tmiss; // rectangle.py:130
    float den;  // rectangle.py:168  x_max - x_min = half - (-half)
};

RF_HD PixelEnv make_pixel_env(const float *cd, const float *rc)
{
    PixelEnv e;
    e.dyn = CamDyn{cd[0], cd[1], cd[2], cd[3], cd[4], cd[5], cd[6], cd[7], cd[8]};
    e.rect = Rect{rc[0], rc[1]};
    e.tt = e.rect.z / e.dyn.llz;
    e.tmiss = (e.tt < 0.001f || e.tt > 1000000.0f);
    e.den = e.rect.half - (-e.rect.half);
    return e;
}

RF_HD Colour sample_axis(Rng &g, const PixelEnv &e, double lens_radius, float s, float t,
                         const CheckerTable &tab)
{
    float p0, p1;
    disc_sample(g, p0, p1);
    float ox = (float)((double)p0 * lens_radius);
    float oy = (float)((double)p1 * lens_radius);
    float dx = (e.dyn.llx + e.dyn.hx * s) - ox;
    float dy = (e.dyn.lly + e.dyn.vy * t) - oy;
    float dz = e.dyn.llz;
    float px = ox + dx * e.tt;
    float py = oy + dy * e.tt;
    const float half = e.rect.half;
    bool hit = !e.tmiss && !(px < -half || px > half || py < -half || py > half);
    bool red = false;
    if (hit) {
        float u = (px + half) / e.den;
        float v = (py + half) / e.den;
        float q0, q1, q2;
        sphere_sample(g, q0, q1, q2);
        dx = q0;
        dy = q1;
        dz = 1.0f + q2;
        red = (checker_sign(u, tab) * checker_sign(v, tab)) > 0;
    }
    double T = sky_t(dx, dy, dz);
    float white = (float)(1.0 - T);
    Colour c;
    if (hit) {
        // attenuation (1,0,0) or (0,1,0): the other channels contribute +0
        float ch = sky_channel(T, white, red ? 0.5f : 0.7f);
        c.r = red ? ch : 0.0f;
        c.g = red ? 0.0f : ch;
        c.b = 0.0f;
    } else {
        c.r = sky_channel(T, white, 0.5f);
        c.g = sky_channel(T, white, 0.7f);
        c.b = sky_channel(T, white, 1.0f);
    }
    return c;
}

template <bool AXIS, bool POW2>
RF_HD void render_pixel(Rng &g, int x, int y, int h, int w, int spp, float inv_w, float inv_h,
                        const PixelEnv &e, const CamStatic &cs, const CheckerTable &tab, float &cr,
                        float &cg, float &cb)
{
    cr = cg = cb = 0.0f;
    for (int k = 0; k < spp; ++k) {
        float xi = rng_uniform(g);
        float s = POW2 ? pixel_coord_pow2(x, xi, inv_w) : pixel_coord_literal(x, xi, w);
        float yi = rng_uniform(g);
        float t = POW2 ? pixel_coord_pow2(y, yi, inv_h) : pixel_coord_literal(y, yi, h);
        Colour c = AXIS ? sample_axis(g, e, cs.lens_radius, s, t, tab)
                        : sample_general(g, e.dyn, cs, e.rect, s, t, tab);
        cr = add2(cr, c.r);
        cg = add2(cg, c.g);
        cb = add2(cb, c.b);
    }
}

} // namespace rf
