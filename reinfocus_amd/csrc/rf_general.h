// rf_general.h -- the general renderer: any number of
// spheres and z-aligned rectangles per environment, per-environment cameras, up to 50
// bounces.  Literal arithmetic of
//   render.device_render      graphics/render.py:31-85
//   camera.from_cameras       graphics/camera.py:255-281 (float64[19] rows)
//   world.hit                 graphics/world.py:126-167
//   sphere.hit / uv           graphics/sphere.py:40-117
//   rectangle.hit / uv        graphics/rectangle.py:49-99, :151-170
//   physics.find_colour       graphics/physics.py:95-145
// with the same rounding points as the oracle (numba typing: math.* on f32 -> f64).
// The float64 library calls: sqrt and '/' are correctly rounded on the device (identical to the
// host on 4M operands each); atan2 / acos / sin come from the device math library and differ from
// glibc in the last 1-2 bits for 25 % / 6 % / 1.5 % of operands -- but their results are only used
// through a float32 cast (sphere.uv) or a sign (checker), and a float64 last-bit difference changes
// the float32 only within ~2^-28 of a rounding boundary: 0 of 4M surface normals give a different
// float32 (u, v), 0 of 4M (frequency, u) pairs a different checker sign (tools/diag_general.py,
// profiles/r02_diag_general.txt).  And u, v are used only for the checker sign, so even a float32
// ulp would matter only where frequency * u crosses an integer.  Frames and final RNG states equal
// the oracle's on every scene tested (tests allow no differing pixel).
#pragma once

#include <math.h>
#include <stdint.h>

#include "rf_math.h"

namespace rf {

constexpr double kPi = 3.14159265358979323846;

struct HitRec {
    float p[3], n[3];
    float t;
    bool red; // physics.py:47-64 colour_checkerboard of the hit: red (true) or green
};

RF_HD float dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

RF_HD bool sphere_hit(const float *sp, const float o[3], const float d[3], float t_min, float t_max, HitRec &r)
{
    const float centre[3] = {sp[0], sp[1], sp[2]};
    const float radius = sp[3];
    const float oc[3] = {o[0] - centre[0], o[1] - centre[1], o[2] - centre[2]};
    const float a = dot3(d, d);
    const float b = dot3(oc, d);
    const float c = dot3(oc, oc) - radius * radius;
    const float disc = b * b - a * c;
    if (disc < 0)
        return false;
    // A certain miss without float64.  With a > 0 and b > 0 (the ray points away from the centre: every ray that has just
    // scattered off this sphere) the first root -(b + sqrtd) / a is negative, and the second, (sqrtd - b) / a, is below
    // t_min exactly when disc < (b + t_min a)^2.  The float32 evaluation of the right side is off by < 2^-22 relative
    // (two additions, two products), the reference's float64 root by < 2^-51 (b / a + t_min): with the factor 1 - 2^-20
    // the comparison below implies root < t_min in the reference's own arithmetic, i.e. `return false` two branches on.
    // (Anything else -- grazing rays, NaNs, a == 0 -- takes the literal path.)  Half of all sphere tests of a scene are
    // such rays; their float64 sqrt and two divisions were a fifth of a sphere scene's instructions.
    if (a > 0.0f && b > 0.0f) {
        const float reach = b + t_min * a;
        if (disc < (reach * reach) * 0.99999904632568359375f /* 1 - 2^-20 */)
            return false;
    }
    const double sqrtd = sqrt((double)disc);
    double root = (-(double)b - sqrtd) / (double)a;
    if (root < (double)t_min || (double)t_max < root) {
        root = (-(double)b + sqrtd) / (double)a;
        if (root < (double)t_min || (double)t_max < root)
            return false;
    }
    const float inv_r = (float)(1.0 / (double)radius);
    for (int k = 0; k < 3; ++k) {
        r.p[k] = add2(o[k], (float)((double)d[k] * root));
        r.n[k] = (r.p[k] - centre[k]) * inv_r;
    }
    r.t = (float)root;
    return true;
}

// The float64 library calls (atan2, acos, sin) expand to long instruction sequences with many
// live registers; inlined into the bounce loop they push the kernel to ~200 VGPRs (2 waves per
// SIMD).  Kept out of line, with arguments and results in registers, the kernel needs ~110.
#if defined(__HIP_DEVICE_COMPILE__)
#define RF_COLD __device__ __attribute__((noinline))
#else
#define RF_COLD inline
#endif
struct TexCoord {
    float u, v;
};

// sphere.py:106-117 uv from the surface normal
RF_COLD TexCoord sphere_uv(float n0, float n1, float n2)
{
    TexCoord t;
    t.u = (float)((atan2(-(double)n2, (double)n0) + kPi) / kPi);
    t.v = (float)(acos(-(double)n1) / kPi);
    return t;
}

// sign of sin((f * pi) * u) as the reference evaluates it in float64: -1, 0, +1 (NaN -> 0)
RF_COLD int checker_sign_f64(float f, float u)
{
    const double m = (double)f * (double)u; // exact: two f32 factors
    if (!(m == m) || m - m != 0.0)          // NaN or infinite argument: sin is NaN
        return 0;
    const double nearest = rint(m);
    const double tol = 1e-9 * (fabs(m) > 1.0 ? fabs(m) : 1.0);
    if (fabs(m - nearest) > tol) {
        // fl64(fl64(f*pi)*u) differs from pi*m by < 4e-16 relative: same side of every zero
        const double k = floor(m);
        return (k - 2.0 * floor(k * 0.5)) != 0.0 ? -1 : 1;
    }
    const double s = sin(((double)f * kPi) * (double)u);
    return s > 0.0 ? 1 : (s < 0.0 ? -1 : 0);
}

// The same sign from float32 arithmetic whenever that is safe: m32 = RN32(f * u) is within
// 2^-24 |m| of the exact product m, so if m32 lies further than 4 * 2^-22 * max(|m32|, 1) from both
// neighbouring integers, floor(m) == floor(m32) and m is nowhere near the band in which
// checker_sign_f64 consults sin -- the sign is (-1)^floor(m32).  Everything else (about one
// coordinate in a million, NaN, huge arguments) takes the float64 function.
RF_HD int checker_sign_general(float f, float u)
{
    const float m = f * u;
    const float fl = __builtin_floorf(m);
    const float fr = m - fl; // exact for |m| < 2^23
    const float am = __builtin_fabsf(m);
    const float margin = (am > 1.0f ? am : 1.0f) * 9.5367431640625e-07f; // 2^-20
    if (am < 65536.0f && fr > margin && fr < 1.0f - margin) // false for NaN
        return ((int)fl & 1) ? -1 : 1;
    return checker_sign_f64(f, u);
}

// --- checker colour of a sphere hit without float64 in the common case ----------------------
// The texture coordinates of a sphere, u = (atan2(-n.z, n.x) + pi) / pi and v = acos(-n.y) / pi
// (sphere.py:106-117), are only ever used for the sign of sin(f * pi * u) * sin(f * pi * v), i.e.
// for the parities of floor(fu * u) and floor(fv * v) -- unless a product is so close to an
// integer that rounding decides.  So: float32 approximations of u and v (|error| < 1e-6,
// bound below), and the parities are taken from them whenever both products are further from
// every integer than 3x that bound allows them to be wrong; otherwise (a few coordinates in
// 10^4), and for anything not finite, the float64 expressions of the reference decide.  Both
// paths give the same colour wherever the fast one is taken -- checked on 10^8 normals including
// ones placed on and next to the checker's edges (tests/test_general_renderer.py).

// atan(z) / z on [0, 1] as a polynomial in z^2 (near-minimax fit, |atan error| < 3.7e-7 in float32)
RF_HD float atan_unit_approx(float z)
{
    const float t = z * z;
    float p = 0.0068117305636405945f;
    p = __builtin_fmaf(p, t, -0.033604010939598083f);
    p = __builtin_fmaf(p, t, 0.07962340861558914f);
    p = __builtin_fmaf(p, t, -0.13233324885368347f);
    p = __builtin_fmaf(p, t, 0.19807811081409454f);
    p = __builtin_fmaf(p, t, -0.3331736624240875f);
    p = __builtin_fmaf(p, t, 0.9999961256980896f);
    return p * z;
}

RF_HD float rcp_approx(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x); // 1 ulp
#else
    return 1.0f / x;
#endif
}

RF_HD float sqrt_approx(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x); // 1 ulp
#else
    return __builtin_sqrtf(x);
#endif
}

// atan2(y, x) in [-pi, pi], |error| < 1e-6 (3.7e-7 polynomial + roundings); NaN for x = y = 0
RF_HD float atan2_approx(float y, float x)
{
    const float ay = __builtin_fabsf(y), ax = __builtin_fabsf(x);
    const float hi = ax > ay ? ax : ay, lo = ax > ay ? ay : ax;
    float r = atan_unit_approx(lo * rcp_approx(hi));
    r = ay > ax ? 1.5707963267948966f - r : r;
    r = x < 0.0f ? 3.14159265358979323846f - r : r;
    return y < 0.0f ? -r : r;
}

// parity of floor(m) when m (known to within `slack`) is further than that from every integer
RF_HD bool safe_parity(float m, float slack, int &odd)
{
    const float fl = __builtin_floorf(m);
    const float fr = m - fl;
    odd = (int)fl & 1;
    return __builtin_fabsf(m) < 65536.0f && fr > slack && fr < 1.0f - slack; // false for NaN
}

RF_HD bool sphere_red_exact(const float n[3], float fu, float fv)
{
    const TexCoord t = sphere_uv(n[0], n[1], n[2]);
    return checker_sign_general(fu, t.u) * checker_sign_general(fv, t.v) > 0;
}

// float32 approximations of sphere.uv, |error| < 1e-6 each (measured: < 4e-7)
RF_HD void sphere_uv_approx(const float n[3], float &u, float &v)
{
    constexpr float kInvPi = 0.3183098861837907f;
    u = (atan2_approx(-n[2], n[0]) + 3.14159265358979323846f) * kInvPi;
    const float s = sqrt_approx((1.0f - n[1]) * (1.0f + n[1])); // NaN when |n.y| > 1
    v = atan2_approx(s, -n[1]) * kInvPi;                          // acos(-n.y) / pi
}

RF_HD bool sphere_red(const float n[3], float fu, float fv)
{
    float u, v;
    sphere_uv_approx(n, u, v);
    const float mu = fu * u, mv = fv * v;
    int odd_u, odd_v;
    // the products are off by at most |f| * 1e-6 + |m| * 2e-7 (approximation, the reference's own
    // float32 rounding of u and v, the product's rounding): twice that and more as the margin
    const bool quick = safe_parity(mu, (__builtin_fabsf(fu) + __builtin_fabsf(mu) + 1.0f) * 2e-6f, odd_u) &&
                       safe_parity(mv, (__builtin_fabsf(fv) + __builtin_fabsf(mv) + 1.0f) * 2e-6f, odd_v);
    if (__builtin_expect(quick, 1))
        return odd_u == odd_v; // sign(sin) = (-1)^floor: the product is positive when the parities agree
    return sphere_red_exact(n, fu, fv);
}

RF_HD bool rectangle_hit(const float *rp, const float o[3], const float d[3], float t_min, float t_max, HitRec &r)
{
    const float t = (rp[4] - o[2]) / d[2];
    if (t < t_min || t > t_max)
        return false;
    float p[3];
    for (int k = 0; k < 3; ++k)
        p[k] = add2(o[k], d[k] * t);
    if (p[0] < rp[0] || p[0] > rp[1] || p[1] < rp[2] || p[1] > rp[3])
        return false;
    for (int k = 0; k < 3; ++k)
        r.p[k] = p[k];
    r.n[0] = 0.0f;
    r.n[1] = 0.0f;
    r.n[2] = 1.0f;
    r.t = t;
    return true;
}

// rectangle.py:151-170 uv and the checker colour of it
RF_HD bool rectangle_red(const float *rp, const HitRec &r)
{
    const float u = (r.p[0] - rp[0]) / (rp[1] - rp[0]);
    const float v = (r.p[1] - rp[2]) / (rp[3] - rp[2]);
    return checker_sign_general(rp[5], u) * checker_sign_general(rp[6], v) > 0;
}

RF_HD bool world_hit(const float *params, const int32_t *types, int n_shapes, int width, const float o[3],
                     const float d[3], float t_min, float t_max, HitRec &rec)
{
    int which = -1;
    float closest = t_max;
    for (int i = 0; i < n_shapes; ++i) {
        HitRec tmp;
        const float *p = params + (long)i * width;
        const bool h = types[i] == 0 ? sphere_hit(p, o, d, t_min, closest, tmp)
                                     : rectangle_hit(p, o, d, t_min, closest, tmp);
        if (h) {
            which = i;
            closest = tmp.t;
            rec = tmp;
        }
    }
    if (which < 0)
        return false;
    // texture coordinates / checker colour of the closest hit only (the reference computes uv for
    // every candidate inside hit(); only the closest one's is ever read)
    const float *shape = params + (long)which * width;
    rec.red = types[which] == 0 ? sphere_red(rec.n, shape[4], shape[5]) : rectangle_red(shape, rec);
    return true;
}

// physics.py:67-92 scatter, without the random_in_unit_sphere draw: the new ray and the
// attenuation of a hit; q is the accepted sphere sample
RF_HD void scatter_step(const HitRec &rec, float q0, float q1, float q2, float o[3], float d[3], float &ar, float &ag,
                        float &ab)
{
    o[0] = rec.p[0];
    o[1] = rec.p[1];
    o[2] = rec.p[2];
    d[0] = add2(rec.n[0], q0);
    d[1] = add2(rec.n[1], q1);
    d[2] = add2(rec.n[2], q2);
    ar = ar * (rec.red ? 1.0f : 0.0f);
    ag = ag * (rec.red ? 0.0f : 1.0f);
    ab = ab * 0.0f;
}

// physics.py:137-145: the sky seen along d times the attenuation gathered so far, in the float32
// form proven equal to the float64 chain (rf_math.h)
RF_HD Colour sky_colour(const float d[3], float ar, float ag, float ab)
{
    const float ud1 = unit_dir_y(d[0], d[1], d[2]);
    const float white = sky_white(ud1);
    Colour c;
    c.r = add2(white, sky_blue(ud1, kSkyHalf[0])) * ar;
    c.g = add2(white, sky_blue(ud1, kSkyHalf[1])) * ag;
    c.b = add2(white, sky_blue(ud1, kSkyHalf[2])) * ab;
    return c;
}

constexpr int kMaxBounces = 50; // physics.py:118

RF_HD Colour find_colour(const float *params, const int32_t *types, int n_shapes, int width, const float o_in[3],
                         const float d_in[3], Rng &g)
{
    float o[3] = {o_in[0], o_in[1], o_in[2]}, d[3] = {d_in[0], d_in[1], d_in[2]};
    float ar = 1.0f, ag = 1.0f, ab = 1.0f;
    for (int bounce = 0; bounce < kMaxBounces; ++bounce) {
        HitRec rec;
        if (world_hit(params, types, n_shapes, width, o, d, 0.001f, 1000000.0f, rec)) {
            float q0, q1, q2;
            sphere_sample(g, q0, q1, q2);
            scatter_step(rec, q0, q1, q2, o, d, ar, ag, ab);
        } else {
            return sky_colour(d, ar, ag, ab);
        }
    }
    return Colour{0.0f, 0.0f, 0.0f};
}

// camera.from_cameras (camera.py:255-281): the float64[19] row of camera.Cameras cast to the
// float32 tuples the kernel works with (lens radius stays float64).  The casts are per
// environment, not per pixel: the GPU path does them once on the host (rf_abi_general.hip) so that the
// kernel reads block-uniform floats into scalar registers.
struct GeneralCamera {
    float f[18]; // lower_left, horizontal, vertical, origin, u, v
    double lens_radius;
    // per-environment constants of general_ray for the cooperative single-rectangle kernel (rf_general_one.h), which
    // reads them into scalar registers instead of keeping 18 vector registers of loop invariants: float64(u),
    // float64(v) (vector.py:190 promotes the float32 components when it scales them by the float64 lens offsets) and
    // the leading `0 + a` of the two three-term sums
    double u64[3], v64[3];
    float origin0[3], lower_left0[3];
    // the lens radius split for the float32 form of the lens offset (rf_math.h lens_offset<1>: rf_general_dense.h uses
    // it where the host has proven it exact for this radius)
    float lens_hi, lens_lo;
};

RF_HD GeneralCamera general_camera(const double *cam /*[19]*/)
{
    GeneralCamera c;
    for (int k = 0; k < 18; ++k)
        c.f[k] = (float)cam[k];
    c.lens_radius = cam[18];
    for (int k = 0; k < 3; ++k) {
        c.u64[k] = (double)c.f[12 + k];
        c.v64[k] = (double)c.f[15 + k];
        c.origin0[k] = 0.0f + c.f[9 + k];
        c.lower_left0[k] = 0.0f + c.f[k];
    }
    c.lens_hi = (float)c.lens_radius;
    c.lens_lo = (float)(c.lens_radius - (double)c.lens_hi);
    return c;
}

// RN32(RN64(x + xi) / w) by the 3-operation quotient of rf_math.h for the frame sizes it is
// proven for (every w <= 4096, see pixel_coord_div), by the IEEE division otherwise
struct GeneralFrame {
    int h, w;
    bool quick;
    // 1 / w, 1 / h for power-of-two frames: RN32(RN64(x + xi) / w) == RN32(x + xi) * 2^-k (rf_math.h
    // pixel_coord_pow2), no float64 (the POW2 instance)
    float inv_w, inv_h;
    double w64, h64, rw64, rh64;
};

RF_HD GeneralFrame general_frame(int h, int w)
{
    GeneralFrame f;
    f.h = h;
    f.w = w;
    f.inv_w = 1.0f / (float)w; // exact for the powers of two it is used for
    f.inv_h = 1.0f / (float)h;
    f.quick = w <= 4096 && h <= 4096;
    f.w64 = (double)w;
    f.h64 = (double)h;
    f.rw64 = 1.0 / f.w64;
    f.rh64 = 1.0 / f.h64;
    return f;
}

// render.py:61-66: the jittered coordinates of one sample, two draws (x first)
// POW2: both frame sizes are powers of two (a separate kernel instance: the float64 coordinate code
// is not even compiled into it)
template <bool POW2>
RF_HD void general_coords(Rng &g, int x, int y, const GeneralFrame &f, float &s, float &t)
{
    const float xi = rng_uniform(g);
    s = POW2 ? pixel_coord_pow2(x, xi, f.inv_w)
             : (f.quick ? pixel_coord_div(x, xi, f.w64, f.rw64) : pixel_coord_literal(x, xi, f.w));
    const float yi = rng_uniform(g);
    t = POW2 ? pixel_coord_pow2(y, yi, f.inv_h)
             : (f.quick ? pixel_coord_div(y, yi, f.h64, f.rh64) : pixel_coord_literal(y, yi, f.h));
}

// camera.get_ray (camera.py:307-350) for a general camera frame; p is the lens-disc sample
RF_HD void general_ray(const CamDyn &dyn, const CamStatic &cs, float p0, float p1, float s, float t, float o[3],
                       float d[3])
{
    const double rd0 = (double)p0 * cs.lens_radius, rd1 = (double)p1 * cs.lens_radius;
    o[0] = add3(cs.ox, (float)((double)cs.ux * rd0), (float)((double)cs.vx * rd1));
    o[1] = add3(cs.oy, (float)((double)cs.uy * rd0), (float)((double)cs.vy * rd1));
    o[2] = add3(cs.oz, (float)((double)cs.uz * rd0), (float)((double)cs.vz * rd1));
    d[0] = add3(dyn.llx, dyn.hx * s, dyn.vx * t) - o[0];
    d[1] = add3(dyn.lly, dyn.hy * s, dyn.vy * t) - o[1];
    d[2] = add3(dyn.llz, dyn.hz * s, dyn.vz * t) - o[2];
}

// one pixel of device_render
template <bool POW2>
RF_HD void render_pixel_general(Rng &g, int x, int y, int h, int w, int spp, const GeneralCamera &cam,
                                const float *params, const int32_t *types, int n_shapes, int width, float &cr,
                                float &cg, float &cb)
{
    const CamDyn dyn{cam.f[0], cam.f[1], cam.f[2], cam.f[3], cam.f[4], cam.f[5], cam.f[6], cam.f[7], cam.f[8]};
    const CamStatic cs{cam.f[9],  cam.f[10], cam.f[11], cam.f[12], cam.f[13],
                       cam.f[14], cam.f[15], cam.f[16], cam.f[17], cam.lens_radius, 0.0f, 0.0f, 0};
    cr = cg = cb = 0.0f;
    const GeneralFrame frame = general_frame(h, w);
    for (int k = 0; k < spp; ++k) {
        float s, t;
        general_coords<POW2>(g, x, y, frame, s, t);
        float p0, p1;
        disc_sample(g, p0, p1);
        float o[3], d[3];
        general_ray(dyn, cs, p0, p1, s, t, o, d);
        const Colour c = find_colour(params, types, n_shapes, width, o, d, g);
        cr = add2(cr, c.r);
        cg = add2(cg, c.g);
        cb = add2(cb, c.b);
    }
}

} // namespace rf
