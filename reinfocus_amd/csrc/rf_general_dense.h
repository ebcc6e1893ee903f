// rf_general_dense.h -- the general renderer for worlds of a few shapes per environment, without
// float64 sequences: render_general_dense_kernel<POW2, NS, TILED, SIMPLE> (the kernel itself: rf_general_one.h, next to the
// fix-up kernel it shares with the one-shape kernel).
//
// Why.  The literal kernel (rf_general_kernels.h) is held to 5 waves per SIMD by what its float64 pieces need while they
// run -- the sphere's roots (sqrt and two divisions), the camera's lens products, the out-of-line atan2 / acos / sin of the
// texture code -- although together they are 4 % of its instructions.  Taking any one of them away buys 1 ... 10 %,
// taking all of them away 32 ... 36 % (timing builds, profiles/r05_ab.txt section 1): 72 registers instead of 95, no
// call sequences, 7 waves.  So this kernel evaluates everything in float32 -- the roots in double-float (pairs of
// float32) -- and a pixel for which that cannot be PROVEN to give the reference's bits abstains: it stores neither its RNG
// state nor its bytes, its index goes to a list, and render_general_fixup_kernel (rf_general_one.h) renders it with the
// literal code from the untouched state.  Abstaining is always safe; a sample that does not abstain has taken, at every
// comparison and at every rounding to float32, the branch / the value the reference's float64 expression takes:
//
//   * camera.get_ray (camera.py:307-350).  SIMPLE instances: every camera's u and v have components in {0, +1, -1} and the
//     float32 form of the lens offset is exact for its radius (rf_abi_ctx.hip lens_split tries every disc coordinate):
//     float32(float64(u_k) * (float64(p) * radius)) is then +-lens_offset(p) or +-0 -- exactly what u_k * lens_offset(p) is
//     in float32.  Other cameras (tilted axes, other radii) take the instances with the reference's float64 lens products,
//     sixteen inline instructions with nothing to decide.
//   * sphere.hit (sphere.py:40-103).  a, b, c and the discriminant are the reference's float32 expressions.  A negative
//     discriminant and the certain miss of rf_general.h are exact decisions.  Otherwise root = (-b -+ sqrt(disc)) / a is
//     evaluated in double-float with a relative error below 2^-43 (kappa + 1), kappa the cancellation of its numerator
//     (bound and measurement: sphere_root_df below); the comparisons with t_min / t_max abstain within 2^-20 relative of
//     the bound, and each of the four roundings to float32 -- t and the three products d_k * root -- abstains unless the
//     double-float value is further than 2^-41 (kappa + 1) relative from every rounding boundary (4x the error bound; the
//     reference's own float64 roundings move its value by < 2^-50 (kappa + 1)).
//   * rectangle.hit is float32 in the reference (IEEE division: kept); texture coordinates by the correctly rounded
//     three-operation quotient (rf_math.h div_by_const) where the extent allows.
//   * checker colours: float32 decisions (checker_sign_dense, sphere_red_dense: rf_general.h's own float32 tests and
//     margins) which abstain where the literal code would consult float64.
//   * the sky is float32 in every kernel (rf_math.h: proven equal to the float64 chain).
//
// NS: the most shapes an environment of the launch holds (1 ... 3; the reference's shape factories build one or two); an
// environment with fewer leaves the loop early (its count is block-uniform).  The shapes' rows and what the kernel
// derives from them per shape (radius^2, float32(1 / radius), texture extents and their reciprocals) come from the host as
// one 64-byte record per shape, read through the constant address space into scalar registers: the bounce loop has no
// memory access.  Worlds of more than three shapes take the literal kernel.
#pragma once

#include "rf_general.h"

namespace rf {

// RF_TEST_DOUBT (tests/gpucheck/libreinfocus_doubt.so): abstentions are rare (about one pixel in 10^4) and a decision
// inside a margin is still right almost always, so a pixel that abstains without being listed would go unnoticed.  The
// test build makes the margins wide (a fifth of the decisions abstain) and the answer of a decision that abstains WRONG.
#ifndef RF_TEST_DOUBT
#define RF_TEST_DOUBT 0
#endif
#if defined(__HIPCC__)
#define RF_UNROLL _Pragma("unroll")
#else
#define RF_UNROLL
#endif

struct ShapeConst {  // 64 bytes per shape, built by shape_const() on the host
    float p[7];      // the reference's row: sphere  cx cy cz radius fu fv -; rectangle  x0 x1 y0 y1 z fu fv
    int32_t type;    // 0 sphere, 1 rectangle (shape.py)
    float k[8];      // sphere: radius^2, float32(1 / radius); rectangle: den_u, den_v, 1 / den_u, 1 / den_v, quick (0 / 1)
};
static_assert(sizeof(ShapeConst) == 64, "one s_load_dwordx16");

RF_HD ShapeConst shape_const(const float *row, int width, int type)
{
    ShapeConst s;
    for (int i = 0; i < 7; ++i)
        s.p[i] = i < width ? row[i] : 0.0f;
    s.type = type;
    for (int i = 0; i < 8; ++i)
        s.k[i] = 0.0f;
    if (type == 0) {
        s.k[0] = row[3] * row[3];                  // sphere.py:72: float32 product
        s.k[1] = (float)(1.0 / (double)row[3]);    // sphere.py:92: numpy.float32(1.0 / radius)
    } else {
        const float den_u = row[1] - row[0], den_v = row[3] - row[2]; // rectangle.py:168-169
        s.k[0] = den_u;
        s.k[1] = den_v;
        s.k[2] = 1.0f / den_u;
        s.k[3] = 1.0f / den_v;
        // div_by_const is proven for divisors in [2^-40, 2^40] (rf_math.h)
        s.k[4] = (den_u >= 9.094947017729282e-13f && den_u <= 1099511627776.0f && den_v >= 9.094947017729282e-13f &&
                  den_v <= 1099511627776.0f) ? 1.0f : 0.0f;
    }
    return s;
}

// A camera the dense kernel can take: u and v made of 0 / +-1 (rf_general.h GeneralCamera::f[12..17])
RF_HD bool camera_axes_simple(const GeneralCamera &c)
{
    for (int k = 12; k < 18; ++k)
        if (!(c.f[k] == 0.0f || c.f[k] == 1.0f || c.f[k] == -1.0f))
            return false;
    return true;
}

// --- double-float pieces ------------------------------------------------------------------------------------------
// (1 ulp approximations on the device, correctly rounded on the host, where tests/hostsim perturbs them by +-1 ulp to
// show that nothing below depends on more than the stated accuracy)
#if defined(RF_HOSTSIM)
extern thread_local unsigned g_dense_perturb; // tests/hostsim: 0, or the state of a generator that nudges the approximations
inline float nudge(float x)
{
    if (g_dense_perturb == 0)
        return x;
    g_dense_perturb = g_dense_perturb * 1103515245u + 12345u;
    const int step = (int)((g_dense_perturb >> 16) % 3u) - 1; // -1, 0, +1 ulp
    uint32_t b;
    __builtin_memcpy(&b, &x, 4);
    b += (uint32_t)step;
    __builtin_memcpy(&x, &b, 4);
    return x;
}
#define RF_NUDGE(x) nudge(x)
#else
#define RF_NUDGE(x) (x)
#endif
RF_HD float sqrt_1ulp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return RF_NUDGE(__builtin_sqrtf(x));
#endif
}
RF_HD float rcp_1ulp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return RF_NUDGE(1.0f / x);
#endif
}

// 2^lo <= |x| <= 2^hi, as one unsigned compare on the exponent field (false for 0, subnormals, infinities, NaN)
RF_HD bool magnitude_within(float x, int lo, int hi)
{
    uint32_t b;
    __builtin_memcpy(&b, &x, 4);
    return ((b & 0x7FFFFFFFu) - ((uint32_t)(lo + 127) << 23)) <= (((uint32_t)(hi - lo)) << 23);
}

constexpr float kDfBoundary = RF_TEST_DOUBT ? 3.725290298461914e-09f /* 2^-28 */ : 4.547473508864641e-13f /* 2^-41 */;
constexpr float kDfCompare = RF_TEST_DOUBT ? 0.125f : 9.5367431640625e-07f /* 2^-20 */;
constexpr float kDfMaxCancel = 65.0f; // numerators that cancel to less than 1 / 64 of their larger operand abstain

// (hi, lo) is a double-float value V = hi + lo (hi = RN32(hi + lo)) whose distance from the real number the reference
// rounds is below tol / 4 * |V|.  True when RN32 of that real number is certainly hi: moving |hi| TOWARDS zero by
// m = |lo| + tol |hi| -- the side on which the floats are at least as dense -- still rounds to |hi|, so every real
// within m of hi on either side rounds to hi.  (Ties: hi - m == hi under ties-to-even also when m is exactly half a
// step and hi is even -- then |lo| is below half a step by tol |hi|, 4x the error bound.)
RF_HD bool rounds_to_hi(float hi, float lo, float tol)
{
    const float ah = __builtin_fabsf(hi);
    const float m = __builtin_fmaf(ah, tol, __builtin_fabsf(lo));
    return (ah - m) == ah;
}

// root = (-b - sign * sqrt(disc)) / a as the reference evaluates it in float64 (sphere.py:79-84), in double-float:
// (qh, ql) with qh = RN32(qh + ql), and tol = 4x a bound on its relative error.  Requires 2^-40 <= a <= 2^40,
// 2^-60 <= disc <= 2^60, |b| in [2^-60, 2^40] or b == 0; returns false (abstain) when the numerator cancels to less than
// 1 / 64 of its larger operand.
// Error.  s = sqrt_1ulp(disc) is within 2^-23 relative (with the product's rounding: 2^-22.4) of S = sqrt(disc); the
// residual r = disc - s^2 comes out of one fma with a relative error of 2^-24; S = s + r / (2 s) - r^2 / (8 s^3) ..., the
// quadratic term is below 2^-46 S, and sl = r * (0.5 * rcp(s)) carries the reciprocal's 2^-22.4 on a term of 2^-22.4 S:
// s + sl is within 2^-44.5 of S.  The numerator's two_sum is exact and its low word absorbs sl with a rounding of 2^-24 on
// a term of 2^-22 max(|b|, S): relative to the numerator N the error is 2^-44.4 kappa, kappa = max(|b|, S) / |N| the
// cancellation.  The quotient: q = nh * ra is within 2^-22.4 of nh / a, the remainder nh - q a comes out of one fma
// (2^-24 relative to itself), and the correction (rem + nl) * ra carries 2^-22.2 on a term of 2^-22.4 Q: 2^-44.6 Q.
// Together below 2^-44.4 (kappa + 1) for approximations of 1 ulp; the bound used is 2^-43 (kappa + 1) -- which also covers
// approximations off by 1.5 ulp and the reference's own float64 roundings (< 2^-50 (kappa + 1)) -- and tol = 2^-41
// (kappa + 1) is 4x that.  tests/test_general_renderer.py measures the error against long double over 10^7 rays of every
// kind with the approximations nudged by +-1 ulp around the correctly rounded values: at most 0.6 of the bound.
RF_HD bool sphere_root_df(float a, float b, float disc, float sign, float &qh, float &ql, float &tol)
{
    const float s = sqrt_1ulp(disc);
    const float r = __builtin_fmaf(-s, s, disc);
    const float sl = r * (0.5f * rcp_1ulp(s));
    // N = -b - sign * (s + sl): two_sum of x = -b and y = -sign * s
    const float x = -b, y = -sign * s;
    const float nh0 = x + y;
    const float bv = nh0 - x;
    const float av = nh0 - bv;
    const float ne = (x - av) + (y - bv);
    const float nl = ne - sign * sl;
    const float big = __builtin_fmaxf(__builtin_fabsf(b), s);
    const float kap1 = __builtin_fmaf(big, rcp_1ulp(__builtin_fabsf(nh0)), 1.0f); // kappa + 1 (NaN / inf for nh0 == 0)
    if (!(kap1 <= kDfMaxCancel))
        return false;
    tol = kap1 * kDfBoundary;
    const float ra = rcp_1ulp(a);
    const float q = nh0 * ra;
    const float rem = __builtin_fmaf(-q, a, nh0);
    const float qc = (rem + nl) * ra;
    qh = q + qc;
    ql = qc - (qh - q);
    return true;
}

constexpr int kMiss = 0, kHit = 1, kDoubt = 2;

// sphere.hit (sphere.py:40-103) against (centre, radius) with a = dot(d, d) given; r2 = float32(radius * radius),
// inv_r = float32(1 / radius) from ShapeConst.  kMiss / kHit are the reference's answers (and, for a hit, its record);
// kDoubt: float32 cannot tell.
RF_HD int sphere_hit_dense(const float centre[3], float r2, float inv_r, const float o[3], const float d[3], float a,
                           float t_min, float t_max, HitRec &rec)
{
    const float oc[3] = {o[0] - centre[0], o[1] - centre[1], o[2] - centre[2]};
    const float b = dot3(oc, d);
    const float c = dot3(oc, oc) - r2;
    const float disc = b * b - a * c;
    if (disc < 0)
        return kMiss;
    if (a > 0.0f && b > 0.0f) { // rf_general.h sphere_hit: the certain miss of a ray that points away from the centre
        const float reach = b + t_min * a;
        if (disc < (reach * reach) * 0.99999904632568359375f /* 1 - 2^-20 */)
            return kMiss;
    }
    // what the double-float evaluation is proven for (anything else -- NaNs, a == 0, a zero discriminant, huge or tiny
    // operands -- abstains)
    if (!(magnitude_within(a, -40, 40) && magnitude_within(disc, -60, 60) && (b == 0.0f || magnitude_within(b, -60, 40)) &&
          t_min > 0.0f && t_max > t_min))
        return kDoubt;
    float qh, ql, tol;
    if (!sphere_root_df(a, b, disc, 1.0f, qh, ql, tol))
        return kDoubt;
    // sphere.py:80: root < t_min or t_max < root, certain only outside a band of 2^-20 around either bound
    const float lo_m = t_min * (1.0f - kDfCompare), lo_p = t_min * (1.0f + kDfCompare);
    const float hi_m = t_max * (1.0f - kDfCompare), hi_p = t_max * (1.0f + kDfCompare);
    bool inside = qh > lo_p && qh < hi_m;
    if (!inside) {
        if (!(qh < lo_m || qh > hi_p))
            return kDoubt;
        if (!sphere_root_df(a, b, disc, -1.0f, qh, ql, tol)) // sphere.py:82: the far root
            return kDoubt;
        inside = qh > lo_p && qh < hi_m;
        if (!inside)
            return (qh < lo_m || qh > hi_p) ? kMiss : kDoubt;
    }
    // the four roundings to float32: t = float32(root), p_k = o_k + float32(d_k * root) (ray.py:29-40)
    bool sure = rounds_to_hi(qh, ql, tol) && magnitude_within(qh, -60, 60);
RF_UNROLL
    for (int k = 0; k < 3; ++k) {
        const float ph = d[k] * qh;
        const float pe = __builtin_fmaf(d[k], qh, -ph); // exact: the product's error
        const float pl = __builtin_fmaf(d[k], ql, pe);
        const float hi = ph + pl;
        const float lo = pl - (hi - ph);
        // d_k == 0: the product is +-0 and o_k + (+-0) is 0 + o_k either way; a product too small for the error term to
        // be exact abstains
        sure = sure && (d[k] == 0.0f || (magnitude_within(ph, -60, 100) && rounds_to_hi(hi, lo, tol)));
        rec.p[k] = add2(o[k], hi);
        rec.n[k] = (rec.p[k] - centre[k]) * inv_r;
    }
    rec.t = qh;
#if RF_TEST_DOUBT
    if (!sure) // (the test build: an abstention's record is visibly wrong)
        rec.p[1] = rec.p[1] + 0.25f, rec.n[1] = -rec.n[1];
#endif
    return sure ? kHit : kDoubt;
}

// float32 decisions of the checker colour, or abstention (the same expressions and margins as rf_general.h
// checker_sign_general / sphere_red, which consult float64 where these abstain)
RF_HD int checker_sign_dense(float f, float u, bool &doubt)
{
    const float m = f * u;
    const float fl = __builtin_floorf(m);
    const float fr = m - fl;
    const float am = __builtin_fabsf(m);
    const float margin = (am > 1.0f ? am : 1.0f) * (RF_TEST_DOUBT ? 0.0625f : 9.5367431640625e-07f); // 2^-20
    const bool quick = am < 65536.0f && fr > margin && fr < 1.0f - margin; // false for NaN
    doubt = doubt || !quick;
    const int sign = ((int)fl & 1) ? -1 : 1;
    return (RF_TEST_DOUBT && !quick) ? -sign : sign;
}

RF_HD bool sphere_red_dense(const float n[3], float fu, float fv, bool &doubt)
{
    float u, v;
    sphere_uv_approx(n, u, v);
    const float mu = fu * u, mv = fv * v;
    int odd_u, odd_v;
    const float slack = RF_TEST_DOUBT ? 750.0f : 1.0f; // (the test build: about a fifth of the decisions abstain)
    const bool quick_u = safe_parity(mu, (__builtin_fabsf(fu) + __builtin_fabsf(mu) + 1.0f) * 2e-6f * slack, odd_u);
    const bool quick_v = safe_parity(mv, (__builtin_fabsf(fv) + __builtin_fabsf(mv) + 1.0f) * 2e-6f * slack, odd_v);
    const bool quick = quick_u && quick_v;
    doubt = doubt || !quick;
    return (odd_u == odd_v) != (RF_TEST_DOUBT && !quick); // (the test build: an abstention's answer is wrong)
}

// rectangle.py:151-170 uv + physics.py:47-64 colour_checkerboard of a hit at p on the rectangle s
RF_HD bool rectangle_red_dense(const float *rp, const float *k, float px, float py, bool &doubt)
{
    float u, v;
    if (k[4] != 0.0f) { // per shape: uniform
        u = div_by_const(px - rp[0], k[0], k[2]);
        v = div_by_const(py - rp[2], k[1], k[3]);
    } else {
        u = (px - rp[0]) / k[0];
        v = (py - rp[2]) / k[1];
    }
    return checker_sign_dense(rp[5], u, doubt) * checker_sign_dense(rp[6], v, doubt) > 0;
}

// camera.get_ray (camera.py:307-350) for any camera: the reference's float64 lens products (rf_general.h general_ray with the
// per-environment constants the host prepared: GeneralCamera::u64 ...)
template <class Cam>
RF_HD void general_ray_any(const Cam &cam, float p0, float p1, float s, float t, float o[3], float d[3])
{
    const double rd0 = (double)p0 * cam.lens_radius, rd1 = (double)p1 * cam.lens_radius;
RF_UNROLL
    for (int k = 0; k < 3; ++k) {
        o[k] = (cam.origin0[k] + (float)(cam.u64[k] * rd0)) + (float)(cam.v64[k] * rd1);
        d[k] = ((cam.lower_left0[k] + cam.f[3 + k] * s) + cam.f[6 + k] * t) - o[k];
    }
}

// camera.get_ray (camera.py:307-350) for a camera with simple axes and a lens radius whose float32 form is exact
// (lens = {hi, lo} of rf_math.h lens_offset<1>); cf = GeneralCamera::f, origin0 / lower_left0 its leading sums
template <class Cam>
RF_HD void general_ray_simple(const Cam &cam, float lens_hi, float lens_lo, float p0, float p1, float s, float t, float o[3],
                              float d[3])
{
    const float off0 = __builtin_fmaf(p0, lens_hi, p0 * lens_lo), off1 = __builtin_fmaf(p1, lens_hi, p1 * lens_lo);
RF_UNROLL
    for (int k = 0; k < 3; ++k) {
        o[k] = (cam.origin0[k] + cam.f[12 + k] * off0) + cam.f[15 + k] * off1;
        d[k] = ((cam.lower_left0[k] + cam.f[3 + k] * s) + cam.f[6 + k] * t) - o[k];
    }
}

// world.hit (world.py:126-167) over the environment's n_shapes <= NS shapes + the checker colour of the closest hit; kDoubt
// poisons the sample
template <int NS, class Shapes>
RF_HD int world_hit_dense(const Shapes *sc, int n_shapes, const float o[3], const float d[3], float t_min, float t_max,
                          HitRec &rec, bool &doubt)
{
    const float a = dot3(d, d);
    float closest = t_max;
    int any = kMiss;
RF_UNROLL
    for (int i = 0; i < NS; ++i) {
        if (i >= n_shapes) // per environment: uniform
            break;
        if (sc[i].type == 0) { // per environment: uniform
            const float centre[3] = {sc[i].p[0], sc[i].p[1], sc[i].p[2]};
            HitRec tmp;
            const int h = sphere_hit_dense(centre, sc[i].k[0], sc[i].k[1], o, d, a, t_min, closest, tmp);
            if (h == kDoubt)
                return kDoubt;
            if (h == kHit) {
                any = kHit;
                closest = tmp.t;
                rec = tmp;
                rec.red = sphere_red_dense(tmp.n, sc[i].p[4], sc[i].p[5], doubt);
            }
        } else { // rectangle.py:49-99: float32 in the reference
            const float rp[7] = {sc[i].p[0], sc[i].p[1], sc[i].p[2], sc[i].p[3], sc[i].p[4], sc[i].p[5], sc[i].p[6]};
            const float kk[5] = {sc[i].k[0], sc[i].k[1], sc[i].k[2], sc[i].k[3], sc[i].k[4]};
            HitRec tmp;
            if (rectangle_hit(rp, o, d, t_min, closest, tmp)) {
                any = kHit;
                closest = tmp.t;
                rec = tmp;
                rec.red = rectangle_red_dense(rp, kk, tmp.p[0], tmp.p[1], doubt);
            }
        }
    }
    return any;
}

// one pixel of device_render (render.py:31-85); false: the pixel abstains (g, cr, cg, cb are then meaningless)
template <bool POW2, int NS, bool SIMPLE, class Cam, class Shapes>
RF_HD bool render_pixel_dense(Rng &g, int x, int y, int spp, const Cam &cam, const Shapes *sc, int n_shapes,
                              const FrameConst &fc, float &cr, float &cg, float &cb)
{
    cr = cg = cb = 0.0f;
    bool doubt = false;
    const float xf = (float)x, yf = (float)y;
    for (int k = 0; k < spp; ++k) {
        float s, t;
        sample_coords<POW2>(g, x, y, xf, yf, fc, s, t); // render.py:61-66
        float p0, p1;
        disc_sample(g, p0, p1);
        float o[3], d[3];
        if (SIMPLE)
            general_ray_simple(cam, cam.lens_hi, cam.lens_lo, p0, p1, s, t, o, d);
        else
            general_ray_any(cam, p0, p1, s, t, o, d);
        float ar = 1.0f, ag = 1.0f, ab = 1.0f;
        bool black = false;
        for (int bounce = 0;;) { // physics.py:95-145 find_colour
            HitRec rec;
            const int hit = world_hit_dense<NS>(sc, n_shapes, o, d, 0.001f, 1000000.0f, rec, doubt);
            if (hit == kDoubt)
                doubt = true;
            if (hit != kHit)
                break;
            float q0, q1, q2;
            sphere_sample(g, q0, q1, q2);
            scatter_step(rec, q0, q1, q2, o, d, ar, ag, ab);
            if (++bounce == kMaxBounces) {
                black = true;
                break;
            }
        }
        Colour c = sky_colour(d, ar, ag, ab);
        if (black)
            c = Colour{0.0f, 0.0f, 0.0f};
        cr = add2(cr, c.r);
        cg = add2(cg, c.g);
        cb = add2(cb, c.b);
    }
    return !doubt;
}

} // namespace rf
