// hostsim.cpp -- TEST INFRASTRUCTURE.  Compiles the render kernel's per-pixel
// arithmetic (reinfocus_amd/csrc/rf_math.h, the exact text the gfx950 kernel
// inlines) for the host, so that CPU-only tests can compare its specialisations
// (AXIS / POW2 / integer uniform / checker table) with the oracle before any GPU
// time is spent.  Never loaded by the product package.
#include <math.h>
#include <string.h>
#include <stdint.h>

#define RF_HOSTSIM 1 // (rf_general_dense.h: its 1-ulp approximations can be nudged from here)
#include "../../reinfocus_amd/csrc/rf_math.h"
#include "../../reinfocus_amd/csrc/rf_jump.h"
#include "../../reinfocus_amd/csrc/rf_general.h"
#include "../../reinfocus_amd/csrc/rf_general_dense.h"
#include "../../reinfocus_amd/csrc/rf_general_chunk.h"
#include "../gpucheck/probe_general.h"

using namespace rf;

thread_local unsigned rf::g_dense_perturb = 0;

static CheckerTable host_checker_table()
{
    CheckerTable t{0};
    for (int k = 1; k <= 32; ++k) {
        const float u = (float)k / 32.0f;
        const double si = ((double)32.0f * 3.14159265358979323846) * (double)u;
        if (sin(si) < 0.0)
            t.neg_mask |= (1ull << k);
    }
    return t;
}

extern "C" {

// mode bit0 = AXIS, bit1 = POW2
int hs_render(uint8_t *frames, int n, int h, int w, int spp, const float *cam_dyn,
              const float *rect, const float *origin, const float *u, const float *v,
              double lens_radius, uint64_t *states, int mode)
{
    const CheckerTable tab = host_checker_table();
    CamStatic cs{origin[0], origin[1], origin[2], u[0], u[1], u[2], v[0], v[1], v[2], lens_radius, 0.0f, 0.0f, 0};
    cs.lens_hi = (float)lens_radius;
    cs.lens_lo = (float)(lens_radius - (double)cs.lens_hi);
    cs.lens_f32 = (mode >> 2) & 1; // bit 2: the float32 lens offset (the caller has checked the radius)
    const float scale = (float)(255.0 / (double)spp);
    const FrameConst fc = frame_const(h, w);
    for (int e = 0; e < n; ++e) {
        const PixelEnv env = make_pixel_env(cam_dyn + 9 * e, rect + 2 * e);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const long pix = ((long)e * h + y) * w + x;
                Rng g = rng_load(states[2 * pix], states[2 * pix + 1]);
                float cr, cg, cb;
                switch (mode & 3) {
                case 3: render_pixel<true, true>(g, x, y, spp, fc, env, cs, tab, cr, cg, cb); break;
                case 1: render_pixel<true, false>(g, x, y, spp, fc, env, cs, tab, cr, cg, cb); break;
                case 2: render_pixel<false, true>(g, x, y, spp, fc, env, cs, tab, cr, cg, cb); break;
                default: render_pixel<false, false>(g, x, y, spp, fc, env, cs, tab, cr, cg, cb); break;
                }
                states[2 * pix] = rng_s0(g);
                states[2 * pix + 1] = rng_s1(g);
                frames[pix * 3 + 0] = (uint8_t)(cr * scale);
                frames[pix * 3 + 1] = (uint8_t)(cg * scale);
                frames[pix * 3 + 2] = (uint8_t)(cb * scale);
            }
    }
    return 0;
}

// the general renderer's per-pixel arithmetic (rf_general.h) on the host
int hs_render_general(uint8_t *frames, int n, int h, int w, int spp, const double *cameras, const float *params,
                      const int32_t *types, const int32_t *sizes, int most, int width, uint64_t *states)
{
    const float scale = (float)(255.0 / (double)spp);
    const bool pow2 = h > 0 && w > 0 && (h & (h - 1)) == 0 && (w & (w - 1)) == 0; // the instance rf_render_general launches
    for (int e = 0; e < n; ++e)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const long pix = ((long)e * h + y) * w + x;
                Rng g = rng_load(states[2 * pix], states[2 * pix + 1]);
                float cr, cg, cb;
                if (pow2)
                    render_pixel_general<true>(g, x, y, h, w, spp, general_camera(cameras + (long)e * 19),
                                               params + ((long)e * most) * width, types + (long)e * most, sizes[e], width, cr,
                                               cg, cb);
                else
                    render_pixel_general<false>(g, x, y, h, w, spp, general_camera(cameras + (long)e * 19),
                                                params + ((long)e * most) * width, types + (long)e * most, sizes[e], width, cr,
                                                cg, cb);
                states[2 * pix] = rng_s0(g);
                states[2 * pix + 1] = rng_s1(g);
                frames[pix * 3 + 0] = (uint8_t)(cr * scale);
                frames[pix * 3 + 1] = (uint8_t)(cg * scale);
                frames[pix * 3 + 2] = (uint8_t)(cb * scale);
            }
    return 0;
}

// returns the number of mismatches between the integer and literal uniform conversions
long hs_check_uniform(const uint64_t *words, long n)
{
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        float a = unit_f32_int(words[i]);
        float b = unit_f32_literal(words[i]);
        float c = unit_f32_scaled64((uint32_t)(words[i] >> 32), (uint32_t)words[i]) * kTwoM64;
        if (!(a == b) || !(c == b))
            ++bad;
    }
    return bad;
}

// pixel_coord_pow2 vs pixel_coord_literal over every x in [0, w) for the given xis
long hs_check_pixel_coord(const float *xis, long n, int w)
{
    long bad = 0;
    const float inv_w = 1.0f / (float)w;
    for (int x = 0; x < w; ++x)
        for (long i = 0; i < n; ++i)
            if (!(pixel_coord_pow2(x, xis[i], inv_w) == pixel_coord_literal(x, xis[i], w)))
                ++bad;
    return bad;
}

// pixel_coord_div vs pixel_coord_literal: every frame size w in [w_lo, w_hi], every x in [0, w)
// (strided for large w unless every_x, so that each size costs about the same), all the given xis
long hs_check_pixel_coord_div(const float *xis, long n, int w_lo, int w_hi, int every_x)
{
    long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic)
    for (int w = w_lo; w <= w_hi; ++w) {
        const double rw = 1.0 / (double)w;
        const int step = (w > 64 && !every_x) ? w / 64 : 1;
        for (int x = 0; x < w; x += (x + step < w || x == w - 1) ? step : (w - 1 - x))
            for (long i = 0; i < n; ++i)
                if (!(pixel_coord_div(x, xis[i], (double)w, rw) == pixel_coord_literal(x, xis[i], w)))
                    ++bad;
    }
    return bad;
}

// lens_offset's float32 form against the literal float64 form for every possible disc
// coordinate (multiples of 2^-24 in [-1, 0), of 2^-23 in [0, 1]); returns the mismatches
long hs_check_lens(double radius)
{
    CamStatic lit{0, 0, 0, 1, 0, 0, 0, 1, 0, radius, 0.0f, 0.0f, 0};
    CamStatic f32 = lit;
    f32.lens_hi = (float)radius;
    f32.lens_lo = (float)(radius - (double)f32.lens_hi);
    f32.lens_f32 = 1;
    long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long k = 0; k <= (1l << 24) + (1l << 23); ++k) {
        const float p = k < (1l << 24) ? (float)((double)k / 16777216.0 - 1.0)
                                       : (float)((double)(k - (1l << 24)) / 8388608.0);
        if (!(lens_offset(p, f32) == lens_offset(p, lit)))
            ++bad;
    }
    return bad;
}

// checker_sign(u) * checker_sign(v) > 0  vs  sin(32 pi u) * sin(32 pi v) > 0
long hs_check_checker(const float *us, const float *vs, long n)
{
    const CheckerTable tab = host_checker_table();
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        const double s0 = ((double)32.0f * 3.14159265358979323846) * (double)us[i];
        const double s1 = ((double)32.0f * 3.14159265358979323846) * (double)vs[i];
        const bool lit = sin(s0) * sin(s1) > 0.0;
        const bool fast = (checker_sign(us[i], tab) * checker_sign(vs[i], tab)) > 0;
        if (lit != fast || lit != checker_red(us[i], vs[i], tab))
            ++bad;
    }
    return bad;
}

// The rejection loops' fast accept decision (approximate test + exact band) against
// the literal one, for caller-supplied 64-bit draw words.  dims = 2 (disc) or 3 (sphere).
// words: n * dims u64.  Returns mismatches; *in_band counts how often the band path ran.
long hs_check_accept(const uint64_t *words, long n, int dims, long *in_band)
{
    long bad = 0, band = 0;
    for (long i = 0; i < n; ++i) {
        float lit[3] = {0, 0, 0}, apx[3] = {0, 0, 0};
        for (int k = 0; k < dims; ++k) {
            const uint64_t w = words[i * dims + k];
            lit[k] = unit_f32_literal(w) * 2.0f - 1.0f;
            apx[k] = approx_pm1((uint32_t)(w >> 32));
        }
        bool want, got;
        float sq;
        if (dims == 2) {
            const float d0 = lit[0] * lit[0], d1 = lit[1] * lit[1];
            want = (double)(d0 + d1) < 1.0;
            sq = __builtin_fmaf(apx[0], apx[0], apx[1] * apx[1]);
        } else {
            want = (double)sq_len(lit[0], lit[1], lit[2]) < 1.0;
            sq = __builtin_fmaf(apx[0], apx[0], __builtin_fmaf(apx[1], apx[1], apx[2] * apx[2]));
        }
        got = sq < 1.0f - kAcceptBand;
        if (!got && sq < 1.0f + kAcceptBand) {
            ++band;
            float e[3] = {0, 0, 0};
            for (int k = 0; k < dims; ++k)
                e[k] = exact_pm1((uint32_t)(words[i * dims + k] >> 32), (uint32_t)words[i * dims + k]);
            if (dims == 2) {
                const float d0 = e[0] * e[0], d1 = e[1] * e[1];
                got = d0 + d1 < 1.0f;
            } else {
                got = sq_len(e[0], e[1], e[2]) < 1.0f;
            }
        }
        if (got != want)
            ++bad;
        // the squared-length form (disc_attempt_sq / sphere_attempt_sq): the approximate length outside the band,
        // the reference's own inside; its comparison with 1 must be the reference's decision
        {
            float sq2 = sq;
            if (__builtin_fabsf(sq2 - 1.0f) < kAcceptBand) {
                float e[3] = {0, 0, 0};
                for (int k = 0; k < dims; ++k)
                    e[k] = exact_pm1((uint32_t)(words[i * dims + k] >> 32), (uint32_t)words[i * dims + k]);
                if (dims == 2) {
                    const float d0 = e[0] * e[0], d1 = e[1] * e[1];
                    sq2 = d0 + d1;
                } else {
                    sq2 = sq_len(e[0], e[1], e[2]);
                }
            }
            if ((sq2 < 1.0f) != want)
                ++bad;
        }
    }
    *in_band = band;
    return bad;
}

// div_by_const against IEEE division: for each divisor d, numerators a = frac * d for
// n_frac fractions in [0, 1] plus the exact multiples k/32 and their neighbours.
long hs_check_div(const float *dens, long n_den, const float *fracs, long n_frac)
{
    long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long i = 0; i < n_den; ++i) {
        const float d = dens[i], rd = 1.0f / d;
        for (long j = 0; j < n_frac; ++j) {
            const float a = fracs[j] * d;
            if ((a == 0.0f || a >= d * 2.98023223876953125e-08f) && !(div_by_const(a, d, rd) == a / d))
                ++bad;
        }
        for (int k = 0; k <= 32; ++k) {
            float a = d * ((float)k / 32.0f);
            for (int s = -3; s <= 3; ++s) {
                float b = a;
                for (int t = 0; t < (s < 0 ? -s : s); ++t)
                    b = __builtin_nextafterf(b, s < 0 ? -1.0f : 1e30f);
                // the kernel's numerator is p + half with |p| <= half: 0 or >= 2^-25 * d
                if ((b == 0.0f || b >= d * 2.98023223876953125e-08f) && !(div_by_const(b, d, rd) == b / d))
                    ++bad;
            }
        }
    }
    return bad;
}

// f32 sky formulas against the literal f64 chain for every float in [lo_bits, hi_bits]
// (two ranges: the non-negative and the negative floats up to 1 + 8 ulp).
long hs_check_sky(void)
{
    long bad = 0;
    const uint32_t top = 0x3F800008u; // 1 + 8 ulp
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long i = 0; i <= 2 * (long)top + 1; ++i) {
        uint32_t bits = (i <= (long)top) ? (uint32_t)i : (0x80000000u | (uint32_t)(i - top - 1));
        float ud;
        __builtin_memcpy(&ud, &bits, 4);
        const double T = sky_t(ud);
        const float white_ref = (float)(1.0 - T);
        if (!(white_ref == sky_white(ud)))
            ++bad;
        const float ks[3] = {0.5f, 0.7f, 1.0f};
        for (int k = 0; k < 3; ++k) {
            const float ref = sky_channel_literal(T, white_ref, ks[k]);
            const float got = add2(sky_white(ud), sky_blue(ud, kSkyHalf[k]));
            if (!(ref == got))
                ++bad;
        }
    }
    return bad;
}

// GF(2) jump tables: state[index] computed directly vs numba's sequential definition
int hs_state_at(uint64_t seed, uint64_t index, uint64_t out[2])
{
    std::vector<Mat128> tables;
    if (!h_build_jump_tables(48, tables))
        return -1;
    S128 s = h_splitmix(seed);
    for (int k = 0; k < 48; ++k)
        if ((index >> k) & 1)
            s = h_matvec(tables[k], s);
    out[0] = s.s0;
    out[1] = s.s1;
    return 0;
}

// host side of tests/gpucheck's probes (glibc's libm): same functions, same operands
int hs_probe_f64(int op, const double *a, const double *b, double *out, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i)
        out[i] = probe::f64_op(op, a[i], b[i]);
    return 0;
}

int hs_probe_uv(const float *normals, float *uv, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i)
        probe::sphere_uv(normals + 3 * i, uv + 2 * i);
    return 0;
}

// the reference's expression itself: sign of sin(fl64(fl64(f * pi) * u)) with the real libm sin
int hs_probe_checker_literal(const float *f, const float *u, int *sign, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i) {
        const double s = sin(((double)f[i] * rf::kPi) * (double)u[i]);
        sign[i] = s > 0.0 ? 1 : (s < 0.0 ? -1 : 0);
    }
    return 0;
}

// checker colour of sphere hits: mode 0 = fast path + fallback, 1 = float64 expressions alone
int hs_probe_sphere_red(int mode, const float *normals, const float *fu, const float *fv, int *red, uint64_t n)
{
#pragma omp parallel for
    for (int64_t i = 0; i < (int64_t)n; ++i)
        red[i] = mode ? probe::sphere_red_exact(normals + 3 * i, fu[i], fv[i])
                      : probe::sphere_red_fast(normals + 3 * i, fu[i], fv[i]);
    return 0;
}

int hs_probe_uv_approx(const float *normals, float *uv, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i)
        probe::sphere_uv_fast(normals + 3 * i, uv + 2 * i);
    return 0;
}

// sphere.hit (sphere.py:40-103) as the reference writes it -- float64 roots for every ray with a non-negative
// discriminant -- next to rf::sphere_hit, which decides the certain misses in float32 first: both on the same rays.
// rec = p[3], n[3], t; returns the number of rays on which they differ (hit flag, or any bit of a hit's record).
static bool sphere_hit_literal(const float *sp, const float o[3], const float d[3], float t_min, float t_max, float rec[7])
{
    const float oc[3] = {o[0] - sp[0], o[1] - sp[1], o[2] - sp[2]};
    const float a = rf::dot3(d, d), b = rf::dot3(oc, d), c = rf::dot3(oc, oc) - sp[3] * sp[3];
    const float disc = b * b - a * c;
    if (disc < 0)
        return false;
    const double sqrtd = sqrt((double)disc);
    double root = (-(double)b - sqrtd) / (double)a;
    if (root < (double)t_min || (double)t_max < root) {
        root = (-(double)b + sqrtd) / (double)a;
        if (root < (double)t_min || (double)t_max < root)
            return false;
    }
    const float inv_r = (float)(1.0 / (double)sp[3]);
    for (int k = 0; k < 3; ++k) {
        rec[k] = rf::add2(o[k], (float)((double)d[k] * root));
        rec[3 + k] = (rec[k] - sp[k]) * inv_r;
    }
    rec[6] = (float)root;
    return true;
}

long hs_check_sphere_hit(const float *spheres, const float *origins, const float *dirs, float t_min, float t_max, long n,
                         long *hits, long *shortcuts)
{
    long bad = 0, n_hit = 0, n_short = 0;
#pragma omp parallel for reduction(+ : bad, n_hit, n_short)
    for (long i = 0; i < n; ++i) {
        const float *sp = spheres + 4 * i, *o = origins + 3 * i, *d = dirs + 3 * i;
        float want[7] = {0, 0, 0, 0, 0, 0, 0};
        const bool hit = sphere_hit_literal(sp, o, d, t_min, t_max, want);
        rf::HitRec r;
        const bool got = rf::sphere_hit(sp, o, d, t_min, t_max, r);
        const float have[7] = {r.p[0], r.p[1], r.p[2], r.n[0], r.n[1], r.n[2], r.t};
        if (got != hit || (hit && memcmp(want, have, sizeof(want)) != 0))
            ++bad;
        n_hit += hit ? 1 : 0;
        // (how often the float32 test is the one that answers: the same expression as in rf_general.h)
        const float oc[3] = {o[0] - sp[0], o[1] - sp[1], o[2] - sp[2]};
        const float a = rf::dot3(d, d), b = rf::dot3(oc, d), c = rf::dot3(oc, oc) - sp[3] * sp[3];
        const float disc = b * b - a * c, reach = b + t_min * a;
        if (!(disc < 0) && a > 0.0f && b > 0.0f && disc < (reach * reach) * 0.99999904632568359375f)
            ++n_short;
    }
    *hits = n_hit;
    *shortcuts = n_short;
    return bad;
}

// rf_general_dense.h sphere_hit_dense (float32 / double-float with abstention) next to the literal float64 form on the
// same rays: wherever it does not abstain, the answer and every bit of a hit's record must be the literal one's.
// perturb != 0: the 1-ulp approximations (sqrt, reciprocals) are nudged by -1 / 0 / +1 ulp at random.
// counts = {hits, misses decided, abstentions}; max_err = the largest relative error of the double-float root against
// long double (x86: 64 significant bits) over the rays whose root was evaluated, in units of the bound it is used with
// (tol / 4: must stay below 1).
long hs_check_sphere_hit_dense(const float *spheres, const float *origins, const float *dirs, float t_min, float t_max, long n,
                               unsigned perturb, long *counts, double *max_err)
{
    long bad = 0, n_hit = 0, n_miss = 0, n_doubt = 0;
    double worst = 0.0;
#pragma omp parallel for reduction(+ : bad, n_hit, n_miss, n_doubt) reduction(max : worst)
    for (long i = 0; i < n; ++i) {
        const float *sp = spheres + 4 * i, *o = origins + 3 * i, *d = dirs + 3 * i;
        float want[7] = {0, 0, 0, 0, 0, 0, 0};
        const bool hit = sphere_hit_literal(sp, o, d, t_min, t_max, want);
        rf::g_dense_perturb = perturb ? (perturb + (unsigned)i * 2654435761u) | 1u : 0u;
        rf::HitRec r;
        const float centre[3] = {sp[0], sp[1], sp[2]};
        const rf::ShapeConst sc = rf::shape_const(sp, 4, 0);
        const int got = rf::sphere_hit_dense(centre, sc.k[0], sc.k[1], o, d, rf::dot3(d, d), t_min, t_max, r);
        if (got == rf::kDoubt) {
            ++n_doubt;
        } else {
            const float have[7] = {r.p[0], r.p[1], r.p[2], r.n[0], r.n[1], r.n[2], r.t};
            if ((got == rf::kHit) != hit || (hit && memcmp(want, have, sizeof(want)) != 0))
                ++bad;
            n_hit += hit ? 1 : 0;
            n_miss += hit ? 0 : 1;
        }
        // the double-float root itself against long double, wherever it is evaluated
        const float oc[3] = {o[0] - sp[0], o[1] - sp[1], o[2] - sp[2]};
        const float a = rf::dot3(d, d), b = rf::dot3(oc, d), c = rf::dot3(oc, oc) - sp[3] * sp[3];
        const float disc = b * b - a * c;
        if (disc > 0 && rf::magnitude_within(a, -40, 40) && rf::magnitude_within(disc, -60, 60) &&
            (b == 0.0f || rf::magnitude_within(b, -60, 40))) {
            for (int sgn = 1; sgn >= -1; sgn -= 2) {
                float qh, ql, tol;
                if (!rf::sphere_root_df(a, b, disc, (float)sgn, qh, ql, tol))
                    continue;
                const long double exact = (-(long double)b - (long double)sgn * sqrtl((long double)disc)) / (long double)a;
                const long double err = fabsl(((long double)qh + (long double)ql) - exact) / fabsl(exact);
                const double units = (double)(err / ((long double)tol * 0.25L)); // in units of the bound (tol = 4x bound)
                worst = units > worst ? units : worst;
            }
        }
    }
    rf::g_dense_perturb = 0;
    counts[0] = n_hit;
    counts[1] = n_miss;
    counts[2] = n_doubt;
    *max_err = worst;
    return bad;
}

// rf_general_dense.h render_pixel_dense on the host: frames / states of the pixels that do not abstain, and which did
// (abstained[pix] = 1: frame bytes and state untouched).  NS = most in {1, 2, 3}, any counts up to it; cameras with canonical axes
// take the SIMPLE instance unless `simple` is 0 (the float32 lens offset is then assumed exact for their radii: the reference's
// default aperture), the others the float64 lens products.
int hs_render_general_dense(uint8_t *frames, int n, int h, int w, int spp, const double *cameras, const float *params,
                            const int32_t *types, const int32_t *sizes, int most, int width, uint64_t *states,
                            uint8_t *abstained, unsigned perturb, int simple)
{
    const float scale = (float)(255.0 / (double)spp);
    const bool pow2 = h > 0 && w > 0 && (h & (h - 1)) == 0 && (w & (w - 1)) == 0;
    if (most < 1 || most > 3 || h > 4096 || w > 4096)
        return -1;
    const FrameConst fc = frame_const(h, w);
    for (int e = 0; e < n; ++e)
        simple = simple && camera_axes_simple(general_camera(cameras + (long)e * 19));
    for (int e = 0; e < n; ++e) {
        if (sizes[e] < 0 || sizes[e] > most)
            return -1;
        const GeneralCamera cam = general_camera(cameras + (long)e * 19);
        ShapeConst sc[3];
        for (int i = 0; i < most; ++i)
            sc[i] = shape_const(params + ((long)e * most + i) * width, width, i < sizes[e] ? types[(long)e * most + i] : 1);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const long pix = ((long)e * h + y) * w + x;
                Rng g = rng_load(states[2 * pix], states[2 * pix + 1]);
                float cr, cg, cb;
                rf::g_dense_perturb = perturb ? (perturb + (unsigned)pix * 2654435761u) | 1u : 0u;
                bool keep;
#define HS_DENSE(P, N, S) keep = render_pixel_dense<P, N, S>(g, x, y, spp, cam, sc, sizes[e], fc, cr, cg, cb)
#define HS_DENSE_N(P, S) do { if (most == 1) HS_DENSE(P, 1, S); else if (most == 2) HS_DENSE(P, 2, S); else HS_DENSE(P, 3, S); } while (0)
                if (pow2 && simple) HS_DENSE_N(true, true);
                else if (pow2) HS_DENSE_N(true, false);
                else if (simple) HS_DENSE_N(false, true);
                else HS_DENSE_N(false, false);
#undef HS_DENSE_N
#undef HS_DENSE
                abstained[pix] = keep ? 0 : 1;
                if (!keep)
                    continue;
                states[2 * pix] = rng_s0(g);
                states[2 * pix + 1] = rng_s1(g);
                frames[pix * 3 + 0] = (uint8_t)(cr * scale);
                frames[pix * 3 + 1] = (uint8_t)(cg * scale);
                frames[pix * 3 + 2] = (uint8_t)(cb * scale);
            }
    }
    rf::g_dense_perturb = 0;
    return 0;
}

int hs_probe_checker(const float *f, const float *u, int *sign, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i)
        sign[i] = rf::checker_sign_general(f[i], u[i]);
    return 0;
}

// rf_render_general's environments per launch of its listed kernels (rf_general_chunk.h): kind 1 = one-shape, 2 = dense
int hs_general_chunk(int kind, int h, int w, unsigned long long *blocks_per_env)
{
    bool flag = false;
    const uint64_t per_env = kind == 2 ? dense_blocks_per_env(h, w, &flag) : one_blocks_per_env(h, w, 3, &flag);
    *blocks_per_env = per_env;
    return general_listed_chunk((uint64_t)h * (uint64_t)w, per_env);
}

// focus_finalize / env_variance: the variance from the exact integer sums (rf_math.h variance_from_sums)
double hs_variance_from_sums(unsigned long long npix, unsigned long long s1, unsigned long long s2)
{
    return variance_from_sums(npix, s1, s2);
}

} // extern "C"
