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

// Value-preserving optimisation fence: stops hipcc from re-deriving a 32-bit word as
// "high half of a 64-bit value" (it then expands u32->f32 as a generic u64->f32).
#if defined(__HIP_DEVICE_COMPILE__)
#define RF_OPAQUE32(x) asm("" : "+v"(x))
#else
#define RF_OPAQUE32(x) ((void)0)
#endif

namespace rf {

// ---------------------------------------------------------------------------
// xoroshiro128+ (numba.cuda.random xoroshiro128p_next; graphics/random.py:33)
//
// The state is kept as four 32-bit words and stepped with v_alignbit_b32 funnel
// shifts: gfx950 runs 64-bit shifts (v_lshlrev_b64 / v_lshrrev_b64) at a fraction of
// the 32-bit rate, and the RNG is ~60 % of the kernel's instructions.
// ---------------------------------------------------------------------------
struct Rng {
    uint32_t a_lo, a_hi; // s0
    uint32_t b_lo, b_hi; // s1
};

RF_HD Rng rng_load(uint64_t s0, uint64_t s1)
{
    return Rng{(uint32_t)s0, (uint32_t)(s0 >> 32), (uint32_t)s1, (uint32_t)(s1 >> 32)};
}
RF_HD uint64_t rng_s0(const Rng &g) { return ((uint64_t)g.a_hi << 32) | g.a_lo; }
RF_HD uint64_t rng_s1(const Rng &g) { return ((uint64_t)g.b_hi << 32) | g.b_lo; }

// ({hi, lo} >> s)[31:0], s in [0, 31]  (v_alignbit_b32)
RF_HD uint32_t funnel_r(uint32_t hi, uint32_t lo, uint32_t s)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, s);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (s & 31));
#endif
}

// result = s0 + s1 (as two words), then the xoroshiro128+ 55/14/36 state update:
// one v_lshl_add_u64 for the sum, v_alignbit_b32 funnel shifts for everything else
// (10 VALU instructions; checked in the ISA, see profiles/HISTORY.md section 4.1).
RF_HD void rng_next(Rng &g, uint32_t &r_hi, uint32_t &r_lo)
{
    const uint32_t a_lo = g.a_lo, a_hi = g.a_hi, b_lo = g.b_lo, b_hi = g.b_hi;
    const uint64_t r = ((((uint64_t)a_hi) << 32) | a_lo) + ((((uint64_t)b_hi) << 32) | b_lo);
    r_lo = (uint32_t)r;
    r_hi = (uint32_t)(r >> 32);
    RF_OPAQUE32(r_lo);
    RF_OPAQUE32(r_hi);
    const uint32_t x_lo = b_lo ^ a_lo, x_hi = b_hi ^ a_hi; // s1 ^= s0
    // rotl(s0, 55) == rotr(s0, 9)
    const uint32_t ro_lo = funnel_r(a_hi, a_lo, 9), ro_hi = funnel_r(a_lo, a_hi, 9);
    // s1 << 14
#if defined(__HIP_DEVICE_COMPILE__)
    // one 64-bit shift issues in the time of one v_alignbit_b32 (tools/ubench)
    uint64_t sh;
    asm("v_lshlrev_b64 %0, 14, %1" : "=v"(sh) : "v"((((uint64_t)x_hi) << 32) | x_lo));
    const uint32_t sh_lo = (uint32_t)sh, sh_hi = (uint32_t)(sh >> 32);
#else
    const uint32_t sh_lo = x_lo << 14, sh_hi = funnel_r(x_hi, x_lo, 18);
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    // gfx950's three-input boolean op as a three-way xor (truth table 0x96): one instruction per word
    g.a_lo = __builtin_amdgcn_bitop3_b32(ro_lo, x_lo, sh_lo, 0x96);
    g.a_hi = __builtin_amdgcn_bitop3_b32(ro_hi, x_hi, sh_hi, 0x96);
#else
    g.a_lo = ro_lo ^ x_lo ^ sh_lo;
    g.a_hi = ro_hi ^ x_hi ^ sh_hi;
#endif
    // rotl(s1, 36) == rotl(swap halves, 4)
    g.b_hi = funnel_r(x_lo, x_hi, 28);
    g.b_lo = funnel_r(x_hi, x_lo, 28);
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

// General integer form, same value for every r: normalise the 53 kept bits, fold the
// bits below the top 32 into a sticky bit (legal because a normalised 32-bit word keeps
// its round bit at position >= 7), let v_cvt_f32_u32 do the single RNE, scale by an
// exact power of two.
RF_HD float unit_f32_int(uint64_t r)
{
    uint32_t hi = (uint32_t)(r >> 32);
    uint32_t lo = (uint32_t)r & 0xFFFFF800u; // the 11 bits numba shifts out never count
    int lz = clz32(hi);
    uint64_t y64 = (((uint64_t)hi << 32) | lo) << (lz & 63);
    uint32_t y = (uint32_t)(y64 >> 32);
    uint32_t rest = (uint32_t)y64;
    y |= (rest != 0u) ? 1u : 0u;
    return ldexp_pow2((float)y, -32 - lz);
}

// Fast form: 2^64 * unit value as ONE correctly rounded f32 (callers fold the exact 2^-64
// into their next fma).  {hi, lo & ~0x7FF} is an exact 53-bit integer in f64 -- one fma of
// two exact u32 conversions -- then a single f64->f32 RNE (the very cast numba performs)
// and an exact power-of-two scaling.  Four instructions of the 4-cycle class, no rare path;
// measured 2.6 % faster end to end than a two-f32 split with sticky bits (tools/ubench).
RF_HD float unit_f32_scaled64(uint32_t r_hi, uint32_t r_lo)
{
    // The same exact 53-bit integer without the two u32 -> f64 conversions (slow-path instructions on
    // gfx950, like the f64 fma): the bit patterns {0x45300000, hi} and {0x43300000, lo & ~0x7FF} ARE the
    // doubles 2^84 + hi 2^32 and 2^52 + (lo & ~0x7FF); (A - (2^84 + 2^52)) is exact (a multiple of 2^32
    // below 2^64), and adding B gives hi 2^32 + (lo & ~0x7FF) exactly (53 significant bits).  Two f64
    // additions and the one rounding f64 -> f32.
    const uint64_t a_bits = 0x4530000000000000ull | (uint64_t)r_hi;
    const uint64_t b_bits = 0x4330000000000000ull | (uint64_t)(r_lo & 0xFFFFF800u);
    double a, b;
    __builtin_memcpy(&a, &a_bits, 8);
    __builtin_memcpy(&b, &b_bits, 8);
    const double d = (a - 19342813118337666422669312.0 /* 2^84 + 2^52 */) + b;
    return (float)d; // < 2^64: the scale is the consumer's (an exact power of two in its fma or product)
}

constexpr float kTwoM64 = 5.421010862427522e-20f;  // 2^-64
constexpr float kTwoM63 = 1.0842021724855044e-19f; // 2^-63

// 2^64 * xoroshiro128p_uniform_float32
RF_HD float rng_uniform64(Rng &g)
{
    uint32_t hi, lo;
    rng_next(g, hi, lo);
    return unit_f32_scaled64(hi, lo);
}

RF_HD float rng_uniform(Rng &g) { return rng_uniform64(g) * kTwoM64; } // exact scaling

// ---------------------------------------------------------------------------
// ndarray.var() of a frame's Laplacian (vision.py:25) from its exact integer sums S1 = sum x, S2 = sum x^2 over N pixels:
// (N S2 - S1^2) / N^2.  The numerator is exact in 128 bits for any frame (x <= 255: below 2^16 N^2) and goes to float64
// in two limbs -- one rounding below 2^64 (every frame up to 3.3 10^7 pixels: the value of one conversion), two above.
// ---------------------------------------------------------------------------
RF_HD double variance_from_sums(unsigned long long npix, unsigned long long s1, unsigned long long s2)
{
    const unsigned __int128 num = (unsigned __int128)npix * s2 - (unsigned __int128)s1 * s1;
    const double hi = (double)(unsigned long long)(num >> 64), lo = (double)(unsigned long long)num;
    const double dn = (double)npix;
    return (hi * 18446744073709551616.0 + lo) / (dn * dn);
}

// ---------------------------------------------------------------------------
// scene parameters
// ---------------------------------------------------------------------------
// rect[e].half of an environment slot that a launch has to skip: the device-resident env step
// enqueues the auto-reset render for all n slots before it knows how many environments ended
// (no host round trip in the middle of a step); env_reset_kernel marks the unused slots.
constexpr uint32_t kSkipEnvBits = 0x7FC0DEADu; // a quiet NaN no arithmetic produces

struct CamStatic { // camera.py:39-52 FastGpuCameras minus the per-env array
    float ox, oy, oz;
    float ux, uy, uz;
    float vx, vy, vz;
    double lens_radius; // numpy.float64
    // lens_radius = lens_hi + lens_lo + (at most 2^-48 of it); lens_f32 != 0 when the host has
    // checked that the float32 form of lens_offset is exact for this radius (see there)
    float lens_hi, lens_lo;
    int lens_f32;
};

// camera.py:343 with vector.py:190: offset = float32(float64(p) * lens_radius) for a disc
// coordinate p.  p is RN32(2u - 1) for a float32 u in [0, 1]: a multiple of 2^-24 in [-1, 0) or
// of 2^-23 in [0, 1] -- 25 165 825 values.  With lens_radius split into two float32 terms,
// fma(p, hi, RN32(p * lo)) carries a relative error below 2^-47 before its single rounding, so
// it equals the doubly rounded reference unless p * lens_radius lies within 2^-47 of a float32
// rounding boundary; whether that happens for any of the possible p is a property of the radius
// alone, and the host checks all of them once per radius (rf_abi_ctx.hip lens_split; none for the
// reference's aperture 0.1).  Otherwise the literal float64 form is used.
// LENS: 1 / 0 = the form is fixed at compile time (the caller has looked at cs.lens_f32),
// -1 = decided at run time.
template <int LENS = -1>
RF_HD float lens_offset(float p, const CamStatic &cs)
{
    if (LENS == 1 || (LENS == -1 && cs.lens_f32))
        return __builtin_fmaf(p, cs.lens_hi, p * cs.lens_lo);
    return (float)((double)p * cs.lens_radius);
}

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

// sin(32 pi u) * sin(32 pi v) > 0 for u, v >= 0: the common case (neither 32u nor 32v an
// integer) is one parity test of trunc(32u) ^ trunc(32v); exact integers take the table.
RF_HD bool checker_red(float u, float v, const CheckerTable &tab)
{
    const float mu = u * 32.0f, mv = v * 32.0f;
    // The same parity without conversions and with one compare (float <-> int conversions and compares
    // issue on gfx950's slow path, float adds and logic ops do not): for 0 <= m <= 32, a = m + 2^23 is
    // m rounded to an integer r (ties to even) whose parity is the lowest bit of a's pattern; d = m - r
    // is exact (|d| <= 1/2), negative exactly when r = floor(m) + 1 and zero exactly when m is an
    // integer.  So parity(floor(m)) = (bits(a) ^ (bits(d) >> 31)) & 1; a zero product du * dv -- an
    // integer point, or an underflow for coordinates no hit produces -- takes the table path, which is
    // the literal logic and right for every input.
    const float au = mu + 8388608.0f, av = mv + 8388608.0f;
    const float du = mu - (au - 8388608.0f), dv = mv - (av - 8388608.0f);
    if (__builtin_expect(du * dv == 0.0f, 0))
        return (checker_sign(u, tab) * checker_sign(v, tab)) > 0;
    uint32_t bau, bav, bdu, bdv;
    __builtin_memcpy(&bau, &au, 4);
    __builtin_memcpy(&bav, &av, 4);
    __builtin_memcpy(&bdu, &du, 4);
    __builtin_memcpy(&bdv, &dv, 4);
    return (((bau ^ bav) ^ ((bdu ^ bdv) >> 31)) & 1u) == 0u;
}

// ---------------------------------------------------------------------------
// one sample: camera.get_ray (camera.py:307-350) + physics.fast_find_colour
// (physics.py:148-193) with rectangle.fast_hit (rectangle.py:102-148).
// Literal/general version: any camera frame.  Returns the sample colour.
// ---------------------------------------------------------------------------
RF_HD float add2(float a, float b) { return (0.0f + a) + b; }
// add2 for a first operand that is never -0 (then 0 + a == a bit for bit, NaN included): the
// running colour sums (start at +0, only ever grow by terms >= +0) and the sky's white part
// fma(-0.5, ud, 0.5), whose only zero is the exact 0.5 - 0.5 = +0.
RF_HD float add2_not_negzero(float a, float b) { return a + b; }
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

// --- rejection loops ---------------------------------------------------------------
// A rejected attempt only has to (a) advance the RNG state and (b) be rejected exactly
// when the reference rejects it.  (b) is decided from a cheap approximation of the
// candidate -- p~ = 2*RN24(r_hi)*2^-32 - 1, i.e. one v_cvt_f32_u32 and one fma per
// coordinate instead of the exact 53-bit -> f32 conversion -- whenever the approximate
// squared length is further than kAcceptBand from 1.  |p~ - p| <= 1.8e-7 per coordinate, so
// the approximate and the reference's rounded squared lengths differ by < 1.5e-6 (three
// coordinates, all roundings included); kAcceptBand = 2^-17 = 7.6e-6 leaves a 5x margin.
// Inside the band (probability ~1e-5 per attempt) the exact expression decides.  The
// accepted attempt is converted exactly after the loop, once.
constexpr float kTwoM31 = 4.656612873077393e-10f; // 2^-31

// gfx950 issues v_cvt_f32_u32 on its slow VALU path (4.3 cycles per wave, like every conversion,
// compare, shift-left and three-operand integer op) but v_lshrrev_b32 and v_fma_f32 on the fast one
// (2.4-2.9 cycles, and float fast-path ops overlap with slow-path ones): tools/ubench/pairbench.
// So the candidate is built without a conversion: the top 23 bits k of the draw's high word, read as
// the bits of a float, are the subnormal k 2^-149, and fma(k 2^-149, 2^127, -1) = k 2^-22 - 1 exactly
// (23 + 1 bits; f32 subnormals are not flushed: tests/gpucheck checks every k on the GPU).
// |p~ - p| < 2^-22 (truncation instead of rounding), so the approximate and the reference's squared
// lengths differ by < 3 (2 * 2.4e-7 + 1.2e-7 + 1.2e-7) + 3.6e-7 = 3e-6 (three coordinates: truncation,
// the reference's own roundings of the coordinate and of its square, the rounding of the approximate
// sum); kAcceptBand = 2^-16 = 1.5e-5 leaves a 5x margin.
constexpr float kAcceptBand = 1.52587890625e-05f; // 2^-16
RF_HD float approx_pm1(uint32_t r_hi)
{
    const uint32_t k = r_hi >> 9;
    float sub;
    __builtin_memcpy(&sub, &k, 4);
    return __builtin_fmaf(sub, 1.7014118346046923e+38f /* 2^127 */, -1.0f);
}
// == RN(xi*2f - 1f): the scalings by powers of two are exact
RF_HD float exact_pm1(uint32_t r_hi, uint32_t r_lo)
{
    return __builtin_fmaf(unit_f32_scaled64(r_hi, r_lo), kTwoM63, -1.0f);
}

// One attempt of camera.py:229-252 random_in_unit_disc: two draws (raw words kept in
// w = {ah, al, bh, bl}); true when the reference accepts, i.e. p0*p0 + p1*p1 < 1 in f32.
RF_HD bool disc_attempt(Rng &g, uint32_t w[4])
{
    rng_next(g, w[0], w[1]);
    rng_next(g, w[2], w[3]);
    const float ta = approx_pm1(w[0]), tb = approx_pm1(w[2]);
    const float sq = __builtin_fmaf(ta, ta, tb * tb);
    bool accept = sq < 1.0f - kAcceptBand;
    if (__builtin_expect(!accept && sq < 1.0f + kAcceptBand, 0)) {
        const float e0 = exact_pm1(w[0], w[1]), e1 = exact_pm1(w[2], w[3]);
        const float d0 = e0 * e0, d1 = e1 * e1;
        accept = d0 + d1 < 1.0f;
    }
    return accept;
}
// The same attempt, returning a squared length whose comparison with 1 IS the reference's decision: the
// approximate one outside the band (where it is on the same side of 1 as the reference's), the reference's own
// inside.  For callers that want the decision as a lane mask after a join: `sq < 1` is then a single compare
// (render_kernel_coop2, RF_MASKS).
RF_HD float disc_attempt_sq(Rng &g, uint32_t w[4])
{
    rng_next(g, w[0], w[1]);
    rng_next(g, w[2], w[3]);
    const float ta = approx_pm1(w[0]), tb = approx_pm1(w[2]);
    float sq = __builtin_fmaf(ta, ta, tb * tb);
    if (__builtin_expect(__builtin_fabsf(sq - 1.0f) < kAcceptBand, 0)) {
        const float e0 = exact_pm1(w[0], w[1]), e1 = exact_pm1(w[2], w[3]);
        const float d0 = e0 * e0, d1 = e1 * e1;
        sq = d0 + d1;
    }
    return sq;
}

RF_HD void disc_finish(const uint32_t w[4], float &p0, float &p1)
{
    p0 = exact_pm1(w[0], w[1]);
    p1 = exact_pm1(w[2], w[3]);
}

// The in-place rejection loops of render_kernel make two attempts per trip: a launch of few blocks is bound by one wave's
// latency, and a taken branch costs that wave more than the instructions around it (1 x 300^2 x 100: 252 -> 245 us per step;
// three per trip: no further gain; profiles/r06_ab.txt section 9).
#ifndef RF_LOOP_UNROLL
#define RF_LOOP_UNROLL 2
#endif
RF_HD void disc_sample(Rng &g, float &p0, float &p1)
{
    uint32_t w[4];
    for (;;) {
        if (disc_attempt(g, w))
            break;
#if RF_LOOP_UNROLL >= 2
        if (disc_attempt(g, w))
            break;
#endif
    }
    disc_finish(w, p0, p1);
}

// sq_len of the three exact components of a sphere candidate, one component at a time: the block that calls this is
// taken once in ~10^4 attempts, and with all three conversions in flight it is where the render kernels' register
// budget spills
RF_HD float sq_len_exact_sequential(const uint32_t w[6])
{
    float aa = exact_pm1(w[0], w[1]);
    aa = aa * aa;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(aa));
#endif
    float bb = exact_pm1(w[2], w[3]);
    bb = aa + bb * bb; // (not contracted: -ffp-contract=off)
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(bb));
#endif
    const float cc = exact_pm1(w[4], w[5]);
    return bb + cc * cc;
}

// One attempt of physics.py:20-44 random_in_unit_sphere: three draws (raw words kept in
// w = {ah, al, bh, bl, ch, cl}); true when the reference accepts the candidate, i.e. when
// float32(q0**2) + float32(q1**2) + float32(q2**2) < 1.
RF_HD bool sphere_attempt(Rng &g, uint32_t w[6])
{
    rng_next(g, w[0], w[1]);
    rng_next(g, w[2], w[3]);
    rng_next(g, w[4], w[5]);
    const float ta = approx_pm1(w[0]), tb = approx_pm1(w[2]), tc = approx_pm1(w[4]);
    const float sq = __builtin_fmaf(ta, ta, __builtin_fmaf(tb, tb, tc * tc));
    bool accept = sq < 1.0f - kAcceptBand;
    if (__builtin_expect(!accept && sq < 1.0f + kAcceptBand, 0))
        accept = sq_len_exact_sequential(w) < 1.0f;
    return accept;
}

RF_HD float sphere_attempt_sq(Rng &g, uint32_t w[6]) // see disc_attempt_sq
{
    rng_next(g, w[0], w[1]);
    rng_next(g, w[2], w[3]);
    rng_next(g, w[4], w[5]);
    const float ta = approx_pm1(w[0]), tb = approx_pm1(w[2]), tc = approx_pm1(w[4]);
    float sq = __builtin_fmaf(ta, ta, __builtin_fmaf(tb, tb, tc * tc));
    if (__builtin_expect(__builtin_fabsf(sq - 1.0f) < kAcceptBand, 0))
        sq = sq_len_exact_sequential(w);
    return sq;
}

RF_HD void sphere_finish(const uint32_t w[6], float &q0, float &q1, float &q2)
{
    q0 = exact_pm1(w[0], w[1]);
    q1 = exact_pm1(w[2], w[3]);
    q2 = exact_pm1(w[4], w[5]);
}

RF_HD void sphere_sample(Rng &g, float &q0, float &q1, float &q2)
{
    uint32_t w[6];
    for (;;) {
        if (sphere_attempt(g, w))
            break;
#if RF_LOOP_UNROLL >= 2
        if (sphere_attempt(g, w))
            break;
#endif
#if RF_LOOP_UNROLL >= 3
        if (sphere_attempt(g, w))
            break;
#endif
    }
    sphere_finish(w, q0, q1, q2);
}

// physics.py:183-193: sky colour of direction d times attenuation.
// Literal form: T = 0.5*(ud.y + 1.0) in f64 (0.5*x is exact so T == fma(ud.y, 0.5, 0.5)),
// white = f32(1 - T), blue_k = f32(f64(k) * T), channel = white + blue_k.
// Correctly rounded sqrt and reciprocal for arguments in the normal range
// [2^-100, 2^100] without the denormal scaling / class handling of the generic
// expansions: v_sqrt_f32 (1 ulp) fixed up by testing both neighbours with an exact fma
// residual; v_rcp_f32 (1 ulp) refined by two fma Newton steps.  Equality with the IEEE
// results is checked for EVERY float in the range on the GPU (tests/gpucheck).  The host
// build (tests/hostsim) uses the plain operators: same values by definition.
RF_HD bool in_fast_range(float x)
{
    uint32_t b;
    __builtin_memcpy(&b, &x, 4);
    return (b - 0x0D800000u) < (0x71800000u - 0x0D800000u); // 2^-100 <= x < 2^100, positive
}

RF_HD float sqrt_rn_fast(float x) // requires in_fast_range(x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float s = __builtin_amdgcn_sqrtf(x);
    uint32_t b;
    __builtin_memcpy(&b, &s, 4);
    uint32_t bd = b - 1u, bu = b + 1u;
    float s_dn, s_up;
    __builtin_memcpy(&s_dn, &bd, 4);
    __builtin_memcpy(&s_up, &bu, 4);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    s = (r_dn <= 0.0f) ? s_dn : s;
    s = (r_up > 0.0f) ? s_up : s;
    return s;
#else
    return __builtin_sqrtf(x);
#endif
}

RF_HD float rcp_rn_fast(float x) // requires in_fast_range(x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float y1 = __builtin_fmaf(__builtin_fmaf(-x, y0, 1.0f), y0, y0);
    return __builtin_fmaf(__builtin_fmaf(-x, y1, 1.0f), y1, y1);
#else
    return 1.0f / x;
#endif
}

RF_HD float unit_dir_y(float d0, float d1, float d2)
{
    float sq = sq_len(d0, d1, d2);
    // float32(math.sqrt(sq)): an f64 sqrt rounded to f32 equals the correctly rounded
    // f32 sqrt (53 >= 2*24+2); inv = float32(1) / len with IEEE division.
    if (__builtin_expect(in_fast_range(sq), 1)) {
        const float len = sqrt_rn_fast(sq); // in [2^-50, 2^50]
        return d1 * rcp_rn_fast(len);
    }
    // generic expansions (need -fhip-fp32-correctly-rounded-divide-sqrt, HIP's default;
    // __fsqrt_rn would be the *native* 1-ulp sqrt)
    float len = __builtin_sqrtf(sq);
    float inv = 1.0f / len;
    return d1 * inv;
}

RF_HD double sky_t(float ud1) { return __builtin_fma((double)ud1, 0.5, 0.5); }

RF_HD float sky_channel_literal(double t, float white, float k)
{
    float blue = (float)((double)k * t);
    return add2(white, blue);
}

// f32 form of the same values.  With ud = ud.y (|ud| <= 1 + few ulp):
//   T exact = 0.5 + 0.5*ud has <= 26 significant bits, so the f64 operations of the
//   literal form are exact (for |ud| < 2^-29 they round, but by < 2^-53, far below the
//   f32 half-ulp of the results), k*T_exact has <= 50 bits (exact in f64), and therefore
//   f32(1 - T) == fma(-0.5, ud, 0.5) and f32(k*T) == fma(k/2, ud, k/2): one rounding of the
//   same exact value.  k/2 is exact for every f32 k.  Checked for every f32 ud in
//   [-1-8ulp, 1+8ulp] by tests/test_hostsim.py::test_sky_f32_equals_literal.
constexpr float kSkyHalf[3] = {0.25f, 0.5f * 0.7f, 0.5f}; // k/2 for k = 0.5f, 0.7f, 1f

RF_HD float sky_white(float ud1) { return __builtin_fmaf(-0.5f, ud1, 0.5f); }
RF_HD float sky_blue(float ud1, float k_half) { return __builtin_fmaf(k_half, ud1, k_half); }

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
    double T = sky_t(unit_dir_y(dx, dy, dz));
    float white = (float)(1.0 - T);
    Colour c;
    c.r = sky_channel_literal(T, white, 0.5f) * ar;
    c.g = sky_channel_literal(T, white, 0.7f) * ag;
    c.b = sky_channel_literal(T, white, 1.0f) * ab;
    return c;
}

// s = float32((x + xi) / w): sum and quotient in f64 (int + f32 -> f64).
RF_HD float pixel_coord_literal(int x, float xi, int w)
{
    return (float)(((double)x + (double)xi) / (double)w);
}

// The same value without the IEEE division expansion (~14 f64 instructions with v_rcp_f64):
// w is a frame size, rw = RN64(1 / w) comes from the host.  q0 = RN(a * rw) is a faithful
// quotient, the residual a - q0 * w is exact in one fma, and q = RN(q0 + rem * rw) is the
// correctly rounded a / w (Markstein).  Checked against '/' for every frame size up to 4096
// and > 10^9 numerators by tests/test_hostsim.py.
RF_HD float pixel_coord_div(int x, float xi, double w, double rw)
{
    const double a = (double)x + (double)xi;
    const double q0 = a * rw;
    const double rem = __builtin_fma(-q0, w, a);
    return (float)__builtin_fma(rem, rw, q0);
}

// For w a power of two the same value needs no f64: the f64 sum is either exact or
// the addend is < 2^-28 ulp-wise irrelevant (profiles/HISTORY.md section 4.1), so RN32(RN64(x+xi)) ==
// RN32(x+xi) == the f32 add, and the division is an exact scaling.
RF_HD float pixel_coord_pow2(int x, float xi, float inv_w)
{
    return ((float)x + xi) * inv_w;
}
// same with xi64 = 2^64 * xi: fma(xi64, 2^-64, x) == RN(x + xi)
RF_HD float pixel_coord_pow2_64(float xf, float xi64, float inv_w)
{
    return __builtin_fmaf(xi64, kTwoM64, xf) * inv_w;
}

// ---------------------------------------------------------------------------
// one pixel of FastRenderer._device_render (render.py:210-246): spp samples
// accumulated in f32.  Shared verbatim by the gfx950 kernel (rf_render.h) and the
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
    float rden; // RN(1 / den), for div_by_const
    bool fast_div; // den in [2^-40, 2^40]: no intermediate of div_by_const can underflow
};

// Correctly rounded a / d for a divisor known in advance (Markstein): with r = RN(1/d),
// q0 = RN(a*r) is within 1 ulp of a/d, the residual a - q0*d is exact in one fma, and
// q = RN(q0 + rem*r) is RN(a/d) (no overflow/underflow in this path: a = p + half lies in
// {0} U [2^-24*half, 2*half]).  3 VALU ops instead of the ~12 of an IEEE division.
// Checked against '/' on 10^9 operand pairs by tests/test_hostsim.py.
RF_HD float div_by_const(float a, float d, float rd)
{
    const float q0 = a * rd;
    const float rem = __builtin_fmaf(-q0, d, a);
    return __builtin_fmaf(rem, rd, q0);
}

RF_HD PixelEnv make_pixel_env(const float *cd, const float *rc)
{
    PixelEnv e;
    e.dyn = CamDyn{cd[0], cd[1], cd[2], cd[3], cd[4], cd[5], cd[6], cd[7], cd[8]};
    e.rect = Rect{rc[0], rc[1]};
    e.tt = e.rect.z / e.dyn.llz;
    e.tmiss = (e.tt < 0.001f || e.tt > 1000000.0f);
    e.den = e.rect.half - (-e.rect.half);
    e.rden = 1.0f / e.den;
    e.fast_div = e.den >= 9.094947017729282e-13f && e.den <= 1099511627776.0f;
    return e;
}

// camera.get_ray + rectangle.fast_hit (+ uv / checker colour on a hit) of one sample for
// the canonical camera frame: everything before the scatter's random_in_unit_sphere.
struct AxisPre {
    bool hit, red;
    float dx, dy, dz; // primary ray direction (the sky direction of a miss)
};
// the ray and where it meets the target's plane (rectangle.py:128-133), before the hit test
struct AxisRay {
    float dx, dy, dz;
    float px, py;
    float reach; // max(|p.x|, |p.y|): hit == !tmiss && !(reach > half)
};

template <int LENS = -1>
RF_HD AxisRay sample_axis_point(float p0, float p1, const PixelEnv &e, const CamStatic &cs, float s, float t)
{
    float ox = lens_offset<LENS>(p0, cs);
    float oy = lens_offset<LENS>(p1, cs);
    AxisRay r;
    r.dx = (e.dyn.llx + e.dyn.hx * s) - ox;
    r.dy = (e.dyn.lly + e.dyn.vy * t) - oy;
    r.dz = e.dyn.llz;
    r.px = ox + r.dx * e.tt;
    r.py = oy + r.dy * e.tt;
    // rectangle.py:135: miss if p.x < -half or p.x > half or p.y < -half or p.y > half.
    // == !(max(|p.x|, |p.y|) > half) including the NaN cases (maxNum drops a NaN operand,
    // exactly as the four comparisons ignore it).
    r.reach = __builtin_fmaxf(__builtin_fabsf(r.px), __builtin_fabsf(r.py));
    return r;
}

// rectangle.uv + colour_checkerboard of a hit at (px, py): true = red
RF_HD bool sample_axis_red(float px, float py, const PixelEnv &e, const CheckerTable &tab)
{
    const float half = e.rect.half;
    float u, v;
    if (e.fast_div) { // per-environment condition: uniform across the block
        u = div_by_const(px + half, e.den, e.rden);
        v = div_by_const(py + half, e.den, e.rden);
    } else {
        u = (px + half) / e.den;
        v = (py + half) / e.den;
    }
    return checker_red(u, v, tab);
}

template <int LENS = -1>
RF_HD AxisPre sample_axis_ray(float p0, float p1, const PixelEnv &e, const CamStatic &cs, float s, float t,
                              const CheckerTable &tab)
{
    const AxisRay ray = sample_axis_point<LENS>(p0, p1, e, cs, s, t);
    AxisPre r;
    r.dx = ray.dx;
    r.dy = ray.dy;
    r.dz = ray.dz;
    r.hit = !e.tmiss && !(ray.reach > e.rect.half);
    r.red = false;
    if (r.hit)
        r.red = sample_axis_red(ray.px, ray.py, e, tab);
    return r;
}

// physics.fast_find_colour's tail (physics.py:183-193) for a hit (scattered direction
// N + q = (q0, q1, 1 + q2), attenuation red or green) or a miss (primary direction).
// (hit / red given apart from the ray: render_kernel_coop2 keeps them as lane masks in scalar registers)
RF_HD Colour sample_axis_shade(bool hit, bool red, float rdx, float rdy, float rdz, float q0, float q1, float q2)
{
    const float dx = hit ? q0 : rdx, dy = hit ? q1 : rdy, dz = hit ? 1.0f + q2 : rdz;
    const float ud1 = unit_dir_y(dx, dy, dz);
    const float white = sky_white(ud1);
    Colour c;
    if (hit) {
        // attenuation (1,0,0) or (0,1,0): the other channels contribute +0
        float ch = add2_not_negzero(white, sky_blue(ud1, red ? kSkyHalf[0] : kSkyHalf[1]));
        c.r = red ? ch : 0.0f;
        c.g = red ? 0.0f : ch;
        c.b = 0.0f;
    } else {
        c.r = add2_not_negzero(white, sky_blue(ud1, kSkyHalf[0]));
        c.g = add2_not_negzero(white, sky_blue(ud1, kSkyHalf[1]));
        c.b = add2_not_negzero(white, sky_blue(ud1, kSkyHalf[2]));
    }
    return c;
}
RF_HD Colour sample_axis_shade(const AxisPre &r, float q0, float q1, float q2)
{
    return sample_axis_shade(r.hit, r.red, r.dx, r.dy, r.dz, q0, q1, q2);
}

RF_HD AxisPre sample_axis_pre(Rng &g, const PixelEnv &e, const CamStatic &cs, float s, float t,
                              const CheckerTable &tab)
{
    float p0, p1;
    disc_sample(g, p0, p1);
    return sample_axis_ray(p0, p1, e, cs, s, t, tab);
}

RF_HD Colour sample_axis(Rng &g, const PixelEnv &e, const CamStatic &cs, float s, float t,
                         const CheckerTable &tab)
{
    const AxisPre r = sample_axis_pre(g, e, cs, s, t, tab);
    float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f;
    if (r.hit)
        sphere_sample(g, q0, q1, q2);
    return sample_axis_shade(r, q0, q1, q2);
}

// what the jittered coordinates need to know about the frame, computed once on the host (kernel arguments: scalar
// registers, no per-lane conversions)
struct FrameConst {
    double w64, h64;    // (double)w, (double)h
    double rw64, rh64;  // RN64(1 / w), RN64(1 / h): pixel_coord_div
    float inv_w, inv_h; // 1 / w, 1 / h: exact for powers of two (pixel_coord_pow2)
};

RF_HD FrameConst frame_const(int h, int w)
{
    FrameConst f;
    f.w64 = (double)w;
    f.h64 = (double)h;
    f.rw64 = 1.0 / f.w64;
    f.rh64 = 1.0 / f.h64;
    f.inv_w = 1.0f / (float)w;
    f.inv_h = 1.0f / (float)h;
    return f;
}

// jittered pixel coordinates of one sample (render.py:229-234), two draws
template <bool POW2>
RF_HD void sample_coords(Rng &g, int x, int y, float xf, float yf, const FrameConst &f, float &s, float &t)
{
    uint32_t xh, xl, yh, yl;
    rng_next(g, xh, xl);
    rng_next(g, yh, yl);
    const float xi = unit_f32_scaled64(xh, xl), yi = unit_f32_scaled64(yh, yl); // 2^64 * uniform
    if (POW2) {
        s = pixel_coord_pow2_64(xf, xi, f.inv_w);
        t = pixel_coord_pow2_64(yf, yi, f.inv_h);
        return;
    }
    // other frame sizes: the float64 quotient, seven instructions per coordinate.  (A double-float form in float32 --
    // two_sum, quotient, exact residual, correction, with the float64 form only next to rounding boundaries and for
    // jitters too small for x + xi to be exact -- is bit-identical too (10^9 quotients on the CPU, the GPU suite) and 2-4 %
    // SLOWER end to end: thirteen float32 operations and two compares do not hide behind the generator's integer work:
    // profiles/r05_ab.txt section 7.)
    s = pixel_coord_div(x, xi * kTwoM64, f.w64, f.rw64);
    t = pixel_coord_div(y, yi * kTwoM64, f.h64, f.rh64);
}

template <bool AXIS, bool POW2>
RF_HD void render_pixel(Rng &g, int x, int y, int spp, const FrameConst &f, const PixelEnv &e, const CamStatic &cs,
                        const CheckerTable &tab, float &cr,
                        float &cg, float &cb)
{
    cr = cg = cb = 0.0f;
    const float xf = (float)x, yf = (float)y;
    for (int k = 0; k < spp; ++k) {
        float s, t;
        sample_coords<POW2>(g, x, y, xf, yf, f, s, t);
        Colour c = AXIS ? sample_axis(g, e, cs, s, t, tab)
                        : sample_general(g, e.dyn, cs, e.rect, s, t, tab);
        cr = add2(cr, c.r);
        cg = add2(cg, c.g);
        cb = add2(cb, c.b);
    }
}

} // namespace rf
