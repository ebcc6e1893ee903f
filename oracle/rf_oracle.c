/*
 * rf_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of jeffwhunter/reinfocus' render-and-measure hot path.
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).  See rf_oracle.h for the arithmetic model and the
 * parity status.  Every float32(...) in the reference is an explicit (float)
 * cast here, every numpy-1.26 promotion to float64 an explicit (double).
 *
 * Must be compiled with -ffp-contract=off -fno-fast-math (see Makefile).
 */
#include "rf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* RNG -- numba.cuda.random (third-party, numba ~=0.59.0, pyproject.toml:31) */
/* reference call sites: reinfocus/graphics/random.py:18 and :33            */
/* ------------------------------------------------------------------------ */

/* math.pi: the float64 nearest to pi */
#define ORC_PI 3.14159265358979323846

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

/* numba/cuda/random.py init_xoroshiro128p_state: SplitMix64 of the seed, s0 = s1 = z. */
void orc_init_state(orc_state *st, uint64_t seed)
{
    uint64_t z = seed + UINT64_C(0x9E3779B97F4A7C15);
    z = (z ^ (z >> 30)) * UINT64_C(0xBF58476D1CE4E5B9);
    z = (z ^ (z >> 27)) * UINT64_C(0x94D049BB133111EB);
    z = z ^ (z >> 31);
    st->s0 = z;
    st->s1 = z;
}

/* numba/cuda/random.py xoroshiro128p_next: xoroshiro128+ with constants 55/14/36. */
uint64_t orc_next(orc_state *st)
{
    uint64_t s0 = st->s0, s1 = st->s1;
    uint64_t result = s0 + s1;
    s1 ^= s0;
    st->s0 = rotl64(s0, 55) ^ s1 ^ (s1 << 14);
    st->s1 = rotl64(s1, 36);
    return result;
}

/* numba/cuda/random.py xoroshiro128p_jump: advance by 2**64 steps. */
void orc_jump(orc_state *st)
{
    static const uint64_t jump[2] = {UINT64_C(0xbeac0467eba5facb),
                                     UINT64_C(0xd86b048b86aa9922)};
    uint64_t s0 = 0, s1 = 0;
    for (int i = 0; i < 2; ++i)
        for (int b = 0; b < 64; ++b) {
            if (jump[i] & (UINT64_C(1) << b)) {
                s0 ^= st->s0;
                s1 ^= st->s1;
            }
            orc_next(st);
        }
    st->s0 = s0;
    st->s1 = s1;
}

/* numba/cuda/random.py init_xoroshiro128p_states_cpu, reached from
 * graphics/random.py:8-18 make_random_states(n, seed): strictly sequential,
 * state[i] = jump(state[i-1]). */
void orc_seed_states(orc_state *states, uint64_t n, uint64_t seed, uint64_t subsequence_start)
{
    if (n < 1)
        return;
    orc_init_state(&states[0], seed);
    for (uint64_t k = 0; k < subsequence_start; ++k)
        orc_jump(&states[0]);
    for (uint64_t i = 1; i < n; ++i) {
        states[i] = states[i - 1];
        orc_jump(&states[i]);
    }
}

/* graphics/random.py:21-33 uniform_float -> numba xoroshiro128p_uniform_float32:
 * float32( float64(x >> 11) * (1 / 2**53) ).  1.0f is reachable. */
float orc_uniform_float(orc_state *st)
{
    uint64_t x = orc_next(st);
    return (float)((double)(x >> 11) * (1.0 / 9007199254740992.0));
}

/* ------------------------------------------------------------------------ */
/* device functions                                                         */
/* ------------------------------------------------------------------------ */

/* graphics/vector.py:116-133 d_add_v3f: x = y = z = float32(0); x += each summand. */
static inline float add2(float a, float b) { return (0.0f + a) + b; }
static inline float add3(float a, float b, float c) { return ((0.0f + a) + b) + c; }

/* graphics/camera.py:229-252 random_in_unit_disc.
 * p = (xi_a, xi_b) * 2f - (1, 1) in f32; accept when p0*p0 + p1*p1 < 1.0. */
void orc_random_in_unit_disc(orc_state *st, float p[2])
{
    for (;;) {
        float a = orc_uniform_float(st);
        float b = orc_uniform_float(st);
        p[0] = a * 2.0f - 1.0f;
        p[1] = b * 2.0f - 1.0f;
        float dot = p[0] * p[0] + p[1] * p[1]; /* vector.py:241-251 d_dot_v2f, f32 */
        if ((double)dot < 1.0)
            return;
    }
}

/* graphics/vector.py:300-314 d_squared_length_v3f:
 * float32(v0**2) + float32(v1**2) + float32(v2**2); v**2 is exact in f64. */
static inline float squared_length(const float v[3])
{
    float a = (float)((double)v[0] * (double)v[0]);
    float b = (float)((double)v[1] * (double)v[1]);
    float c = (float)((double)v[2] * (double)v[2]);
    return (a + b) + c;
}

/* graphics/physics.py:20-44 random_in_unit_sphere. */
void orc_random_in_unit_sphere(orc_state *st, float p[3])
{
    for (;;) {
        float a = orc_uniform_float(st);
        float b = orc_uniform_float(st);
        float c = orc_uniform_float(st);
        p[0] = a * 2.0f - 1.0f;
        p[1] = b * 2.0f - 1.0f;
        p[2] = c * 2.0f - 1.0f;
        if ((double)squared_length(p) < 1.0)
            return;
    }
}

/* graphics/camera.py:307-350 get_ray with cam = from_fast_cameras (:284-304).
 * dyn = [lower_left, horizontal, vertical] (row-major 3x3 f32).
 * rd = disc * lens_radius is float64 (lens_radius is numpy.float64 and
 * vector.py:179-190 d_smul_v2f has no cast); d_smul_v3f (vector.py:208-223)
 * rounds each product to f32 once. */
void orc_get_ray(const float dyn[9], const orc_cam_static *cs, float s, float t,
                 orc_state *st, float origin[3], float direction[3])
{
    float p[2];
    orc_random_in_unit_disc(st, p);
    double rd0 = (double)p[0] * cs->lens_radius;
    double rd1 = (double)p[1] * cs->lens_radius;
    for (int k = 0; k < 3; ++k) {
        float ur = (float)((double)cs->u[k] * rd0);
        float vr = (float)((double)cs->v[k] * rd1);
        origin[k] = add3(cs->origin[k], ur, vr);
    }
    for (int k = 0; k < 3; ++k) {
        float hs = dyn[3 + k] * s; /* f32 * f32 */
        float vt = dyn[6 + k] * t;
        direction[k] = add3(dyn[k], hs, vt) - origin[k];
    }
}

/* graphics/rectangle.py:151-170 uv. */
void orc_uv(const float point[2], float x_min, float x_max, float y_min, float y_max,
            float uv[2])
{
    uv[0] = (point[0] - x_min) / (x_max - x_min);
    uv[1] = (point[1] - y_min) / (y_max - y_min);
}

/* graphics/rectangle.py:102-148 fast_hit; ray.py:29-40 point_at_parameter.
 * rec = flattened hit record (tests/graphics/numba_test_utils.py flatten_hit_record
 * order): p[3], n[3], t, uv[2], uf[2], m ; rec[12] unused pad.  Returns did_hit.
 * Python semantics kept: "t < t_min or t > t_max" (a NaN t is a hit). */
int orc_fast_hit(const float rect[2], const float origin[3], const float direction[3],
                 float t_min, float t_max, float rec[13])
{
    float radius = rect[0];
    float z_pos = rect[1];
    memset(rec, 0, 13 * sizeof(float));

    float t = (z_pos - origin[2]) / direction[2];
    if (t < t_min || t > t_max)
        return 0;

    float p[3];
    for (int k = 0; k < 3; ++k)
        p[k] = add2(origin[k], direction[k] * t);

    if (p[0] < -radius || p[0] > radius || p[1] < -radius || p[1] > radius)
        return 0;

    rec[0] = p[0];
    rec[1] = p[1];
    rec[2] = p[2];
    rec[3] = 0.0f;
    rec[4] = 0.0f;
    rec[5] = 1.0f;
    rec[6] = t;
    orc_uv(p, -radius, radius, -radius, radius, &rec[7]);
    rec[9] = 32.0f;
    rec[10] = 32.0f;
    rec[11] = 1.0f; /* shape.RECTANGLE */
    return 1;
}

/* graphics/physics.py:47-64 colour_checkerboard:
 * si = uf * math.pi * uv evaluated left to right in float64; red when
 * sin(si0) * sin(si1) > 0 else green. */
void orc_colour_checkerboard(const float uf[2], const float uv[2], float colour[3])
{
    double si0 = ((double)uf[0] * ORC_PI) * (double)uv[0];
    double si1 = ((double)uf[1] * ORC_PI) * (double)uv[1];
    int red = sin(si0) * sin(si1) > 0.0;
    colour[0] = red ? 1.0f : 0.0f;
    colour[1] = red ? 0.0f : 1.0f;
    colour[2] = 0.0f;
}

/* graphics/physics.py:67-92 scatter. */
void orc_scatter(const float rec[13], orc_state *st, float origin[3], float direction[3],
                 float attenuation[3])
{
    float q[3];
    orc_random_in_unit_sphere(st, q);
    for (int k = 0; k < 3; ++k) {
        origin[k] = rec[k];
        direction[k] = add2(rec[3 + k], q[k]);
    }
    orc_colour_checkerboard(&rec[9], &rec[7], attenuation);
}

/* graphics/physics.py:148-193 fast_find_colour; vector.py:329-364 d_length_v3f /
 * d_norm_v3f.  len = float32(math.sqrt(sq)) (f64 sqrt of an f32, rounded to f32);
 * inv = float32(1) / len; t = 0.5 * (ud.y + 1.0) is float64. */
void orc_fast_find_colour(const float rect[2], const float origin[3],
                          const float direction[3], orc_state *st, float colour[3])
{
    float att[3] = {1.0f, 1.0f, 1.0f};
    float dir[3] = {direction[0], direction[1], direction[2]};
    float rec[13];

    if (orc_fast_hit(rect, origin, direction, 0.001f, 1000000.0f, rec)) {
        float o2[3], a2[3];
        orc_scatter(rec, st, o2, dir, a2);
        for (int k = 0; k < 3; ++k)
            att[k] = att[k] * a2[k];
    }

    float len = (float)sqrt((double)squared_length(dir));
    float inv = 1.0f / len;
    float ud1 = dir[1] * inv;
    double t = 0.5 * ((double)ud1 + 1.0);
    static const float sky[3] = {0.5f, 0.7f, 1.0f};
    for (int k = 0; k < 3; ++k) {
        float white = (float)((double)1.0f * (1.0 - t));
        float blue = (float)((double)sky[k] * t);
        colour[k] = add2(white, blue) * att[k];
    }
}

/* ------------------------------------------------------------------------ */
/* kernel: graphics/render.py:190-246 FastRenderer._device_render           */
/* One "thread" per pixel (e, y, x); pixel_index = e*h*w + y*w + x selects   */
/* the RNG state (render.py:217).  s/t: float32((x + xi) / w) with the sum   */
/* and quotient in float64 (int + f32 -> f64).  Draw order: xi_x, xi_y,      */
/* disc pairs, then sphere triples on a hit.  Final store truncates to u8.   */
/* ------------------------------------------------------------------------ */
void orc_render(uint8_t *frames, int n, int h, int w, int spp, const float *cam_dyn,
                const float *rect, const orc_cam_static *cs, orc_state *states,
                int n_threads)
{
    const float scale = (float)(255.0 / (double)spp);
    const long rows = (long)n * h;
    if (n_threads < 1)
        n_threads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads)
    for (long row = 0; row < rows; ++row) {
        const int e = (int)(row / h);
        const int y = (int)(row % h);
        for (int x = 0; x < w; ++x) {
            const long pix = ((long)e * h + y) * w + x;
            orc_state st = states[pix];
            float colour[3] = {0.0f, 0.0f, 0.0f};
            for (int k = 0; k < spp; ++k) {
                float xi = orc_uniform_float(&st);
                float s = (float)(((double)x + (double)xi) / (double)w);
                float yi = orc_uniform_float(&st);
                float t = (float)(((double)y + (double)yi) / (double)h);
                float ro[3], rd[3], sample[3];
                orc_get_ray(&cam_dyn[9 * e], cs, s, t, &st, ro, rd);
                orc_fast_find_colour(&rect[2 * e], ro, rd, &st, sample);
                for (int c = 0; c < 3; ++c)
                    colour[c] = add2(colour[c], sample[c]);
            }
            for (int c = 0; c < 3; ++c)
                frames[pix * 3 + c] = (uint8_t)(colour[c] * scale);
            states[pix] = st;
        }
    }
}

/* graphics/vector.py device helpers, exported one by one so that the reference's own known
 * answers for them (tests/graphics/vector_test.py, ray_test.py) can be asserted against the very
 * helpers the renderers above are written with.
 * op: 0 d_add_v3f((a, b, c))  1 d_sub_v3f(a, b)  2 d_smul_v3f(a, s)  3 d_dot_v3f(a, b) -> out[0]
 *     4 d_squared_length_v3f(a) -> out[0]  5 d_length_v3f(a) -> out[0]  6 d_norm_v3f(a)
 *     7 ray.point_at_parameter(origin a, direction b, s)  8 d_dot_v2f(a, b) -> out[0] */
void orc_vector_op(int op, const float a[3], const float b[3], const float c[3], float s, float out[3])
{
    out[0] = out[1] = out[2] = 0.0f;
    switch (op) {
    case 0:
        for (int k = 0; k < 3; ++k)
            out[k] = add3(a[k], b[k], c[k]);
        break;
    case 1:
        for (int k = 0; k < 3; ++k)
            out[k] = a[k] - b[k];
        break;
    case 2:
        for (int k = 0; k < 3; ++k)
            out[k] = a[k] * s;
        break;
    case 3:
        out[0] = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
        break;
    case 4:
        out[0] = squared_length(a);
        break;
    case 5:
        out[0] = (float)sqrt((double)squared_length(a));
        break;
    case 6: {
        const float inv = 1.0f / (float)sqrt((double)squared_length(a));
        for (int k = 0; k < 3; ++k)
            out[k] = a[k] * inv;
        break;
    }
    case 7: /* ray.py:29-40: origin + d_smul_v3f(direction, t), summed by d_add_v3f */
        for (int k = 0; k < 3; ++k)
            out[k] = add2(a[k], b[k] * s);
        break;
    default:
        out[0] = a[0] * b[0] + a[1] * b[1];
        break;
    }
}

/* ------------------------------------------------------------------------ */
/* general renderer: render.py:31-119, world.py:126-167, physics.py:95-145,  */
/* sphere.py:40-117, rectangle.py:49-99                                      */
/* ------------------------------------------------------------------------ */

static inline float dot3(const float a[3], const float b[3])
{
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; /* vector.py:254-265, f32 */
}

/* sphere.py:106-117 uv: (atan2(-z, x) + pi) / pi, acos(-y) / pi in float64 (math.* on
 * f32 arguments), cast to f32.  acos outside [-1, 1] is NaN (device semantics). */
void orc_sphere_uv(const float point[3], float uv[2])
{
    uv[0] = (float)((atan2(-(double)point[2], (double)point[0]) + ORC_PI) / ORC_PI);
    uv[1] = (float)(acos(-(double)point[1]) / ORC_PI);
}

/* sphere.py:40-103 hit.  sphere = {x, y, z, r, fx, fy}.  a, b, c, discriminant in f32;
 * sqrtd = math.sqrt -> f64; root = (-b -/+ sqrtd) / a in f64; p = o + f32(d * root);
 * n = (p - centre) * f32(1.0 / r). */
int orc_sphere_hit(const float *sp, const float origin[3], const float direction[3],
                   float t_min, float t_max, float rec[13])
{
    memset(rec, 0, 13 * sizeof(float));
    const float centre[3] = {sp[0], sp[1], sp[2]};
    const float radius = sp[3];
    float oc[3];
    for (int k = 0; k < 3; ++k)
        oc[k] = origin[k] - centre[k];
    const float a = dot3(direction, direction);
    const float b = dot3(oc, direction);
    const float c = dot3(oc, oc) - radius * radius;
    const float disc = b * b - a * c;
    if (disc < 0)
        return 0;
    const double sqrtd = sqrt((double)disc);
    double root = (-(double)b - sqrtd) / (double)a;
    if (root < (double)t_min || (double)t_max < root) {
        root = (-(double)b + sqrtd) / (double)a;
        if (root < (double)t_min || (double)t_max < root)
            return 0;
    }
    float p[3], n[3];
    const float inv_r = (float)(1.0 / (double)radius);
    for (int k = 0; k < 3; ++k)
        p[k] = add2(origin[k], (float)((double)direction[k] * root));
    for (int k = 0; k < 3; ++k)
        n[k] = (p[k] - centre[k]) * inv_r;
    rec[0] = p[0]; rec[1] = p[1]; rec[2] = p[2];
    rec[3] = n[0]; rec[4] = n[1]; rec[5] = n[2];
    rec[6] = (float)root;
    orc_sphere_uv(n, &rec[7]);
    rec[9] = sp[4];
    rec[10] = sp[5];
    rec[11] = 0.0f; /* shape.SPHERE */
    return 1;
}

/* rectangle.py:49-99 hit.  rect = {x_min, x_max, y_min, y_max, z, fx, fy}. */
int orc_rectangle_hit(const float *rp, const float origin[3], const float direction[3],
                      float t_min, float t_max, float rec[13])
{
    memset(rec, 0, 13 * sizeof(float));
    const float t = (rp[4] - origin[2]) / direction[2];
    if (t < t_min || t > t_max)
        return 0;
    float p[3];
    for (int k = 0; k < 3; ++k)
        p[k] = add2(origin[k], direction[k] * t);
    if (p[0] < rp[0] || p[0] > rp[1] || p[1] < rp[2] || p[1] > rp[3])
        return 0;
    rec[0] = p[0]; rec[1] = p[1]; rec[2] = p[2];
    rec[3] = 0.0f; rec[4] = 0.0f; rec[5] = 1.0f;
    rec[6] = t;
    orc_uv(p, rp[0], rp[1], rp[2], rp[3], &rec[7]);
    rec[9] = rp[5];
    rec[10] = rp[6];
    rec[11] = 1.0f; /* shape.RECTANGLE */
    return 1;
}

/* world.py:126-167 hit: closest hit over the env's shapes (t_max shrinks). */
int orc_world_hit(const float *params, const int32_t *types, int n_shapes, int width,
                  const float origin[3], const float direction[3], float t_min, float t_max,
                  float rec[13])
{
    int hit_anything = 0;
    float closest = t_max;
    float tmp[13];
    memset(rec, 0, 13 * sizeof(float));
    for (int i = 0; i < n_shapes; ++i) {
        const float *p = params + (long)i * width;
        const int h = types[i] == 0 ? orc_sphere_hit(p, origin, direction, t_min, closest, tmp)
                                    : orc_rectangle_hit(p, origin, direction, t_min, closest, tmp);
        if (h) {
            hit_anything = 1;
            closest = tmp[6];
            memcpy(rec, tmp, sizeof(tmp));
        }
    }
    return hit_anything;
}

/* physics.py:95-145 find_colour: up to 50 bounces, black afterwards. */
void orc_find_colour(const float *params, const int32_t *types, int n_shapes, int width,
                     const float origin[3], const float direction[3], orc_state *st, float colour[3])
{
    float o[3] = {origin[0], origin[1], origin[2]};
    float d[3] = {direction[0], direction[1], direction[2]};
    float att[3] = {1.0f, 1.0f, 1.0f};
    float rec[13];
    for (int bounce = 0; bounce < 50; ++bounce) {
        if (orc_world_hit(params, types, n_shapes, width, o, d, 0.001f, 1000000.0f, rec)) {
            float a2[3];
            orc_scatter(rec, st, o, d, a2);
            for (int k = 0; k < 3; ++k)
                att[k] = att[k] * a2[k];
        } else {
            const float len = (float)sqrt((double)squared_length(d));
            const float inv = 1.0f / len;
            const float ud1 = d[1] * inv;
            const double t = 0.5 * ((double)ud1 + 1.0);
            static const float sky[3] = {0.5f, 0.7f, 1.0f};
            for (int k = 0; k < 3; ++k) {
                const float white = (float)((double)1.0f * (1.0 - t));
                const float blue = (float)((double)sky[k] * t);
                colour[k] = add2(white, blue) * att[k];
            }
            return;
        }
    }
    colour[0] = colour[1] = colour[2] = 0.0f;
}

/* render.py:31-85 device_render with camera.from_cameras (camera.py:255-281): the camera
 * row is float64[19]; every vector is cast to f32, the lens radius stays f64. */
void orc_render_general(uint8_t *frames, int n, int h, int w, int spp, const double *cameras,
                        const float *params, const int32_t *types, const int32_t *sizes, int most,
                        int width, orc_state *states, int n_threads)
{
    const float scale = (float)(255.0 / (double)spp);
    const long rows = (long)n * h;
    if (n_threads < 1)
        n_threads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads)
    for (long row = 0; row < rows; ++row) {
        const int e = (int)(row / h), y = (int)(row % h);
        const double *cam = cameras + (long)e * 19;
        float dyn[9];
        orc_cam_static cs;
        for (int k = 0; k < 9; ++k)
            dyn[k] = (float)cam[k];
        for (int k = 0; k < 3; ++k) {
            cs.origin[k] = (float)cam[9 + k];
            cs.u[k] = (float)cam[12 + k];
            cs.v[k] = (float)cam[15 + k];
        }
        cs.lens_radius = cam[18];
        for (int x = 0; x < w; ++x) {
            const long pix = ((long)e * h + y) * w + x;
            orc_state st = states[pix];
            float colour[3] = {0.0f, 0.0f, 0.0f};
            for (int k = 0; k < spp; ++k) {
                float xi = orc_uniform_float(&st);
                float s = (float)(((double)x + (double)xi) / (double)w);
                float yi = orc_uniform_float(&st);
                float t = (float)(((double)y + (double)yi) / (double)h);
                float ro[3], rd[3], sample[3];
                orc_get_ray(dyn, &cs, s, t, &st, ro, rd);
                orc_find_colour(params + ((long)e * most) * width, types + (long)e * most, sizes[e], width, ro, rd,
                                &st, sample);
                for (int c = 0; c < 3; ++c)
                    colour[c] = add2(colour[c], sample[c]);
            }
            for (int c = 0; c < 3; ++c)
                frames[pix * 3 + c] = (uint8_t)(colour[c] * scale);
            states[pix] = st;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* vision.py:11-39 -- OpenCV (opencv-python ~=4.9.0.80) restated            */
/* ------------------------------------------------------------------------ */

/* cv2.cvtColor(image, COLOR_RGB2GRAY), 8-bit fixed point (vision.py:24).
 * gray_mode 15: OpenCV >= 4 RGB2Gray<uchar>: (R*9798 + G*19235 + B*3735 + 2^14) >> 15
 * gray_mode 14: older:                       (R*4899 + G*9617  + B*1868 + 2^13) >> 14 */
void orc_gray(const uint8_t *rgb, int h, int w, int gray_mode, uint8_t *gray)
{
    const long n = (long)h * w;
    if (gray_mode == 14) {
        for (long i = 0; i < n; ++i)
            gray[i] = (uint8_t)((rgb[3 * i] * 4899 + rgb[3 * i + 1] * 9617 +
                                 rgb[3 * i + 2] * 1868 + 8192) >> 14);
    } else {
        for (long i = 0; i < n; ++i)
            gray[i] = (uint8_t)((rgb[3 * i] * 9798 + rgb[3 * i + 1] * 19235 +
                                 rgb[3 * i + 2] * 3735 + 16384) >> 15);
    }
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static int cmp_u8(const void *a, const void *b)
{
    return (int)*(const uint8_t *)a - (int)*(const uint8_t *)b;
}

/* cv2.medianBlur(gray, 3) (vision.py:24): 3x3 median, BORDER_REPLICATE. */
void orc_median3(const uint8_t *src, int h, int w, uint8_t *dst)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint8_t v[9];
            int k = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx)
                    v[k++] = src[(long)clampi(y + dy, 0, h - 1) * w + clampi(x + dx, 0, w - 1)];
            qsort(v, 9, 1, cmp_u8);
            dst[(long)y * w + x] = v[4];
        }
}

/* BORDER_REFLECT_101: -1 -> 1, n -> n-2 (single-pixel axis maps to 0). */
static inline int reflect101(int i, int n)
{
    if (n == 1)
        return 0;
    if (i < 0)
        return -i;
    if (i >= n)
        return 2 * n - 2 - i;
    return i;
}

/* cv2.Laplacian(m, cv2.CV_8U) (vision.py:23-25): ksize=1 => kernel
 * [[0,1,0],[1,-4,1],[0,1,0]], scale 1, delta 0, BORDER_DEFAULT (REFLECT_101),
 * saturate_cast<uchar> (negatives clamp to 0). */
void orc_laplacian_u8(const uint8_t *src, int h, int w, uint8_t *dst)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int up = src[(long)reflect101(y - 1, h) * w + x];
            int dn = src[(long)reflect101(y + 1, h) * w + x];
            int lf = src[(long)y * w + reflect101(x - 1, w)];
            int rt = src[(long)y * w + reflect101(x + 1, w)];
            int v = up + dn + lf + rt - 4 * src[(long)y * w + x];
            dst[(long)y * w + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
}

/* numpy's pairwise float64 summation (numpy/core/src/umath/loops_utils.h
 * DOUBLE_pairwise_sum): blocks of 128, 8 accumulators. */
static double pairwise_sum(const double *a, long n)
{
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; ++i)
            res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        long i;
        for (int k = 0; k < 8; ++k)
            r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k)
                r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i)
            res += a[i];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

/* ndarray.var() of a uint8 image (vision.py:25): numpy/core/_methods.py _var:
 * mean = sum(arr, dtype=f8) / N; x = arr - mean; var = sum(x*x) / N. */
double orc_var_u8(const uint8_t *src, long n)
{
    double total = 0.0; /* exact: integer sum < 2^53 */
    for (long i = 0; i < n; ++i)
        total += (double)src[i];
    double mean = total / (double)n;
    double *x = (double *)malloc((size_t)n * sizeof(double));
    for (long i = 0; i < n; ++i) {
        double d = (double)src[i] - mean;
        x[i] = d * d;
    }
    /* the reduction's inner loop runs over iterator chunks of at most 8192 elements
     * (NPY_BUFSIZE), each summed pairwise and accumulated left to right -- checked
     * empirically against numpy 1.26.4 and 2.2.6 for sizes up to 1024^2 */
    double acc = 0.0;
    for (long i = 0; i < n; i += 8192)
        acc += pairwise_sum(x + i, (n - i) < 8192 ? (n - i) : 8192);
    double ret = acc / (double)n;
    free(x);
    return ret;
}

/* vision.py:11-25 focus_value. */
double orc_focus_value(const uint8_t *rgb, int h, int w, int gray_mode)
{
    long n = (long)h * w;
    uint8_t *a = (uint8_t *)malloc((size_t)n);
    uint8_t *b = (uint8_t *)malloc((size_t)n);
    orc_gray(rgb, h, w, gray_mode, a);
    orc_median3(a, h, w, b);
    orc_laplacian_u8(b, h, w, a);
    double v = orc_var_u8(a, n);
    free(a);
    free(b);
    return v;
}

/* vision.py:28-39 focus_values: one focus_value per image. */
void orc_focus_values(const uint8_t *frames, int n, int h, int w, int gray_mode,
                      double *out, int n_threads)
{
    if (n_threads < 1)
        n_threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int e = 0; e < n; ++e)
        out[e] = orc_focus_value(frames + (long)e * h * w * 3, h, w, gray_mode);
}
