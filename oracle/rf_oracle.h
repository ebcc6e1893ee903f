/*
 * rf_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the reinfocus render-and-measure hot path
 * (reference files cited per function in rf_oracle.c).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Arithmetic model: numba typing == numpy-1.26 scalar promotion, IEEE-754,
 * no FMA contraction (what NUMBA_ENABLE_CUDASIM=1 computes).  Build with
 *   gcc -O2 -ffp-contract=off -fno-fast-math
 *
 * Parity status: PINNED.  The reference cannot be imported in this image (numba / cv2 /
 * gymnasium absent), but its repository holds real outputs of itself: the cells of
 * examples/environment.ipynb (numba on CUDA + OpenCV 4.9).  Replaying that call sequence on
 * this oracle reproduces every printed digit (tests/test_reference_known_answers.py::
 * test_oracle_reproduces_reference_notebook), which exercises seeding, four renders, the
 * OpenCV chain and var().  In addition: (i) every known answer the reference's own tests
 * hold for the path, (ii) an independent numpy-1.26.4 restatement
 * (oracle/np126_restatement.py -> tests/golden), (iii) scipy.ndimage and numpy for the
 * vision stages.  Third-party arithmetic (numba xoroshiro128+, OpenCV gray / median /
 * Laplacian) is restated from the published algorithms and confirmed only through (the
 * strong) notebook check above.
 */
#ifndef RF_ORACLE_H
#define RF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t s0, s1; } orc_state;

/* The per-env-independent half of FastGpuCameras (camera.py:39-52). */
typedef struct {
    float origin[3];
    float u[3];
    float v[3];
    double lens_radius; /* numpy.float64: numpy.divide(aperture, 2.0) camera.py:124 */
} orc_cam_static;

/* ---- RNG: numba.cuda.random (numba ~=0.59), called at graphics/random.py:18,33 ---- */
void orc_init_state(orc_state *st, uint64_t seed);
uint64_t orc_next(orc_state *st);
void orc_jump(orc_state *st);
void orc_seed_states(orc_state *states, uint64_t n, uint64_t seed, uint64_t subsequence_start);
float orc_uniform_float(orc_state *st);

/* ---- device functions ---- */
void orc_random_in_unit_disc(orc_state *st, float p[2]);
void orc_random_in_unit_sphere(orc_state *st, float p[3]);
void orc_get_ray(const float dyn[9], const orc_cam_static *cs, float s, float t,
                 orc_state *st, float origin[3], float direction[3]);
void orc_uv(const float point[2], float x_min, float x_max, float y_min, float y_max,
            float uv[2]);
int orc_fast_hit(const float rect[2], const float origin[3], const float direction[3],
                 float t_min, float t_max, float rec[13]);
void orc_colour_checkerboard(const float uf[2], const float uv[2], float colour[3]);
void orc_scatter(const float rec[13], orc_state *st, float origin[3], float direction[3],
                 float attenuation[3]);
void orc_fast_find_colour(const float rect[2], const float origin[3],
                          const float direction[3], orc_state *st, float colour[3]);

/* ---- kernel: FastRenderer._device_render render.py:190-246 ---- */
void orc_render(uint8_t *frames, int n, int h, int w, int spp, const float *cam_dyn,
                const float *rect, const orc_cam_static *cs, orc_state *states,
                int n_threads);

/* ---- general renderer: spheres + rectangles, 50-bounce find_colour ---- */
int orc_sphere_hit(const float *sphere, const float origin[3], const float direction[3],
                   float t_min, float t_max, float rec[13]);
void orc_sphere_uv(const float point[3], float uv[2]);
int orc_rectangle_hit(const float *rect, const float origin[3], const float direction[3],
                      float t_min, float t_max, float rec[13]);
int orc_world_hit(const float *params, const int32_t *types, int n_shapes, int width,
                  const float origin[3], const float direction[3], float t_min, float t_max,
                  float rec[13]);
void orc_find_colour(const float *params, const int32_t *types, int n_shapes, int width,
                     const float origin[3], const float direction[3], orc_state *st, float colour[3]);
void orc_render_general(uint8_t *frames, int n, int h, int w, int spp, const double *cameras /*[n][19]*/,
                        const float *params /*[n][most][width]*/, const int32_t *types /*[n][most]*/,
                        const int32_t *sizes /*[n]*/, int most, int width, orc_state *states,
                        int n_threads);

/* ---- vision.py:11-39 (OpenCV 4.9 semantics restated) ---- */
void orc_gray(const uint8_t *rgb, int h, int w, int gray_mode, uint8_t *gray);
void orc_median3(const uint8_t *src, int h, int w, uint8_t *dst);
void orc_laplacian_u8(const uint8_t *src, int h, int w, uint8_t *dst);
double orc_var_u8(const uint8_t *src, long n);
double orc_focus_value(const uint8_t *rgb, int h, int w, int gray_mode);
void orc_focus_values(const uint8_t *frames, int n, int h, int w, int gray_mode,
                      double *out, int n_threads);

/* graphics/vector.py device helpers (see rf_oracle.c for the op codes). */
void orc_vector_op(int op, const float a[3], const float b[3], const float c[3], float s, float out[3]);

#ifdef __cplusplus
}
#endif
#endif
