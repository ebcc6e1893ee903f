// rf_general_one.h -- the general renderer for worlds of ONE shape per environment -- one rectangle
// in every environment, or one sphere in every environment --, on the fast path's cooperative organisation:
//
//   render_general_one_kernel<POW2, SPHERE>  every pixel: render_kernel_coop2's structure (three pixels per thread, masks
//                                       in scalar registers, in-wave / block-cooperative rejection tails: rf_coop2.h)
//                                       with the general renderer's arithmetic -- per-environment camera with float64
//                                       lens products, any z-aligned rectangle or any sphere, any checker frequencies
//   render_general_fixup_kernel<POW2>   the literal per-pixel code (rf_general.h) for the pixels the first kernel left
//
// Why this class of scenes.  With a single shape in the world a scattered ray cannot hit anything: off a rectangle it
// starts in the rectangle's plane and leaves it (direction (q0, q1, 1 + q2), |q| < 1), off a sphere it starts on the
// surface and points outwards (n + q) -- so find_colour (physics.py:95-145) is one hit test, one random_in_unit_sphere,
// one more hit test that misses, and the sky: the shape of the fast path's sample, and none of the per-set state
// (origin, direction, hit record) that makes packing unprofitable for general worlds (profiles/r04_ab.txt sections 2
// and 5: with bounces a pixel set costs ~30 registers, here 3: what the second test needs -- the plane test's numerator,
// or the hit point).  Worlds with several shapes, or of different kinds in different environments, take the literal kernel.
//
// Exactness.  A few things are decided here in float32 only, and a pixel for which float32 cannot decide is left alone --
// its RNG state is not stored, its index goes to a list, the fix-up kernel renders it with the literal code from that
// untouched state (the colour of a hit only scales the attenuation, physics.py:67-92: neither the path nor the number of
// draws depends on it, so abandoning a pixel at any point is harmless):
//   * the checker sign of a rectangle's texture coordinate when frequency x coordinate is within 2^-20 of an integer
//     (checker_sign_general's own float32 test; the reference's float64 sine decides otherwise), and a sphere's checker
//     colour when sphere_red's float32 approximation of (u, v) is too close to a checker edge (rf_general.h: the margin
//     at which the literal code calls the float64 atan2 / acos);
//   * the second hit test.  Rectangle: the scattered ray's plane parameter t2 = (z - p.z) / (1 + q2) must be below t_min
//     or above t_max as the reference evaluates it.  Sphere: sphere_hit's discriminant must be negative or its float32
//     certain-miss test (rf_general.h: a ray pointing away from the centre whose far root lies below t_min) must hold.
//     Anything else -- a ray that grazes the surface it started on -- abstains.
// Everything else is the literal arithmetic of rf_general.h, operation by operation (tests/test_gpu_general.py: frames
// and final RNG states equal the oracle's, no pixel budget).
#pragma once

#include "rf_coop2.h"
#include "rf_general_dense.h"
#include "rf_general_kernels.h"

namespace rf {

// arguments of the kernels that may leave pixels to the fix-up kernel (this file's, and rf_general_dense.h's)
struct GeneralOneArgs {
    GeneralArgs g;
    unsigned *redo_count; // [1], zeroed before the launch: pixels that abstained (may exceed redo_cap)
    unsigned *redo_list;  // [redo_cap]: pixel indices (e * hw + p within the launch) for the fix-up kernel
    unsigned redo_cap;    // entries the list holds; a launch that abstains more often is rendered again by the literal
                          // kernel from fresh states (rf_abi_general.hip)
    const ShapeConst *shapes; // [n][NS] (render_general_dense_kernel)
    int simple_cameras;       // every camera of the launch: canonical axes, lens radius with an exact float32 offset
    FrameConst fc; // frame sizes in the forms the jittered coordinates use (rf_math.h)
};

// a pixel that abstains: counted always, listed while the list has room
__device__ __forceinline__ void redo_append(const GeneralOneArgs &ra, unsigned pix)
{
    const unsigned slot = atomicAdd(ra.redo_count, 1u);
    if (slot < ra.redo_cap)
        ra.redo_list[slot] = pix;
}

// camera.get_ray (camera.py:307-350): general_ray of rf_general.h with the per-environment constants from the host
// (GeneralCamera::u64 ...: loop invariants the kernel would otherwise keep in 18 vector registers)
__device__ __forceinline__ void general_ray_scalar(const_as<GeneralCamera> &cam, float p0, float p1, float s, float t,
                                                   float o[3], float d[3])
{
    const double rd0 = (double)p0 * cam.lens_radius, rd1 = (double)p1 * cam.lens_radius;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = (cam.origin0[k] + (float)(cam.u64[k] * rd0)) + (float)(cam.v64[k] * rd1);
        d[k] = ((cam.lower_left0[k] + cam.f[3 + k] * s) + cam.f[6 + k] * t) - o[k];
    }
}

// (the float32 decisions that abstain -- checker_sign_dense, sphere_red_dense -- and the RF_TEST_DOUBT build that makes every
// abstention frequent and visibly wrong: rf_general_dense.h)

constexpr int kGeneralOneOcc = 6; // waves per SIMD the register allocator is held to (5 ... 7 measured: within 1 %)
// SPHERE: the environment's one shape is a sphere (sphere.py:40-117) instead of a rectangle.  A ray that scattered off
// it starts on its surface and points outwards (direction n + q, |q| < 1): the second hit test is sphere_hit's own
// float32 certain-miss test (rf_general.h), and a ray for which that test does not settle it abstains.  Across the sphere
// call a hit keeps its point p (normal and the second test's oc are (p - centre) again) and two lane masks.
// WX: waves of 32 x 2 pixels side by side in a block's tile -- 4: 128 x 6 pixels (render_kernel_coop2's default layout), 2:
// 64 x 12, for widths that leave fewer dead columns that way (300 px: 320 instead of 384).
template <bool POW2, bool SPHERE = false, int WX = 4>
__global__ __launch_bounds__(kBlock2, kGeneralOneOcc) void
render_general_one_kernel(GeneralOneArgs ra)
{
    const GeneralArgs &a = ra.g;
    // tile of a block: WX waves of 32 x 2 pixels side by side, 4 / WX down, kSets sets down; the same per-instance
    // choices for the disc tails and the list slots as render_kernel_coop2
    static_assert(WX == 4 || WX == 2, "tile layouts");
    constexpr int tWaveW = 32, tWaveH = 2, tWavesX = WX, tTileW = WX * tWaveW, tTileH = (4 / WX) * tWaveH,
                  tTileH2 = tTileH * kSets;
    constexpr bool kDiscInWave = POW2, kWaveSlots = POW2;
    __shared__ CoopLds2 lds;
    static_assert(sizeof(lds.words4) >= (size_t)kSets * kBlock2 * 3, "stage does not fit");
    uint32_t *const stage = reinterpret_cast<uint32_t *>(lds.words4);
    __shared__ float lds_colour[kColourLds][3][kBlock2];

    // (one-dimensional grid, the environment fastest: see render_general_dense_kernel)
    const int e = (int)(blockIdx.x % (unsigned)a.n);
    const int block_in_env = (int)(blockIdx.x / (unsigned)a.n);
    const int tid = threadIdx.x;
    if (tid < 2)
        lds.cnt[tid] = 0;
    if (tid == 2)
        lds.cnt2 = 0;
    __syncthreads();
    const int tiles_x = (a.w + tTileW - 1) / tTileW;
    const int tile_y = block_in_env / tiles_x, tile_x = block_in_env - tile_y * tiles_x;

    // pixel geometry of a thread, re-derived inside the loop from an index the compiler cannot see through (rf_coop2.h)
    struct Geometry {
        int col, row0, x, y0, w, h;
        static __device__ __forceinline__ int opaque(int v)
        {
            asm volatile("" : "+v"(v));
            return v;
        }
        __device__ __forceinline__ int y_of(int j) const { return y0 + j * tTileH; }
        __device__ __forceinline__ bool live_of(int j) const { return x < w && y_of(j) < h; }
    };
    auto geometry = [&](int t) {
        __builtin_assume(t >= 0 && t < kBlock2);
        const unsigned ut = (unsigned)t, wv = ut >> 6, lane = ut & 63u;
        Geometry r;
        r.col = (int)(wv & (unsigned)(tWavesX - 1)) * tWaveW + (int)(lane & (unsigned)(tWaveW - 1));
        r.row0 = (int)(wv / (unsigned)tWavesX) * tWaveH + (int)(lane / (unsigned)tWaveW);
        r.x = tile_x * tTileW + r.col;
        r.y0 = tile_y * tTileH2 + r.row0;
        r.w = a.w;
        r.h = a.h;
        return r;
    };
    auto pix_of = [&](const Geometry &q, int j) {
        return (size_t)e * a.hw + (q.live_of(j) ? (size_t)q.y_of(j) * a.w + q.x : 0);
    };
    Rng g[kSets];
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        const Geometry g0 = geometry(tid);
        g[j] = rng_load(0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull); // dead lanes: any state
        if (g0.live_of(j)) {
            const ulonglong2 st = a.states_in[pix_of(g0, j)];
            g[j] = rng_load(st.x, st.y);
        }
    }
    // the environment's camera and its one shape -- a rectangle's row: x_min, x_max, y_min, y_max, z, frequency u,
    // frequency v; a sphere's: centre x, y, z, radius, frequency u, frequency v
    const_as<GeneralCamera> &cam = *as_const(a.cameras + e);
    const_as<float> *const rp = as_const(a.params + ((size_t)e * a.most) * a.width);
    const float x_min = rp[0], x_max = rp[1], y_min = rp[2], y_max = rp[3], z_pos = rp[4], freq_u = rp[5], freq_v = rp[6];
    const float centre[3] = {rp[0], rp[1], rp[2]}, radius = rp[3], sfreq_u = rp[4], sfreq_v = rp[5];
    // block-uniform values computed with vector instructions: keep them in scalar registers (rf_coop2.h)
    auto uniform = [](float v) {
        int bits = __builtin_bit_cast(int, v);
        asm volatile("" : "+v"(bits));
        return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(bits));
    };
    const float inv_r = uniform((float)(1.0 / (double)radius)); // float32(1 / radius), as sphere_hit has it
    const float den_u = uniform(x_max - x_min), den_v = uniform(y_max - y_min); // rectangle.py:168-169
    // the two divisors of uv are the rectangle's: with their reciprocals the correctly rounded quotient is three
    // operations (rf_math.h div_by_const, proven for divisors in [2^-40, 2^40]: anything else divides)
    const float rden_u = uniform(1.0f / den_u), rden_v = uniform(1.0f / den_v);
    const int quick_div = __builtin_amdgcn_readfirstlane(
        (int)(den_u >= 9.094947017729282e-13f && den_u <= 1099511627776.0f && den_v >= 9.094947017729282e-13f &&
              den_v <= 1099511627776.0f));

    float cr[kSets], cg[kSets], cb[kSets];
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        cr[j] = cg[j] = cb[j] = 0.0f;
        if (j < kColourLds)
            lds_colour[j][0][tid] = lds_colour[j][1][tid] = lds_colour[j][2][tid] = 0.0f;
    }
    int sphere_trips = kCoopTrips2; // in-wave sphere attempts of the current sample (block-uniform)
    lanemask live_m[kSets];
#pragma unroll
    for (int j = 0; j < kSets; ++j)
        live_m[j] = lanes_where(geometry(tid).x < a.w) & lanes_where(geometry(tid).y_of(j) < a.h);
    constexpr float t_min = 0.001f, t_max = 1000000.0f; // physics.py:121

    for (int k = 0; k < a.spp; ++k) {
        uint32_t w[kSets][6];
        float s[kSets], t[kSets];
        lanemask need_m[kSets];
        const Geometry gk = geometry(Geometry::opaque(tid));
#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            // render.py:61-66 (the x coordinate's draw first): sample_coords is general_coords for frames up to 4096
            sample_coords<POW2>(g[j], gk.x, gk.y_of(j), (float)gk.x, (float)gk.y0 + (float)(j * tTileH), ra.fc, s[j], t[j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                w[j][i] = any_u32();
            const float sq = disc_attempt_sq(g[j], w[j]);
            need_m[j] = live_m[j] & ~lanes_where(sq < 1.0f);
        }
        if (kDiscInWave)
            disc_tails_wave(lds.state[0], need_m, g, w, tid);
        else
            coop_finish2m<2, kWaveSlots, false>(lds, 0, &lds.cnt[0], need_m, g, w, tid);

        // After this loop, per set: a lane that missed keeps its ray's direction in rd (the sky it sees); a lane that hit
        // keeps z - p.z in rd[0] (the numerator of the scattered ray's plane test), its attenuation's red channel (1 / 0)
        // in rd[1] and NaN in rd[2] if its colour could not be decided -- its direction is (q0, q1, 1 + q2).
        float rd[kSets][3];
        lanemask hit_m[kSets];
        lanemask red_m[kSets], doubt_m[kSets]; // (SPHERE: rd holds the hit point)
#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            float p0, p1;
            disc_finish(w[j], p0, p1);
            float o[3], d[3];
            // (cameras with canonical axes and a lens radius whose float32 offset is exact -- all of the launch's, the host
            // says: rf_general_dense.h -- need no float64 lens products: +3 % on one-rectangle scenes)
            if (scalar_now(ra.simple_cameras))
                general_ray_simple(cam, cam.lens_hi, cam.lens_lo, p0, p1, s[j], t[j], o, d);
            else
                general_ray_scalar(cam, p0, p1, s[j], t[j], o, d);
            red_m[j] = doubt_m[j] = 0;
            if (SPHERE) {
                const float sp[4] = {centre[0], centre[1], centre[2], radius};
                HitRec rec;
                rec.p[0] = rec.p[1] = rec.p[2] = 0.0f;
                rec.n[0] = rec.n[1] = rec.n[2] = 0.0f;
                bool hit = false, red = false, doubt = false;
                // (the literal float64 roots: the dense kernel's double-float form is 2 % slower here -- this kernel's
                // occupancy is set by its three pixel sets, not by the roots: profiles/r05_ab.txt section 11)
                if (lane_in(live_m[j]))
                    hit = sphere_hit(sp, o, d, t_min, t_max, rec);
                hit_m[j] = live_m[j] & lanes_where(hit);
                if (lane_in(hit_m[j]))
                    red = sphere_red_dense(rec.n, sfreq_u, sfreq_v, doubt);
                red_m[j] = hit_m[j] & lanes_where(red);
                doubt_m[j] = hit_m[j] & lanes_where(doubt);
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    rd[j][i] = hit ? rec.p[i] : d[i];
            } else {
            // rectangle.py:49-99 hit
            const float th = (z_pos - o[2]) / d[2];
            bool hit = !(th < t_min || th > t_max);
            float px = 0.0f, py = 0.0f, pz = 0.0f;
            if (hit) {
                px = add2(o[0], d[0] * th);
                py = add2(o[1], d[1] * th);
                pz = add2(o[2], d[2] * th);
                hit = !(px < x_min || px > x_max || py < y_min || py > y_max);
            }
            hit_m[j] = live_m[j] & lanes_where(hit);
            bool red = false, doubt = false;
            if (lane_in(hit_m[j])) { // rectangle.py:151-170 uv, physics.py:47-64 colour_checkerboard
                float u, v;
                if (scalar_now(quick_div)) { // block-uniform, a scalar
                    u = div_by_const(px - x_min, den_u, rden_u);
                    v = div_by_const(py - y_min, den_v, rden_v);
                } else {
                    u = (px - x_min) / den_u;
                    v = (py - y_min) / den_v;
                }
                red = checker_sign_dense(freq_u, u, doubt) * checker_sign_dense(freq_v, v, doubt) > 0;
            }
            rd[j][0] = hit ? z_pos - pz : d[0];
            rd[j][1] = hit ? (red ? 1.0f : 0.0f) : d[1];
            rd[j][2] = hit ? (doubt ? __builtin_nanf("") : 0.0f) : d[2];
            }
            // (one set's ray at a time: interleaved by the scheduler, three sets' float64 temporaries do not fit the budget)
            __builtin_amdgcn_sched_barrier(0);
            w[j][4] = any_u32();
            w[j][5] = any_u32();
            need_m[j] = hit_m[j];
#pragma unroll
            for (int trip = 0; trip < kCoopTrips2 + 1; ++trip) {
                if (trip >= kCoopTrips2 && scalar_now(sphere_trips) <= kCoopTrips2) // block-uniform, a scalar
                    break;
                if (need_m[j] != 0) { // wave-uniform
                    float sq = 2.0f;
                    if (lane_in(need_m[j]))
                        sq = sphere_attempt_sq(g[j], w[j]);
                    asm volatile("" : "+v"(sq));
                    need_m[j] &= ~lanes_where(sq < 1.0f);
                }
            }
        }
        int *const sphere_cnt = &lds.cnt[kDiscInWave ? (k & 1) : 1];
        const int stragglers = __builtin_amdgcn_readfirstlane(
            coop_finish2m<3, kWaveSlots, kDiscInWave>(lds, 1, sphere_cnt, need_m, g, w, tid));
        if (sphere_trips == kCoopTrips2 && stragglers > kCoopCap + kAdaptOn)
            sphere_trips = kCoopTrips2 + 1;
        else if (sphere_trips != kCoopTrips2 && 2 * stragglers < kCoopCap + kAdaptOff)
            sphere_trips = kCoopTrips2;

#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            float dir[3] = {rd[j][0], rd[j][1], rd[j][2]};
            float ar = 1.0f, ag = 1.0f, ab = 1.0f;
            bool doubt = false;
            if (SPHERE && lane_in(hit_m[j])) {
                float q0, q1, q2;
                sphere_finish(w[j], q0, q1, q2);
                // sphere.py:92-101 normal, physics.py:81-87 scatter
                const float oc[3] = {rd[j][0] - centre[0], rd[j][1] - centre[1], rd[j][2] - centre[2]};
                dir[0] = add2(oc[0] * inv_r, q0);
                dir[1] = add2(oc[1] * inv_r, q1);
                dir[2] = add2(oc[2] * inv_r, q2);
                ar = lane_in(red_m[j]) ? 1.0f : 0.0f;
                ag = 1.0f - ar;
                ab = 0.0f;
                // the scattered ray against the sphere again (sphere_hit, rf_general.h, up to its certain misses)
                const float qa = dot3(dir, dir), qb = dot3(oc, dir);
                const float qc = dot3(oc, oc) - radius * radius;
                const float disc = qb * qb - qa * qc;
                bool miss = disc < 0;
                if (qa > 0.0f && qb > 0.0f) {
                    const float reach = qb + t_min * qa;
                    miss = miss || disc < (reach * reach) * 0.99999904632568359375f /* 1 - 2^-20 */;
                }
                doubt = !miss || lane_in(doubt_m[j]);
#if RF_TEST_DOUBT
                if ((w[j][5] & 0xF000u) == 0) {
                    doubt = true;
                    dir[1] = -dir[1];
                }
#endif
            } else if (lane_in(hit_m[j])) {
                float q0, q1, q2;
                sphere_finish(w[j], q0, q1, q2);
                // physics.py:81-87 scatter from the rectangle's normal (0, 0, 1); attenuation red or green
                dir[0] = add2(0.0f, q0);
                dir[1] = add2(0.0f, q1);
                dir[2] = add2(1.0f, q2);
                ar = rd[j][1];        // red: (1, 0, 0), green: (0, 1, 0)
                ag = 1.0f - rd[j][1];
                ab = 0.0f;
                // the scattered ray against the same plane (rectangle.py:63-71): it has to miss by its parameter
                // (t2 = num / dir.z; with dir.z > 0, num < (t_min dir.z)(1 - 2^-20) implies RN32(num / dir.z) < t_min -- the
                // bound lies four float32 steps below t_min, the factor covers the two roundings of the right side --,
                // which is every ray but the grazing ones: those divide)
                const float num = rd[j][0];
                bool grazing = !(dir[2] > 0.0f && num < (t_min * dir[2]) * 0.99999904632568359375f /* 1 - 2^-20 */);
                if (__builtin_expect(__any(grazing), 0)) {
                    if (grazing) {
                        const float t2 = num / dir[2];
                        grazing = !(t2 < t_min || t2 > t_max);
                    }
                }
#if RF_TEST_DOUBT
                if ((w[j][5] & 0xF000u) == 0) { // (one scattered ray in sixteen: "grazing", and visibly wrong if it is kept)
                    grazing = true;
                    dir[1] = -dir[1];
                }
#endif
                doubt = grazing || rd[j][2] != 0.0f; // (NaN in rd[2]: the checker colour was not decided)
            }
            Colour c = sky_colour(dir, ar, ag, ab); // physics.py:137-145
            // A pixel that abstains carries NaN in its red sum from here on (NaN + x = NaN): no mask to keep across the
            // loop.  (A sum that is NaN for another reason -- a ray of length zero -- sends its pixel to the literal
            // code too, which then computes that NaN itself.)
            c.r = doubt ? __builtin_nanf("") : c.r;
            if (j < kColourLds) {
                lds_colour[j][0][tid] = add2(lds_colour[j][0][tid], c.r);
                lds_colour[j][1][tid] = add2(lds_colour[j][1][tid], c.g);
                lds_colour[j][2][tid] = add2(lds_colour[j][2][tid], c.b);
                continue;
            }
            cr[j] = add2(cr[j], c.r);
            cg[j] = add2(cg[j], c.g);
            cb[j] = add2(cb[j], c.b);
        }
    }
#pragma unroll
    for (int j = 0; j < kColourLds && j < kSets; ++j) {
        cr[j] = lds_colour[j][0][tid];
        cg[j] = lds_colour[j][1][tid];
        cb[j] = lds_colour[j][2][tid];
    }
    __syncthreads(); // the cooperative arrays are dead from here on: words4 becomes the stage

    uint8_t *sb = reinterpret_cast<uint8_t *>(stage);
    const Geometry ge = geometry(Geometry::opaque(tid));
    const bool staged = (a.w & 3) == 0;
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        uint8_t r8 = 0, g8 = 0, b8 = 0;
        const bool keep = ge.live_of(j) && cr[j] == cr[j]; // (NaN: the pixel abstained)
        if (ge.live_of(j) && !keep) // abstain: state untouched, pixel listed for render_general_fixup_kernel
            redo_append(ra, (unsigned)pix_of(ge, j));
        if (keep) {
            a.states[pix_of(ge, j)] = make_ulonglong2(rng_s0(g[j]), rng_s1(g[j]));
            r8 = (uint8_t)(cr[j] * a.scale);
            g8 = (uint8_t)(cg[j] * a.scale);
            b8 = (uint8_t)(cb[j] * a.scale);
        }
        if (staged) {
            const int slot = (j * tTileH + ge.row0) * tTileW + ge.col;
            sb[slot * 3 + 0] = r8;
            sb[slot * 3 + 1] = g8;
            sb[slot * 3 + 2] = b8;
        } else if (keep) {
            uint8_t *dst = a.frames + pix_of(ge, j) * 3;
            dst[0] = r8;
            dst[1] = g8;
            dst[2] = b8;
        }
    }
    if (staged) {
        // the tile's rows (tTileW * 3 B each) -> LDS -> coalesced dword stores per row (a listed pixel's bytes are
        // placeholders the fix-up kernel overwrites)
        __syncthreads();
        constexpr int kRowDw = tTileW * 3 / 4;
        for (int i = tid; i < tTileH2 * kRowDw; i += kBlock2) {
            const int r = i / kRowDw, dw = i - r * kRowDw;
            const int yy = tile_y * tTileH2 + r;
            const int valid_dw = min(tTileW, a.w - tile_x * tTileW) * 3 / 4; // w % 4 == 0
            if (yy < a.h && dw < valid_dw) {
                uint32_t *dst = reinterpret_cast<uint32_t *>(
                    a.frames + (((size_t)e * a.h + yy) * a.w + (size_t)tile_x * tTileW) * 3);
                dst[dw] = stage[r * kRowDw + dw];
            }
        }
    }
}

// The listed pixels, literally (rf_general.h render_pixel_general: float64 sines where float32 cannot decide, any
// number of bounces).  Runs after the first kernel on the same stream; grid-stride over the list, whose length it reads
// itself.  The list is short (a launch's pixels / 10^3) and the kernel's duration is the LATENCY of one pixel's samples
// -- about 5 us per sample for a wave, whatever it executes and however many lanes it has (profiles/r05_ab.txt section
// 13: 80 us at 16 samples, 0.5 ms at 100).  A wave takes as few pixels as still lets every listed pixel's wave be
// resident at once (kFixupWaves waves, about two per SIMD: 1 ... 64 pixels per wave, chosen from the list's length):
// fewer unrelated pixels per wave, fewer max-over-lanes trips (-10 ... -18 % of this kernel's time against a fixed 32).
// The registers are unbounded (no spills).
#ifndef RF_FIXUP_WAVES
#define RF_FIXUP_WAVES 2048
#endif
constexpr unsigned kFixupWaves = RF_FIXUP_WAVES;
constexpr unsigned kFixupBlocks = 1024; // the grid: 4 096 waves, all resident (124 registers: 4 waves per SIMD)
template <bool POW2>
__global__ __launch_bounds__(kBlock) void render_general_fixup_kernel(GeneralOneArgs ra)
{
    const GeneralArgs &a = ra.g;
    const unsigned total = min(*ra.redo_count, ra.redo_cap); // (more than the list holds: the host renders the launch again)
    const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, waves = gridDim.x * (kBlock / 64);
    unsigned lanes = 1; // pixels per wave
    while (lanes < 64u && (total + lanes - 1) / lanes > kFixupWaves)
        lanes <<= 1;
    if (lane >= lanes)
        return;
    for (unsigned i = wave * lanes + lane; i < total; i += waves * lanes) {
        const unsigned pix = ra.redo_list[i];
        const int e = (int)(pix / (unsigned)a.hw), p = (int)(pix - (unsigned)e * (unsigned)a.hw);
        const int y = p / a.w, x = p - y * a.w;
        const ulonglong2 st = a.states_in[pix];
        Rng g = rng_load(st.x, st.y);
        float cr, cg, cb;
        render_pixel_general<POW2>(g, x, y, a.h, a.w, a.spp, a.cameras[e], a.params + ((size_t)e * a.most) * a.width,
                                   a.types + (size_t)e * a.most, a.sizes[e], a.width, cr, cg, cb);
        a.states[pix] = make_ulonglong2(rng_s0(g), rng_s1(g));
        uint8_t *dst = a.frames + (size_t)pix * 3;
        dst[0] = (uint8_t)(cr * a.scale);
        dst[1] = (uint8_t)(cg * a.scale);
        dst[2] = (uint8_t)(cb * a.scale);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// render_general_dense_kernel<POW2, NS>: device_render (render.py:31-85) for worlds of NS shapes per environment in
// float32 / double-float arithmetic with pixel-level abstention (rf_general_dense.h).  One pixel per thread, lanes along
// x, rejection and bounce loops inside the wave, frame bytes staged through LDS -- the literal kernel's organisation:
// what changes is what a sample costs.  Scene constants come through the constant address space into scalar registers.
constexpr int kDenseOcc = 7; // waves per SIMD the register allocator is held to
// TILED: which pixels a block's 256 threads own -- a scheduling choice only (a pixel's RNG stream is its own): a 16 x 16
// tile with waves of 8 x 8 pixels, or (frames that such tiles would pad too much: rf_abi_general.hip) 256 consecutive
// pixels of the frame in row-major order.  Compact waves touch fewer of a shape's edge pixels, so fewer waves run the
// hit-only code for a few lanes: 12 % fewer instructions on two-sphere scenes (lane utilisation 0.56 -> 0.64), and
// row-major waves of a frame whose width is not a multiple of 64 straddle rows: +17 % at 300 px.  (16 x 4 pixel waves
// are as fast, 32 x 2 ones 1 ... 6 % slower: profiles/r05_ab.txt section 3.)
// SIMPLE: every camera of the launch has canonical axes and a lens radius whose float32 offset is exact (rf_general_dense.h)
template <bool POW2, int NS, bool TILED, bool SIMPLE>
__global__ __launch_bounds__(kBlock, kDenseOcc) void render_general_dense_kernel(GeneralOneArgs ra)
{
    __shared__ uint32_t stage[kBlock * 3 / 4];
    const GeneralArgs &a = ra.g;
    // One-dimensional grid with the ENVIRONMENT as the fastest index (block b: environment b % n, tile b / n).  Blocks
    // go to the eight XCDs round-robin by their index; with the tile index fastest, a scene whose expensive tiles sit at
    // fixed columns (a sphere at the frame's edge: tile columns 0-5 and 10-15 of 16) gives some XCDs twice the work of
    // others (measured: 0.24 instead of 0.34 VALU instructions per cycle per SIMD, profiles/r05_ab.txt section 3).
    const int e = (int)(blockIdx.x % (unsigned)a.n);
    const int block_in_env = (int)(blockIdx.x / (unsigned)a.n);
    constexpr int tTileW = 16, tTileH = 16;
    int x, y, p0 = 0, x0 = 0, y0 = 0;
    bool live;
    if (!TILED) {
        p0 = block_in_env * kBlock;
        const int p = p0 + threadIdx.x;
        live = p < a.hw;
        y = p / a.w;
        x = p - y * a.w;
    } else {
        const int tiles_x = (a.w + tTileW - 1) / tTileW;
        const int ty = block_in_env / tiles_x, tx = block_in_env - ty * tiles_x;
        const unsigned wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        x0 = tx * tTileW;
        y0 = ty * tTileH;
        x = x0 + (int)((wv & 1u) * 8u + (lane & 7u));
        y = y0 + (int)((wv >> 1) * 8u + (lane >> 3));
        live = x < a.w && y < a.h;
    }
    const size_t pix = (size_t)e * a.hw + (live ? (size_t)y * a.w + x : 0);
    uint8_t r8 = 0, g8 = 0, b8 = 0;
    if (live) {
        const ulonglong2 st = a.states_in[pix];
        Rng g = rng_load(st.x, st.y);
        const_as<GeneralCamera> &cam = *as_const(a.cameras + e);
        const_as<ShapeConst> *const sc = as_const(ra.shapes + (size_t)e * NS);
        const int n_shapes = as_const(a.sizes)[e];
        float cr, cg, cb;
        const bool keep = render_pixel_dense<POW2, NS, SIMPLE>(g, x, y, a.spp, cam, sc, n_shapes, ra.fc, cr, cg, cb);
        if (keep) {
            a.states[pix] = make_ulonglong2(rng_s0(g), rng_s1(g));
            r8 = (uint8_t)(cr * a.scale);
            g8 = (uint8_t)(cg * a.scale);
            b8 = (uint8_t)(cb * a.scale);
        } else { // abstain: state untouched, pixel listed for render_general_fixup_kernel (its bytes below are placeholders)
            redo_append(ra, (unsigned)pix);
        }
    }
    uint8_t *sb = reinterpret_cast<uint8_t *>(stage);
    if (!TILED) {
        // a full block whose first byte is dword-aligned goes through LDS; anything else stores bytes (render_general_kernel)
        const size_t first_byte = ((size_t)e * a.hw + p0) * 3;
        const bool staged = p0 + kBlock <= a.hw && (reinterpret_cast<uintptr_t>(a.frames + first_byte) & 3) == 0; // block-uniform
        if (staged) {
            sb[threadIdx.x * 3 + 0] = r8;
            sb[threadIdx.x * 3 + 1] = g8;
            sb[threadIdx.x * 3 + 2] = b8;
            __syncthreads();
            if (threadIdx.x < kBlock * 3 / 4)
                reinterpret_cast<uint32_t *>(a.frames + first_byte)[threadIdx.x] = stage[threadIdx.x];
            return;
        }
    } else {
        // a whole tile inside a frame whose rows are dwords (w % 4 == 0) and whose base is dword-aligned: the tile's rows
        // (tTileW * 3 bytes each) go through LDS and leave as dword stores; anything else stores bytes
        const bool staged = x0 + tTileW <= a.w && y0 + tTileH <= a.h && (a.w & 3) == 0 &&
                            (reinterpret_cast<uintptr_t>(a.frames + (size_t)e * a.hw * 3) & 3) == 0; // block-uniform
        if (staged) {
            const int slot = (y - y0) * tTileW + (x - x0);
            sb[slot * 3 + 0] = r8;
            sb[slot * 3 + 1] = g8;
            sb[slot * 3 + 2] = b8;
            __syncthreads();
            constexpr int kRowDw = tTileW * 3 / 4;
            if (threadIdx.x < tTileH * kRowDw) {
                const int r = threadIdx.x / kRowDw, dw = threadIdx.x - r * kRowDw;
                uint32_t *dst = reinterpret_cast<uint32_t *>(a.frames + (((size_t)e * a.h + y0 + r) * a.w + x0) * 3);
                dst[dw] = stage[r * kRowDw + dw];
            }
            return;
        }
    }
    if (live) {
        uint8_t *dst = a.frames + pix * 3;
        dst[0] = r8;
        dst[1] = g8;
        dst[2] = b8;
    }
}

} // namespace rf
