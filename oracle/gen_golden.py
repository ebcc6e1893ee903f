"""Writes tests/golden/*.npz from the numpy-1.26 restatement (np126_restatement.py).

Run:  /opt/conda/bin/python3.9 oracle/gen_golden.py
(needs numpy 1.26.x; the system interpreter's numpy 2.x has NEP-50 promotion and
would not compute what the reference computes).  The outputs are small, committed,
and consumed by tests/test_oracle_golden.py (C oracle) and tests/test_gpu_parity.py
(HIP path).  TEST INFRASTRUCTURE, not the product.
"""

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import np126_restatement as r  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def states_array(states):
    return np.array(states, dtype=np.uint64)


def rng_vectors():
    st = r.seed_states(6, 0)
    seeded = states_array(st)
    s0 = list(st[0])
    raw = np.array([r.nxt(s0) for _ in range(8)], dtype=np.uint64)
    s1 = list(st[3])
    uni = np.array([r.uniform(s1) for _ in range(16)], dtype=np.float32)
    st7 = states_array(r.seed_states(3, 7))
    # a draw that rounds to exactly 1.0f: (x >> 11) = 2**53 - 1
    one = r.f32(np.float64((1 << 53) - 1) * (np.float64(1) / np.float64(1 << 53)))
    np.savez(os.path.join(OUT, "rng.npz"), seeded=seeded, raw=raw, uniform=uni,
             seeded_seed7=st7, top_draw=np.float32(one))


def packing_vectors():
    fps = np.array([10, 5, 7.3, 9.99, 5.311405, 8.98009, 1, 40], dtype=np.float32)
    dyn, origin, u, v, lens = r.pack_cameras(fps)
    tg = np.array([1, 2, 3, 10, 5.311405, 7.77, 9.125], dtype=np.float32)
    rect20 = r.pack_worlds(tg, 20)
    rect30 = r.pack_worlds(tg, 30)
    np.savez(os.path.join(OUT, "packing.npz"), focus_planes=fps, cam_dyn=dyn,
             origin=np.array(origin, dtype=np.float32), u=np.array(u, dtype=np.float32),
             v=np.array(v, dtype=np.float32), lens_radius=np.float64(lens), targets=tg,
             rect_r20=rect20, rect_r30=rect30)


def device_fn_vectors():
    """Single-call known answers for the unit device functions, seed 0."""
    st = r.seed_states(4, 0)
    disc = np.array([r.unit_disc(st[0]) for _ in range(8)], dtype=np.float32)
    sphere = np.array([r.unit_sphere(st[1]) for _ in range(8)], dtype=np.float32)
    dyn, origin, u, v, lens = r.pack_cameras([7.5])
    cam = (tuple(dyn[0, 0]), tuple(dyn[0, 1]), tuple(dyn[0, 2]), origin, u, v, lens)
    rays = []
    for s, t in ((0.5, 0.5), (0.1, 0.9), (0.999, 0.001)):
        ro, rd = r.get_ray(cam, r.f32(s), r.f32(t), st[2])
        rays.append(list(ro) + list(rd))
    rect = r.pack_worlds([6.25])[0]
    cols = []
    for ro_rd in rays:
        cols.append(r.fast_find_colour(rect, tuple(ro_rd[0:3]), tuple(ro_rd[3:6]), st[3]))
    np.savez(os.path.join(OUT, "device_fns.npz"), disc=disc, sphere=sphere,
             cam_dyn=dyn, rect=rect, st_s=np.array([0.5, 0.1, 0.999], dtype=np.float32),
             st_t=np.array([0.5, 0.9, 0.001], dtype=np.float32),
             rays=np.array(rays, dtype=np.float32), colours=np.array(cols, dtype=np.float32),
             states_after=states_array(st))


def render_case(name, targets, focus, h, w, spp, r_size=20, passes=1):
    n = len(targets)
    cams = r.pack_cameras(focus)
    rects = r.pack_worlds(targets, r_size)
    st = r.seed_states(n * h * w, 0)
    out = {}
    for p in range(passes):
        frames, colours = r.render(cams, rects, h, w, spp, st)
        out["frames%d" % p] = frames
        out["colours%d" % p] = colours
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        targets=np.asarray(targets, dtype=np.float32),
                        focus=np.asarray(focus, dtype=np.float32),
                        h=h, w=w, spp=spp, r_size=r_size, passes=passes,
                        states_after=states_array(st), **out)
    print(name, "done", flush=True)


def general_case(name, cams, env_shapes, h, w, spp):
    cameras = r.pack_general_cameras(cams)
    world = r.pack_worlds_general(env_shapes)
    st = r.seed_states(len(env_shapes) * h * w, 0)
    frames = r.render_general(cameras, world, h, w, spp, st)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), cameras=cameras, params=world[0], types=world[1],
                        sizes=world[2], h=h, w=w, spp=spp, frames=frames, states_after=states_array(st))
    print(name, "done", flush=True)


def general_vectors():
    tan15 = __import__("math").tan(__import__("math").radians(15))
    tan10 = __import__("math").tan(__import__("math").radians(10))
    one_sphere = [r.sphere_shape((0, 0, -10.0), 10.0 * tan10)]
    two_sphere = [r.sphere_shape((-20.0 * tan15, 0, -20.0), 20.0 * tan10), r.sphere_shape((5.0 * tan15, 0, -5.0), 5.0 * tan10)]
    one_rect = [r.rectangle_shape((-10 * tan10, 10 * tan10), (-10 * tan10, 10 * tan10), -10.0)]
    mixed = [r.sphere_shape((-5.0 * tan15, 0, -5.0), 5.0 * tan10),
             r.rectangle_shape((10 * tan15 - 10 * tan10, 10 * tan15 + 10 * tan10), (-10 * tan10, 10 * tan10), -10.0)]
    facing = [r.rectangle_shape((-3, 3), (-3, 3), -6.0, (4, 4)), r.sphere_shape((0, 0, -4.0), 1.0, (8, 8))]
    cams = [r.make_gpu_camera(), r.make_gpu_camera(focus_distance=5.0),
            r.make_gpu_camera(aperture=0.5, focus_distance=8.0, vfov=40, aspect_ratio=1.5),
            r.make_gpu_camera(look_from=(0.5, 0.3, 1.0), look_at=(0, 0, -6), focus_distance=6.0, aperture=0.2)]
    general_case("general_small", cams, [one_sphere, two_sphere, mixed, facing], 12, 18, 3)
    general_case("general_rect", [r.make_gpu_camera()], [one_rect], 16, 16, 4)
    # single-function known answers
    hit = r.sphere_hit(r.sphere_shape((0, 0, 0), 1, (4, 8))[0], r.v3(10, 0, 0), r.v3(-1, 0, 0), r.f32(0), r.f32(100))
    flat = [1.0] + list(hit[0]) + list(hit[1]) + [hit[2]] + list(hit[3]) + list(hit[4]) + [hit[5]]
    np.savez(os.path.join(OUT, "general_fns.npz"), sphere_hit=np.array(flat, dtype=np.float32),
             cameras=r.pack_general_cameras(cams))


def main():
    os.makedirs(OUT, exist_ok=True)
    rng_vectors()
    packing_vectors()
    device_fn_vectors()
    # power-of-two frame (exact s/t scaling), two passes to cover state carry-over
    render_case("render_pow2", [10, 5.3, 8.1], [10, 9.7, 5.2], 16, 16, 3, passes=2)
    # non-power-of-two frame: float64 division in s/t matters
    render_case("render_npot", [7.0, 9.5], [7.0, 5.5], 12, 12, 5)
    # reference render_test.py:86-98 geometry (r_size 30, every ray hits), tiny
    render_case("render_rsize30", [10], [10], 10, 10, 4, r_size=30)
    # BASELINE.json configs[0] shape: 1 env, 64x64, 1 spp
    render_case("render_cfg1", [7.5], [6.0], 64, 64, 1)
    # wider coverage of the rejection loops / edge pixels
    render_case("render_mid", [5.0, 6.6, 8.3, 10.0], [10.0, 6.6, 7.9, 5.0], 32, 32, 8)
    general_vectors()


if __name__ == "__main__":
    main()
