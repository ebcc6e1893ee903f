"""Development benchmark of the general renderer (rf_render_general):
the reference's notebook-style scenes (one rectangle / one sphere / two shapes per environment,
per-environment cameras, 50-bounce find_colour), frames left on the device.
usage (GPU box):  python tools/bench_general.py [n_envs] [frame] [spp] [--scene one_rect|one_sphere|two_sphere|mixed]"""
import sys
import time

import numpy as np

import os  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reinfocus_amd import _native  # noqa: E402
from reinfocus_amd.graphics import camera, shape_factory as sf, world  # noqa: E402


def main():
    argv = list(sys.argv[1:])
    only = None
    if "--scene" in argv:  # one scene only (profiles/run_profiles.sh profiles them separately)
        at = argv.index("--scene")
        only = argv[at + 1]
        del argv[at:at + 2]
    n = int(argv[0]) if len(argv) > 0 else 256
    frame = int(argv[1]) if len(argv) > 1 else 256
    spp = int(argv[2]) if len(argv) > 2 else 16
    rng = np.random.default_rng(0)
    ctx = _native.Context(0)
    for label, make in (("one_rect", sf.one_rect), ("one_sphere", sf.one_sphere), ("two_sphere", None),
                        ("mixed", None)):
        if only is not None and label != only:
            rng.uniform(5, 10, 2 * n)  # keep the other scenes' parameters what they are in a full run
            continue
        if make is not None:
            shapes = [make(sf.ShapeParameters(distance=float(d), r_size=20)) for d in rng.uniform(5, 10, n)]
        elif label == "two_sphere":
            shapes = [sf.two_sphere(sf.ShapeParameters(float(d) + 10), sf.ShapeParameters(float(d)))
                      for d in rng.uniform(5, 10, n)]
        else:
            shapes = [sf.mixed(sf.ShapeParameters(float(d)), sf.ShapeParameters(float(d) + 5))
                      for d in rng.uniform(5, 10, n)]
        worlds = world.Worlds(*shapes)
        cams = camera.Cameras(*[camera.make_gpu_camera(focus_distance=float(f)) for f in rng.uniform(5, 10, n)])
        params, types, sizes = worlds.device_data()
        params = np.ascontiguousarray(np.pad(params, ((0, 0), (0, 0), (0, max(0, 7 - params.shape[2])))),
                                      dtype=np.float32)
        types, sizes = np.ascontiguousarray(types, dtype=np.int32), np.ascontiguousarray(sizes, dtype=np.int32)
        cameras = np.ascontiguousarray(cams.device_data(), dtype=np.float64)
        lib, h = ctx._lib, ctx._h
        args = (h, n, frame, frame, spp, _native._ptr(cameras), _native._ptr(params), _native._ptr(types),
                _native._ptr(sizes), params.shape[1], params.shape[2], None)
        _native._check(lib.rf_render_general(*args))
        ctx.synchronize()
        times = []
        for _ in range(5):  # (the median: now and then a call on a shared box takes twice as long)
            t0 = time.perf_counter()
            _native._check(lib.rf_render_general(*args))
            ctx.synchronize()
            times.append(time.perf_counter() - t0)
        dt = sorted(times)[len(times) // 2]
        print(f"{label}: {n} envs x {frame}^2 x {spp} spp: {dt * 1e3:.2f} ms per render "
              f"(incl. re-seeding), {n * frame * frame * spp / dt / 1e9:.2f} G samples/s, "
              f"{ctx.general_redo_pixels() / (n * frame * frame):.4%} of the pixels fixed up, "
              f"{ctx.render_kernel_name()}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
