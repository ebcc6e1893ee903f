"""A performance guard for the kernels that carry the numbers in profiles/: the fused two-pass instance of the fast path
(the headline's kernel), the strip kernel of the reference's default 300 x 300 x 100 shape, the wave-cooperative kernel at
the reference's training shape (8 x 300 x 300 x 100), the focus kernel at the headline's frames, and the general
renderer's one-shape and dense kernels.  Each is timed with HIP events on the context's own stream (rf_timing) around a few
launches at a size that fills the device (the best of four counts), and must reach FLOOR x the rate recorded in profiles/r06_perf_guard.json --
measured by this very test on the round-6 tree (REINFOCUS_PERF_GUARD_RECORD=<file> writes what a run measures).  The kernel
each launch took is asserted too: a change of the dispatch that sends a shape to another kernel shows up by name.  The
floor (0.90) is wide enough for the boxes of the pool (the same build measured up to 4 % apart on different boxes, kernel time),
and narrow enough for what a careless edit costs: see profiles/r05_ab.txt section 5 for the builds it was tried on.
Marked `perf`, not `gpu`: a wall-clock floor belongs in a run of its own (tools/gpu/run.sh guard), where a noisy or
shared box cannot abort the parity suite behind `pytest -x -m gpu`."""

import json
import os

import numpy as np
import pytest

from tests import helpers
from tests.test_general_renderer import _factory_worlds

# (`-m "not gpu"` on a host without a GPU selects this file too: it skips itself there)
pytestmark = [pytest.mark.perf, pytest.mark.skipif(not os.path.exists("/dev/kfd"), reason="needs a real MI355X")]

RECORD = os.path.join(helpers.ROOT, "profiles", "r06_perf_guard.json")
FLOOR = 0.90


def _measure():
    from reinfocus_amd import _native
    from reinfocus_amd.environments import harness

    got = {}
    # 1. the fused environment step's render: render_kernel_coop2<.., TWO> (1024 x 256^2 x 16, three steps)
    env = harness.DeviceVectorDiscreteSteps(num_envs=1024, frame_height=256, samples_per_pixel=16, seed=3, device=0)
    env.reset()
    rng = np.random.default_rng(1)
    env.step(rng.integers(0, 13, 1024))  # (warm-up)
    env._ctx.timing(True)
    rates = []
    for _ in range(4):  # (the best of four launches: one slow launch on a shared box is not a regression)
        before, ms = _native.pixels_rendered(), env._ctx.timing_read()["render_ms"]
        env.step(rng.integers(0, 13, 1024))
        rates.append((_native.pixels_rendered() - before) * 16 / ((env._ctx.timing_read()["render_ms"] - ms) * 1e-3) / 1e9)
    got["fused_step_1024x256x16"] = {"kernel": env._ctx.render_kernel_name(), "g_samples_per_s": max(rates)}
    env.close()
    # 2. the strip kernel: 128 x 300^2 x 100
    ctx = _native.Context(0)
    n, h, spp = 128, 300, 100
    ctx.seed(n * h * h, 0, 0)
    ctx.set_scene(*helpers.pack_scene(*helpers.random_scene(np.random.default_rng(2), n)))
    ctx.render(n, h, h, spp)
    ctx.timing(True)

    def best_of(launch, samples, repeats=4):
        times = []
        for _ in range(repeats):
            ms = ctx.timing_read()["render_ms"]
            launch()
            times.append(ctx.timing_read()["render_ms"] - ms)
        return samples / (min(times) * 1e-3) / 1e9

    got["strip_128x300x100"] = {"kernel": ctx.render_kernel_name(),
                                "g_samples_per_s": best_of(lambda: ctx.render(n, h, h, spp), n * h * h * spp)}
    ctx.timing(False)
    # 2b. the focus kernel on 2048 frames of 256 x 256 (focus_kernel_roll: 3 bytes per pixel from HBM)
    n, h = 2048, 256
    ctx.seed(n * h * h, 0, 0)
    ctx.set_scene(*helpers.pack_scene(*helpers.random_scene(np.random.default_rng(5), n)))
    ctx.render(n, h, h, 1)
    ctx.focus(n, h, h, 15)
    ctx.timing(True)
    times = []
    for _ in range(16):  # (0.12 ms launches: the best of sixteen -- single launches were measured 3.0 ... 3.4 TB/s on one box)
        ms = ctx.timing_read()["focus_ms"]
        ctx.focus(n, h, h, 15)
        times.append(ctx.timing_read()["focus_ms"] - ms)
    ctx.timing(False)
    got["focus_2048x256"] = {"kernel": "focus_kernel_roll", "g_samples_per_s": n * h * h * 3 / (min(times) * 1e-3) / 1e9}  # GB/s
    # 2c. the wave-cooperative kernel at the reference's training shape, chosen by the library: 8 x 300^2 x 100
    saved = os.environ.pop("REINFOCUS_RENDER_SETS", None)
    try:
        auto = _native.Context(0)
    finally:
        if saved is not None:
            os.environ["REINFOCUS_RENDER_SETS"] = saved
    n, h, spp = 8, 300, 100
    auto.seed(n * h * h, 0, 0)
    auto.set_scene(*helpers.pack_scene(*helpers.random_scene(np.random.default_rng(7), n)))
    auto.render(n, h, h, spp)
    auto.timing(True)
    times = []
    for _ in range(4):
        ms = auto.timing_read()["render_ms"]
        auto.render(n, h, h, spp)
        times.append(auto.timing_read()["render_ms"] - ms)
    got["wave_8x300x100"] = {"kernel": auto.render_kernel_name(), "g_samples_per_s": n * h * h * spp / (min(times) * 1e-3) / 1e9}
    auto.close()
    # 3. the general renderer at 64 x 256^2 x 16: one rectangle (the cooperative one-shape kernel), mixed (the dense kernel)
    from reinfocus_amd.graphics import camera, shape_factory as sf, world

    n, h, spp = 64, 256, 16
    rng = np.random.default_rng(0)
    p, ty, si = world.Worlds(*[sf.one_rect(sf.ShapeParameters(float(d))) for d in rng.uniform(5, 10, n)]).device_data()
    cams = camera.Cameras(*[camera.make_gpu_camera(focus_distance=float(f)) for f in rng.uniform(5, 10, n)]).device_data()
    scenes = {"general_one_rect_64x256x16": (np.ascontiguousarray(cams, dtype=np.float64), (p, ty, si)),
              "general_mixed_64x256x16": _factory_worlds(rng, n, "mixed")}
    for name, (cameras, (params, types, sizes)) in scenes.items():
        ctx.render_general(cameras, params, types, sizes, h, h, spp, to_host=False)
        ctx.timing(True)
        rate = best_of(lambda: ctx.render_general(cameras, params, types, sizes, h, h, spp, to_host=False), n * h * h * spp)
        got[name] = {"kernel": ctx.render_kernel_name(), "g_samples_per_s": rate}
        ctx.timing(False)
    ctx.close()
    return got


def test_the_kernels_keep_their_names_and_their_speed():
    got = _measure()
    if os.environ.get("REINFOCUS_PERF_GUARD_RECORD"):
        with open(os.environ["REINFOCUS_PERF_GUARD_RECORD"], "w") as out:
            json.dump(got, out, indent=1)
    want = json.load(open(RECORD))
    assert set(got) == set(want["kernels"])
    for name, reference in want["kernels"].items():
        assert got[name]["kernel"] == reference["kernel"], (name, got[name]["kernel"])
        floor = FLOOR * reference["g_samples_per_s"]
        assert got[name]["g_samples_per_s"] >= floor, (name, got[name]["g_samples_per_s"], "floor", floor)
