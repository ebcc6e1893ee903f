"""TEST INFRASTRUCTURE (not collected by pytest): randomised parity soak of the general renderer
(rf_render_general) against the CPU oracle: frames and final RNG states, bit for bit.
usage (GPU box, repo root): python tests/soak/soak_general.py [scenes] [seed]
Every scene is rendered by three contexts: the library's own choice of kernel (scenes this small: the literal kernel), the
dense / one-shape kernels forced (REINFOCUS_GENERAL_DENSE=1, REINFOCUS_GENERAL_ONE=1), and the same with a fix-up list of
48 entries (REINFOCUS_GENERAL_REDO_CAP: the overflow path re-renders the launch literally); the run fails unless the dense,
the one-shape and the literal kernel have all been seen."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle  # noqa: E402
from reinfocus_amd import _native  # noqa: E402
from tests.test_general_renderer import _few_shape_worlds, _random_scene  # noqa: E402


def main():
    scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    oracle.build()
    for name in ("REINFOCUS_GENERAL_DENSE", "REINFOCUS_GENERAL_ONE", "REINFOCUS_GENERAL_REDO_CAP"):
        os.environ.pop(name, None)
    contexts = {"choice": _native.Context(0)}  # (the switches are read when a context is created)
    os.environ["REINFOCUS_GENERAL_DENSE"] = os.environ["REINFOCUS_GENERAL_ONE"] = "1"
    contexts["forced"] = _native.Context(0)
    t0 = time.time()
    pixels = 0
    kernels = {}
    overflowed = 0
    for scene in range(scenes):
        rng = np.random.default_rng([seed, scene])
        n = int(rng.integers(1, 10))
        h, w = int(rng.integers(8, 97)), int(rng.integers(8, 129))
        spp = int(rng.integers(1, 13))
        # every third scene: ragged worlds of one to three shapes under tilted cameras (the literal kernel); the others:
        # one or two shapes in every environment under canonical axes (the dense kernel, or the one-shape kernel)
        if scene % 3 == 0:
            cameras, (params, types, sizes) = _random_scene(rng, n)
        else:
            cameras, (params, types, sizes) = _few_shape_worlds(rng, n, 1 + scene % 2)
        states = oracle.seed_states(n * h * w, 0)
        want = oracle.render_general(cameras, params, types, sizes, h, w, spp, states, n_threads=16)
        for label in ("choice", "forced", "overflow"):
            ctx = contexts["forced" if label == "overflow" else label]
            if label == "overflow":  # (the list's capacity is read per call)
                os.environ["REINFOCUS_GENERAL_REDO_CAP"] = "48"
            try:
                got = ctx.render_general(cameras, params, types, sizes, h, w, spp)
            finally:
                os.environ.pop("REINFOCUS_GENERAL_REDO_CAP", None)
            assert np.array_equal(got, want), (scene, label, n, h, w, spp, int(np.any(got != want, axis=-1).sum()))
            assert np.array_equal(ctx.get_states(0, n * h * w), states), (scene, label, "states")
            pixels += n * h * w
            key = label + ":" + ctx.render_kernel_name().split("<")[0]
            kernels[key] = kernels.get(key, 0) + 1
            if label == "overflow":
                overflowed += int(ctx.general_redo_pixels() > 48)
        if scene % 20 == 19:
            print(f"scene {scene}: ok ({pixels} pixels so far, {time.time() - t0:.0f} s)", flush=True)
    for ctx in contexts.values():
        ctx.close()
    seen = {key.split(":")[1] for key in kernels}
    assert {"render_general_kernel", "render_general_dense_kernel", "render_general_one_kernel"} <= seen, kernels
    assert scenes < 30 or overflowed > 0, "no launch overflowed its 48-entry list"
    print(f"general soak ok: {scenes} scenes x 3 contexts, {pixels} pixels, kernels {kernels}, {overflowed} launches re-rendered "
          "after overflowing a 48-entry fix-up list")


if __name__ == "__main__":
    main()
