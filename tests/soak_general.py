"""TEST INFRASTRUCTURE (not collected by pytest): randomised parity soak of the general renderer
(rf_render_general) against the CPU oracle: frames and final RNG states, bit for bit.
usage (GPU box, repo root): python tests/soak_general.py [scenes] [seed]"""
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
    ctx = _native.Context(0)
    t0 = time.time()
    pixels = 0
    kernels = {}
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
        got = ctx.render_general(cameras, params, types, sizes, h, w, spp)
        assert np.array_equal(got, want), (scene, n, h, w, spp, int(np.any(got != want, axis=-1).sum()))
        assert np.array_equal(ctx.get_states(0, n * h * w), states), (scene, "states")
        pixels += n * h * w
        kernels[ctx.render_kernel_name().split("<")[0]] = kernels.get(ctx.render_kernel_name().split("<")[0], 0) + 1
        if scene % 20 == 19:
            print(f"scene {scene}: ok ({pixels} pixels so far, {time.time() - t0:.0f} s)", flush=True)
    ctx.close()
    print(f"general soak ok: {scenes} scenes, {pixels} pixels, kernels {kernels}")


if __name__ == "__main__":
    main()
