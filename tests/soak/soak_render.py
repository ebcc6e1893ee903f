"""TEST INFRASTRUCTURE (not collected by pytest): randomised parity soak of the render + focus path
against the CPU oracle.  usage (GPU box, repo root): python tests/soak/soak_render.py [cases] [seed]
REINFOCUS_RENDER_SETS=3 / w3 / w2 / w1 / 1 in the environment forces one render kernel for every case (default: the
library's choice by launch size); the kernels that rendered are counted and printed."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle  # noqa: E402
from reinfocus_amd import _native  # noqa: E402
from tests import helpers  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    oracle.build()
    ctx = _native.Context(0)
    t0 = time.time()
    kernels = {}
    for case in range(cases):
        h = int(rng.choice([16, 31, 64, 96, 100, 128, 130, 200, 256, 300]))
        w = h if rng.random() < 0.7 else int(rng.integers(1, 400))
        n = int(rng.integers(1, 6))
        spp = int(rng.integers(1, 24))
        r_size = float(rng.choice([5, 20, 30, 45, 60]))
        lo, hi = (5.0, 10.0) if rng.random() < 0.7 else (0.5, 40.0)
        targets = rng.uniform(lo, hi, n).astype(np.float32)
        focus = rng.uniform(lo, hi, n).astype(np.float32)
        d = helpers.pack_scene(targets, focus, r_size)
        passes = int(rng.integers(1, 4))
        st = oracle.seed_states(n * h * w, 0)
        ctx.seed(n * h * w, 0, 0)
        ctx.set_scene(*d)
        for p in range(passes):
            want = oracle.render(d[0], d[1], h, w, spp, st, n_threads=8)
            got = ctx.render(n, h, w, spp, to_host=True)
            name = ctx.render_kernel_name().split("<")[0]
            kernels[name] = kernels.get(name, 0) + 1
            assert np.array_equal(got, want), (case, p, n, h, w, spp, r_size)
            fv = ctx.focus(n, h, w)
            ref = np.array(oracle.focus_values(want))
            assert np.allclose(fv, ref, rtol=1e-12, atol=0), (case, p, "focus")
        assert np.array_equal(ctx.get_states(0, n * h * w), st), (case, "states")
        print(f"case {case}: n={n} {h}x{w} spp={spp} r_size={r_size} passes={passes} ok ({time.time() - t0:.0f} s)",
              flush=True)
    ctx.close()
    print(f"soak ok: {cases} cases, kernels {kernels}")


if __name__ == "__main__":
    main()
