"""TEST INFRASTRUCTURE (not collected by pytest): randomised parity soak of the focus measure (rf_focus: focus_kernel_roll for widths
that are multiples of 4, the byte-per-thread kernel over column tiles for the others) against the CPU oracle on frames of any size --
narrow, wide (up to 4400 columns), tall, one row, one column; both gray modes; every band height of the rolling kernel.
usage (GPU box, repo root): python tests/soak/soak_focus.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle  # noqa: E402
from reinfocus_amd import _native  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    oracle.build()
    contexts = {}
    for band in ("", "8", "16", "32", "64", "3"):  # (the switches are read when a context is created)
        os.environ.pop("REINFOCUS_FOCUS_BAND", None)
        if band:
            os.environ["REINFOCUS_FOCUS_BAND"] = band
        contexts[band or "auto"] = _native.Context(0)
    os.environ.pop("REINFOCUS_FOCUS_BAND", None)
    os.environ["REINFOCUS_FOCUS_KERNEL"] = "byte"
    contexts["byte"] = _native.Context(0)
    os.environ.pop("REINFOCUS_FOCUS_KERNEL", None)
    t0 = time.time()
    pixels = 0
    for case in range(cases):
        kind = rng.random()
        if kind < 0.3:
            h, w = int(rng.integers(1, 40)), int(rng.integers(1, 4400))
        elif kind < 0.6:
            h, w = int(rng.integers(1, 700)), 4 * int(rng.integers(2, 160))
        else:
            h, w = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        n = int(rng.integers(1, max(2, min(6, 2_000_000 // (h * w) + 1))))
        gray = int(rng.choice([15, 14]))
        frames = rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
        if rng.random() < 0.5:  # structure for the median / Laplacian: a smooth ramp with some noise
            yy, xx = np.mgrid[0:h, 0:w]
            frames[0] = ((xx * int(rng.integers(1, 9)) + yy * int(rng.integers(1, 9))) % 256)[..., None] + rng.integers(0, 3, (h, w, 3))
        want = oracle.focus_values(frames, gray, 8)
        first = None
        for name, ctx in contexts.items():
            ctx.upload_frames(frames)
            got = ctx.focus(n, h, w, gray)
            assert np.allclose(got, want, rtol=1e-12, atol=1e-12), (case, name, n, h, w, gray, got, want)
            if first is None:
                first = got
            assert np.array_equal(got, first), (case, name, "kernels disagree", n, h, w)
        pixels += n * h * w
        if case % 20 == 19:
            print(f"case {case}: ok ({pixels} pixels so far, {time.time() - t0:.0f} s)", flush=True)
    for ctx in contexts.values():
        ctx.close()
    print(f"focus soak ok: {cases} cases x {len(contexts)} contexts, {pixels} pixels")


if __name__ == "__main__":
    main()
