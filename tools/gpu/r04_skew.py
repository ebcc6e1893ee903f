"""Round 4, one-off: the round-3 form of the sphere call (no barrier after the collect, one counter; -DRF_TEST_UNFENCED) under
RF_TEST_SKEW's delayed waves, next to the shipped form under the same delays: pixels that differ from the product build's frames.
usage (GPU box): python tools/gpu/r04_skew.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import helpers  # noqa: E402

SCRIPT = """
import sys; sys.path.insert(0, %r)
import numpy as np
from reinfocus_amd import _native
d = np.load(%r)
c = _native.Context(0); c.seed(%d, 0, 0)
c.set_scene(d['dyn'], d['rect'], d['origin'], d['u'], d['v'], float(d['lens']))
f = c.render(%d, %d, %d, %d, to_host=True); s = c.get_states()
np.savez(%r, frames=f, states=s); c.close()
"""


def render(lib, d, n, h, spp, work):
    out = os.path.join(work, "out.npz")
    scene = os.path.join(work, "scene.npz")
    np.savez(scene, dyn=d[0], rect=d[1], origin=d[2], u=d[3], v=d[4], lens=d[5])
    env = dict(os.environ)
    if lib:
        env["REINFOCUS_HIP_LIB"] = os.path.join(ROOT, lib)
    subprocess.check_call([sys.executable, "-c", SCRIPT % (ROOT, scene, n * h * h, n, h, h, spp, out)], env=env)
    got = np.load(out)
    return got["frames"], got["states"]


def main():
    n, spp = 8, 8
    rng = np.random.default_rng(4)
    d = helpers.pack_scene(*helpers.random_scene(rng, n))
    with tempfile.TemporaryDirectory() as work:
        for h in (128, 256):
            want, want_states = render(None, d, n, h, spp, work)
            for label, lib in (("shipped form, delayed waves", "tests/gpucheck/libreinfocus_skew.so"),
                               ("round-3 form (no B4, one counter), delayed waves", "tools/lib_skew_unfenced.so")):
                try:
                    got, states = render(lib, d, n, h, spp, work)
                except subprocess.CalledProcessError as error:
                    print(f"{h} px, {label}: the render FAILED ({error})", flush=True)
                    continue
                wrong = int((got != want).any(axis=3).sum())
                print(f"{h} px, {label}: {wrong} of {n * h * h} pixels differ, "
                      f"{int((states != want_states).any(axis=1).sum())} RNG states differ", flush=True)


if __name__ == "__main__":
    main()
