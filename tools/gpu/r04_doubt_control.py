"""Round 4, one-off: does tests/test_gpu_general.py::test_every_abstention_of_the_one_shape_kernel_is_repaired see a lost
abstention?  The RF_TEST_DOUBT build of a kernel whose checker abstention is dropped on the way (the bug the first version of
rf_general_one.h had) against the oracle.  usage (GPU box): python tools/gpu/r04_doubt_control.py <library>"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402  (tools/gpu one-off: a checker, like the tests)
from tests.test_gpu_general import _random_one_shape_worlds  # noqa: E402

SCRIPT = """
import sys; sys.path.insert(0, %r)
import numpy as np
from reinfocus_amd import _native
d = np.load(%r)
c = _native.Context(0)
f = c.render_general(d['cameras'], d['params'], d['types'], d['sizes'], %d, %d, %d)
np.savez(%r, frames=f, redo=c.general_redo_pixels()); c.close()
"""


def main(lib):
    rng = np.random.default_rng(21)
    n, h, w, spp = 4, 64, 64, 6
    cameras, (params, types, sizes) = _random_one_shape_worlds(rng, n)
    with tempfile.TemporaryDirectory() as work:
        scene, out = os.path.join(work, "scene.npz"), os.path.join(work, "out.npz")
        np.savez(scene, cameras=cameras, params=params, types=types, sizes=sizes)
        subprocess.check_call([sys.executable, "-c", SCRIPT % (ROOT, scene, h, w, spp, out)],
                              env=dict(os.environ, REINFOCUS_HIP_LIB=os.path.join(ROOT, lib)))
        got = np.load(out)
    st = oracle.seed_states(n * h * w, 0)
    want = oracle.render_general(cameras, params, types, sizes, h, w, spp, st, n_threads=16)
    wrong = int((got["frames"] != want).any(axis=-1).sum())
    print(f"{lib}: {int(got['redo'])} of {n * h * w} pixels listed, {wrong} pixels differ from the oracle")


if __name__ == "__main__":
    main(sys.argv[1])
