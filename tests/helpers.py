"""Shared helpers for the parity tests (test infrastructure)."""

import hashlib
import os
import subprocess

import numpy as np

from reinfocus_amd.graphics import camera, world

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def on_gpu_box():
    return os.path.exists("/dev/kfd")


def verify_srchash(library, extra=None):
    """The Makefiles write <library>.srchash (sha256 of every prerequisite, paths relative to the
    Makefile's directory; csrc/Makefile adds an "extra: <-D options>" line) when they link: the library
    was built from the sources next to it iff every recorded hash still matches.  `extra`: the -D options
    the library has to have been built with ("" for the product: a build with timing-experiment or test
    options must never pass for it)."""
    stamp = library + ".srchash"
    assert os.path.exists(library), f"{library} is missing: run __graft_entry__.build() where the sources are edited"
    assert os.path.exists(stamp), f"{stamp} is missing: {library} was not built by its Makefile"
    # the stamp of a library built with OUT=<elsewhere> still names paths relative to the Makefile that built it
    bases = [os.path.dirname(library), os.path.join(ROOT, "reinfocus_amd", "csrc")]
    text = [line for line in open(stamp).read().splitlines() if line.strip()]
    recorded_extra = [line[len("extra:"):].strip() for line in text if line.startswith("extra:")]
    if extra is not None:
        assert recorded_extra == [extra], f"{library} was built with options {recorded_extra}, wanted [{extra!r}]"
    lines = [line.split() for line in text if not line.startswith(("extra:", "flags:"))]
    assert lines, f"{stamp} is empty"
    for digest, path in lines:
        source = next((os.path.join(b, path) for b in bases if os.path.exists(os.path.join(b, path))), None)
        assert source is not None, f"{stamp}: {path} does not exist any more"
        now = hashlib.sha256(open(source, "rb").read()).hexdigest()
        assert now == digest, f"{library} is stale: {path} changed since it was built (run __graft_entry__.build())"


def built(directory, target, make_target=None):
    """Path of a native library of this repository.  Where the sources are edited (no GPU) it is brought
    up to date with make first; on the GPU box NOTHING is compiled -- the library travelled with the
    checkout -- and it must be the build of the sources that came with it (verify_srchash), otherwise
    the test fails loudly instead of testing something else."""
    path = os.path.join(ROOT, directory, target)
    if not on_gpu_box() and not os.environ.get("REINFOCUS_NO_AUTOBUILD"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, directory), make_target or target])
    verify_srchash(path)
    return path


def pack_scene(targets, focus_planes, r_size=20):
    """Packs a scene exactly as FastRenderer would (host logic under test elsewhere)."""
    cams = camera.FastCameras()
    cams.update(focus_planes)
    worlds = world.FastWorlds(r_size)
    worlds.update(targets)
    dyn, origin, u, v, lens = cams.device_data()
    return dyn, worlds.device_data(), origin, u, v, float(lens)


def random_scene(rng, n, lo=5.0, hi=10.0):
    targets = rng.uniform(lo, hi, n).astype(np.float32)
    focus = rng.uniform(lo, hi, n).astype(np.float32)
    return targets, focus
