"""The C-ABI shared library: it loads, exports every symbol include/reinfocus_hip.h
declares, and fails loudly (no CPU fallback) when no GPU is present.  No compute calls."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "reinfocus_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rf_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_match_binding():
    from reinfocus_amd import _native

    declared = _declared_symbols()
    assert declared, "no rf_* declarations found in include/reinfocus_hip.h"
    assert sorted(_native.SYMBOLS) == declared


def test_library_exports_every_declared_symbol():
    from reinfocus_amd import _native

    assert os.path.exists(_native.LIB_PATH), "build with __graft_entry__.build() first"
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in the header but not exported"
    assert _native.load().rf_abi_version() == 1


def test_no_gpu_means_loud_failure():
    """On a host without a GPU the product path must raise, not fall back."""
    from reinfocus_amd import _native

    if _native.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        _native.Context(0)
    from reinfocus_amd.graphics import render

    with pytest.raises(RuntimeError):
        render.FastRenderer()
    from reinfocus_amd import vision
    import numpy as np

    with pytest.raises(RuntimeError):
        vision.focus_value(np.zeros((4, 4, 3), dtype=np.uint8))


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "reinfocus_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, os.path.join(dirpath, f)
                assert "librf_oracle" not in text and "rf_oracle.h" not in text, os.path.join(dirpath, f)
                # (the arithmetic headers that tests/hostsim compiles for the CPU say so in comments; rf_general_dense.h has a
                # hook -- nudged approximations -- that only exists when hostsim defines RF_HOSTSIM before including it)
                assert "hostsim" not in text or f in ("rf_math.h", "rf_general_dense.h", "rf_general_chunk.h"), os.path.join(dirpath, f)
                assert "RF_HOSTSIM" not in text or f == "rf_general_dense.h", os.path.join(dirpath, f)


def test_built_libraries_are_not_older_than_their_sources():
    """The .so files are git-ignored but travel to the GPU box: a stale one would be tested there in
    place of what the checkout builds.  `make -q` says whether a rebuild is due (the Makefiles list
    every header)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for directory, target in (("reinfocus_amd/csrc", "all"), ("tests/gpucheck", "all"), ("tests/hostsim", "libhostsim.so"),
                              ("oracle", "all")):
        rc = subprocess.call(["make", "-q", "-C", os.path.join(root, directory), target],
                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert rc == 0, f"{directory}: {target} is out of date -- run __graft_entry__.build()"


def test_every_entry_point_that_touches_the_device_selects_it():
    """One process may hold contexts on several GPUs (harness.ShardedVectorDiscreteSteps: a thread per device): an entry
    point that allocates, copies or launches must make the ctx's device current first, whatever the calling thread used
    last.  Source-level: every extern "C" definition of csrc/rf_abi_*.hip calls hipSetDevice itself or consists of calls
    to entry points that do; the ones listed here touch no device."""
    hostside = {"rf_device_count", "rf_device_info", "rf_num_states", "rf_env_last_step_branch", "rf_env_step_abort",
                "rf_env_scene_len", "rf_last_error", "rf_pixels_rendered", "rf_allocations_poisoned", "rf_render_kernel_name",
                "rf_general_redo_pixels", "rf_step", "rf_abi_version"}  # (rf_step = rf_render + rf_focus)
    seen = set()
    csrc = os.path.join(ROOT, "reinfocus_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if not (name.startswith("rf_abi_") and name.endswith(".hip")):
            continue
        text = open(os.path.join(csrc, name)).read()
        for m in re.finditer(r"^(?:int|unsigned|unsigned long long|const char \*)\s*(rf_[a-z_0-9]+)\(", text, flags=re.M):
            entry = m.group(1)
            start = text.index("{", m.end())
            depth, end = 0, start
            while True:  # (the definition's closing brace)
                depth += {"{": 1, "}": -1}.get(text[end], 0)
                end += 1
                if depth == 0:
                    break
            seen.add(entry)
            if entry not in hostside:
                assert "hipSetDevice(" in text[start:end], f"{name}: {entry} does not select its device"
    assert seen == set(_declared_symbols()), sorted(seen ^ set(_declared_symbols()))
