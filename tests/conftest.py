import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: wall-clock performance floors on a real MI355X (tests/test_gpu_perf_guard.py; "
                            "run alone with -m perf: a noisy box must not abort the parity suite)")
    # The library renders launches of few blocks with its kernel without cooperative tails (rf_abi_render.hip few_blocks) --
    # which is most of what tests render.  The tests are there for the benchmarked kernel: three pixels per thread at every size, unless a
    # test asks otherwise (tests/test_gpu_parity.py::test_few_blocks_take_one_pixel_per_thread and the notebook outputs
    # run the automatic choice).
    os.environ.setdefault("REINFOCUS_RENDER_SETS", "3")
    # likewise the general renderer's kernel for worlds of one shape: the library takes it for launches that fill the
    # device; the tests render it at every size (test_gpu_general.py::test_small_one_shape_launches_take_the_literal_kernel
    # runs the library's choice)
    os.environ.setdefault("REINFOCUS_GENERAL_ONE", "1")
    # and its dense kernel for worlds of up to three shapes (the library: launches of more than 2 M pixels)
    os.environ.setdefault("REINFOCUS_GENERAL_DENSE", "1")
    # every device / pinned-host allocation of the library starts as 0xA5 bytes (csrc/rf_host.h dev_malloc): a result that
    # depends on what fresh or recycled memory holds -- a list not cleared, a sum not zeroed -- fails here instead of in a
    # long-lived process (smoke() and bench.py run without it)
    os.environ.setdefault("REINFOCUS_POISON_ALLOC", "1")
    # the general renderer proves a lens radius it does not know (60 ms on the host) only once it has come back in 64 calls
    # (rf_abi_ctx.hip lens_exact_if_known); the tests want the float32-lens kernel instances from the first call on
    # (tests/test_gpu_general.py::test_an_unknown_lens_radius_is_not_proven_at_once runs the library's own count)
    os.environ.setdefault("REINFOCUS_LENS_PROVE_AFTER", "1")


def pytest_sessionstart(session):
    """On a host without a GPU (where the sources are edited) every library is brought up to date
    before the tests run -- make is a no-op when nothing changed -- so that what travels to the GPU
    box is what the checkout builds.  The GPU box compiles nothing: it runs the binaries it was sent
    (tests/helpers.py verifies that they are the build of the sources they came with).  A host
    without hipcc still gets the CPU-side libraries (oracle, hostsim); the tests that need the HIP
    library say that it is missing."""
    if os.path.exists("/dev/kfd") or os.environ.get("REINFOCUS_NO_AUTOBUILD"):
        return
    import shutil
    import subprocess
    import warnings

    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    have_hipcc = os.path.exists(hipcc)
    for directory, target, needs_hipcc in (("reinfocus_amd/csrc", "all", True), ("oracle", "all", False),
                                           ("tests/hostsim", "libhostsim.so", False), ("tests/gpucheck", "all", True)):
        if needs_hipcc and not have_hipcc:
            warnings.warn(f"no hipcc on this host: {directory} is not built")
            continue
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, directory), target])


@pytest.fixture(scope="session", autouse=True)
def shipped_libraries_match_their_sources():
    """On the GPU box nothing is compiled: the product library (and the checkers) travelled with the
    checkout.  If any of them is not the build of the sources next to it -- someone edited a header and did
    not run __graft_entry__.build() before shipping -- the whole session fails here, loudly, instead of
    testing something else."""
    if os.path.exists("/dev/kfd"):
        from tests import helpers

        helpers.verify_srchash(os.path.join(ROOT, "reinfocus_amd/libreinfocus_hip.so"), extra="")  # no -D options
        helpers.verify_srchash(os.path.join(ROOT, "oracle/librf_oracle.so"))
        helpers.verify_srchash(os.path.join(ROOT, "oracle/librf_oracle_o3.so"))
    yield


@pytest.fixture(params=["three pixels per thread", "the library's choice", "wave-cooperative"])
def kernel_choice(request, monkeypatch):
    """The render kernels a context created inside the test takes.  The session's default forces the benchmarked kernels at
    every size (pytest_configure above); a test that names this fixture runs three times -- like that; with the
    switches unset, i.e. with the kernels the library itself picks for the launch (for most test sizes: render_kernel
    without cooperative tails, and the dense or literal general kernel), which is what the reference's default use takes
    (state_observer.py:335: one 300 x 300 environment); and with render_kernel_wave<.., 3> forced at every size (rf_wave.h:
    the library's choice for launches of 0.5-2.2 M pixels -- examples/ppo_tuned.yml:5: n_envs 8 -- which few tests are)."""
    if request.param == "the library's choice":
        monkeypatch.delenv("REINFOCUS_RENDER_SETS", raising=False)
        monkeypatch.delenv("REINFOCUS_GENERAL_ONE", raising=False)
        monkeypatch.delenv("REINFOCUS_GENERAL_DENSE", raising=False)
    elif request.param == "wave-cooperative":
        monkeypatch.setenv("REINFOCUS_RENDER_SETS", "w3")
    return request.param


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/), built on demand.  Test infrastructure only."""
    from oracle import oracle as orc

    orc.build()
    return orc


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
