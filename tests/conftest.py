import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """On a host without a GPU (where the sources are edited) every library is brought up to date
    before the tests run -- make is a no-op when nothing changed -- so that what travels to the GPU
    box is what the checkout builds.  The GPU box runs the binaries it was sent."""
    if os.path.exists("/dev/kfd") or os.environ.get("REINFOCUS_NO_AUTOBUILD"):
        return
    import subprocess

    for directory, target in (("reinfocus_amd/csrc", "all"), ("oracle", "librf_oracle.so"),
                              ("tests/hostsim", "libhostsim.so"), ("tests/gpucheck", "all")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, directory), target])


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/), built on demand.  Test infrastructure only."""
    from oracle import oracle as orc

    orc.build()
    return orc


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
