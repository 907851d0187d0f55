import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ilqr-admm_amd"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def pytest_sessionstart(session):
    """Bring the two native libraries up to date before any test loads them (a no-op when they are current): a stale
    liboracle_isls.so / libisls_hip.so against a changed include/isls_hip.h would read argument blocks with the wrong layout.
    hipcc cross-compiles gfx950 without a GPU; the same `make` is what __graft_entry__.build() runs."""
    import shutil
    import subprocess
    for d, jobs in ((os.path.join(ROOT, "ilqr-admm_amd", "csrc"), "4"), (os.path.join(ROOT, "oracle"), "1")):
        if shutil.which("make") is None:
            return
        if subprocess.call(["make", "-q", "-C", d], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) != 0:
            subprocess.check_call(["make", "-C", d, "-j", jobs], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    kern, lib = orc.load()
    return kern


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load
