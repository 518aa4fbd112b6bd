import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _library_is_built():
    """The shared library is git-ignored: a fresh checkout has none until __graft_entry__.build() (or
    make) has run.  Build it here if it is missing, so that the suite does not depend on the order in
    which a driver runs `build` and `pytest`."""
    so = os.path.join(ROOT, "cusmc_amd", "libcusmc_hip.so")
    if not os.path.exists(so) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cusmc_amd", "csrc"), "-j8"])


@pytest.fixture(scope="session")
def golden():
    """tests/golden/golden.npz -- see tests/golden/make_golden.py for where each value comes from."""
    path = os.path.join(ROOT, "tests", "golden", "golden.npz")
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


DIMS = [1, 2, 3, 8, 32, 64, 65, 256]
RESAMPLE_CASES = ["equal", "zeros", "dominant", "negative", "tiny", "with_nan"]


def spd(rng, d):
    A = rng.standard_normal((d, d))
    return A @ A.T / d + np.eye(d)
