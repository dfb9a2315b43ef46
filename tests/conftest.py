import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "lte-gnu-radio-code_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The torch wheel carries its own HIP runtime next to the system one libofdm_mi355x.so links against.  Both coexist as long
    # as torch initialises first (bench.py does); a torch.cuda call made AFTER the library has touched the GPU fails with "No HIP
    # GPUs are available".  Tests that use torch for device memory therefore need it initialised before any other test runs.
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def relerr(a, b):
    """max|a-b| / max|b| : the norm-relative error every tolerance in this suite is stated in."""
    a = np.asarray(a)
    b = np.asarray(b)
    d = np.max(np.abs(a - b)) if a.size else 0.0
    s = np.max(np.abs(b)) if b.size else 1.0
    return float(d / s) if s > 0 else float(d)
