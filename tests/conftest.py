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
