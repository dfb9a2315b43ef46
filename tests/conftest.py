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


def rms_relerr(a, b):
    """||a-b||_2 / ||b||_2"""
    a = np.asarray(a).ravel()
    b = np.asarray(b).ravel()
    den = float(np.sqrt(np.mean(np.abs(b) ** 2))) if b.size else 1.0
    num = float(np.sqrt(np.mean(np.abs(a - b) ** 2))) if b.size else 0.0
    return num / den if den > 0 else num


def elem_relerr(a, b, floor=1e-2):
    """max over the elements with |b| > floor*max|b| of |a-b|/|b|: small-magnitude bins (fades) judged relative to themselves."""
    a = np.asarray(a).ravel()
    b = np.asarray(b).ravel()
    if not b.size:
        return 0.0
    m = np.abs(b) > floor * np.max(np.abs(b))
    return float(np.max(np.abs(a[m] - b[m]) / np.abs(b[m]))) if m.any() else 0.0


# Tolerances of the fp32 GPU outputs against the fp64 oracle (north_star: "within 1e-5 relative fp32"):
TOL_MAX = 1e-5      # max|a-b| / max|b|
TOL_RMS = 1e-5      # ||a-b|| / ||b||
TOL_ELEM = 5e-4     # per element, for elements above 1 % of the largest (an fp32 FFT error of ~1e-6 of the peak is 1e-4 of such a bin)


def assert_close(a, b, what=""):
    e1, e2, e3 = relerr(a, b), rms_relerr(a, b), elem_relerr(a, b)
    assert e1 < TOL_MAX and e2 < TOL_RMS and e3 < TOL_ELEM, "%s: max-norm %.3g (<%g), rms %.3g (<%g), element-wise %.3g (<%g)" % (
        what, e1, TOL_MAX, e2, TOL_RMS, e3, TOL_ELEM)
    return e1, e2, e3


def poisoned(om, nbytes, device=0):
    """Output buffer pre-filled with 0xFF (float NaN pattern / bit value 255): a row the kernel does not WRITE shows up as
    poison instead of passing as whatever a fresh allocation happens to hold."""
    nbytes = max(8, int(nbytes))
    return om.DeviceBuffer(nbytes, device).upload(np.full(nbytes, 0xFF, np.uint8))
