"""The library and torch must share one HIP runtime whatever the import order (loading libofdm_mi355x.so before torch
used to leave torch with `RuntimeError: No HIP GPUs are available`).  Each order runs in a fresh interpreter."""
import os
import subprocess
import sys

import pytest

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu

LIB_FIRST = """
import sys
sys.path[:0] = [%r, %r]
import numpy as np
import ofdm_mi355x as om
rx = om.RxEngine(8, 64, 16, 62, (1, 3), 60, 100)          # touches the GPU through the library BEFORE torch is imported
buf = om.DeviceBuffer(64).upload(np.arange(16, dtype=np.float32))
import torch
assert torch.cuda.is_available()
x = torch.arange(8, device="cuda", dtype=torch.float32)
assert float((x * 2).sum().item()) == 56.0
from ofdm_mi355x import _lib
maps = _lib._mapped_hip_runtimes()
assert len(maps) == 1, maps
print("ok", _lib.hip_runtime_path)
"""

TORCH_FIRST = """
import sys
sys.path[:0] = [%r, %r]
import torch
torch.cuda.init()
import numpy as np
import ofdm_mi355x as om
rx = om.RxEngine(8, 64, 16, 62, (1, 3), 60, 100)
t = torch.zeros(16, device="cuda")
from ofdm_mi355x import _lib
assert len(_lib._mapped_hip_runtimes()) == 1
print("ok", _lib.hip_runtime_path)
"""


@pytest.mark.parametrize("code", [LIB_FIRST, TORCH_FIRST], ids=["library-first", "torch-first"])
def test_either_import_order_works(code):
    # One fresh interpreter per order.  Seen once in ~25 suite runs on the GPU pool: the library-first child did not come back
    # within 300 s on a fresh box (never reproduced; the same child takes 6 s).  A child that overruns 120 s is killed and the
    # order is tried once more, so that a stuck box start-up does not read as an import-order failure; a second overrun fails.
    for attempt in (1, 2):
        try:
            r = subprocess.run([sys.executable, "-c", code % (ROOT, PKG)], capture_output=True, text=True, timeout=120)
            break
        except subprocess.TimeoutExpired:
            if attempt == 2:
                raise
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    assert "torch/lib/libamdhip64" in r.stdout          # torch is installed here: its bundled runtime is the shared one
