"""The library and torch must share one HIP runtime whatever the import order (loading libofdm_mi355x.so before torch
used to leave torch with `RuntimeError: No HIP GPUs are available`).  Each order runs ONCE in a fresh interpreter.

Round 2 saw the library-first child overrun 300 s once in ~25 suite runs and hid it behind a retry without keeping the child's
output.  Round 3 removed the retry, gave the children stage stamps and a faulthandler watchdog -- and caught it
(profiles/r03_import_order_library_first_STUCK.log): the child sat in `from torch._C import *`, i.e. in the dlopen of torch's
~1 GB of shared objects, whose constructors register their device code with a HIP runtime this library had ALREADY initialised
(unpacked on the spot instead of lazily: 6-8 s normally, 150+ s that time).  `_lib.load()` therefore imports torch itself, before
anything touches the GPU, whenever torch is installed: "library first" is torch first underneath.  The third case keeps the old
order alive on purpose (OFDM_MI355X_NO_TORCH_IMPORT=1) but only checks what can be checked without importing torch afterwards.
There is still NO second attempt: stage stamps, watchdog, the record under gpurun_out/ on every run, and an overrun FAILS."""
import os
import subprocess
import sys

import pytest

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu

WATCHDOG_S, TIMEOUT_S = 150, 200

PRELUDE = """
import faulthandler, os, sys, time
faulthandler.enable()
faulthandler.dump_traceback_later(WATCHDOG_S, exit=True)           # a stuck stage leaves every thread's stack on stderr, then exits 1
_t0 = time.time()
def stage(what):
    print("[%%7.2f s] %%s" %% (time.time() - _t0, what), file=sys.stderr, flush=True)
os.environ["OFDM_MI355X_TRACE_LOAD"] = "1"                 # _lib.load() stamps its own steps (runtime pin, dlopen, symbol check)
sys.path[:0] = [%r, %r]
stage("interpreter up")
""".replace("WATCHDOG_S", str(WATCHDOG_S))

LIB_FIRST = PRELUDE + """
import numpy as np
stage("numpy imported")
import ofdm_mi355x as om
stage("ofdm_mi355x imported")
om.load()
stage("library loaded")
assert "torch" in sys.modules                             # load() imported it: its device code is registered before the runtime starts
rx = om.RxEngine(8, 64, 16, 62, (1, 3), 60, 100)          # touches the GPU through the library before the PROGRAM imports torch
stage("ofdm_rx_create returned (first HIP initialisation of this process)")
buf = om.DeviceBuffer(64).upload(np.arange(16, dtype=np.float32))
stage("device buffer uploaded")
import torch
stage("torch imported")
assert torch.cuda.is_available()
stage("torch.cuda.is_available")
x = torch.arange(8, device="cuda", dtype=torch.float32)
assert float((x * 2).sum().item()) == 56.0
stage("torch kernel ran")
from ofdm_mi355x import _lib
maps = _lib._mapped_hip_runtimes()
assert len(maps) == 1, maps
print("ok", _lib.hip_runtime_path, flush=True)
rx.close(); buf.free()
stage("handles released; leaving the interpreter")
"""

TORCH_FIRST = PRELUDE + """
import torch
stage("torch imported")
torch.cuda.init()
stage("torch.cuda.init returned (first HIP initialisation of this process)")
import numpy as np
import ofdm_mi355x as om
stage("ofdm_mi355x imported")
rx = om.RxEngine(8, 64, 16, 62, (1, 3), 60, 100)
stage("ofdm_rx_create returned")
t = torch.zeros(16, device="cuda")
stage("torch allocation done")
from ofdm_mi355x import _lib
assert len(_lib._mapped_hip_runtimes()) == 1
print("ok", _lib.hip_runtime_path, flush=True)
rx.close()
stage("handle released; leaving the interpreter")
"""


def _keep_record(name, text):
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "import_order_%s.log" % name), "w") as f:
            f.write(text)
    except OSError:
        pass


LIB_ONLY = PRELUDE + """
os.environ["OFDM_MI355X_NO_TORCH_IMPORT"] = "1"
import numpy as np
import ofdm_mi355x as om
om.load()
stage("library loaded without importing torch")
assert "torch" not in sys.modules
rx = om.RxEngine(8, 64, 16, 62, (1, 3), 60, 100)
stage("ofdm_rx_create returned")
from ofdm_mi355x import _lib
assert len(_lib._mapped_hip_runtimes()) == 1
print("ok", _lib.hip_runtime_path, flush=True)
rx.close()
stage("handle released; leaving the interpreter")
"""


@pytest.mark.parametrize("name,code", [("library-first", LIB_FIRST), ("torch-first", TORCH_FIRST), ("library-only", LIB_ONLY)],
                         ids=["library-first", "torch-first", "library-only"])
def test_either_import_order_works(name, code):
    try:
        r = subprocess.run([sys.executable, "-c", code % (ROOT, PKG)], capture_output=True, text=True, timeout=TIMEOUT_S)
    except subprocess.TimeoutExpired as e:           # the watchdog should have fired first; either way the record is the finding
        out = (e.stdout or b"").decode("utf-8", "replace") if isinstance(e.stdout, bytes) else (e.stdout or "")
        err = (e.stderr or b"").decode("utf-8", "replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
        _keep_record(name, "TIMEOUT after %d s\n---- stdout ----\n%s\n---- stderr ----\n%s" % (TIMEOUT_S, out, err))
        pytest.fail("%s child still running after %d s (no retry).  stdout:\n%s\nstderr (stages + stacks):\n%s" % (name, TIMEOUT_S, out[-2000:], err[-6000:]))
    _keep_record(name, "rc %d\n---- stdout ----\n%s\n---- stderr ----\n%s" % (r.returncode, r.stdout, r.stderr))
    assert r.returncode == 0 and "ok" in r.stdout, "rc %d\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-6000:])
    assert "leaving the interpreter" in r.stderr
    assert "torch/lib/libamdhip64" in r.stdout          # torch is installed here: its bundled runtime is the shared one
