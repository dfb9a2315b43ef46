"""The N > 1 step loop of bench.py on the real backend (run with -m gpu).

RCCL refuses two ranks on one device, so a one-GPU box cannot run two ranks -- but a one-rank RCCL group executes every call of
that path: group creation with a bound device, asynchronous `all_gather_into_tensor` on the launch stream with deferred waits
(`GatherPipeline`), the re-assembly check, the int64 `all_reduce` of the counts-only leg, barriers.  `BENCH_FORCE_DIST=1` makes
bench.py do exactly that; the line it prints is not a measurement."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("extra", [[], ["--config", "cfg4"]], ids=["cfg2", "cfg4"])
def test_one_rank_rccl_group_runs_the_multi_gpu_path(extra):
    env = dict(os.environ, BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "128", "--steps", "3", "--warmup", "1", "--no-cpu",
                        "--no-probes"] + extra, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # RCCL's version banner must not land on stdout next to the JSON line
    d = json.loads(lines[0])
    g = d["config"]["all_gather"]
    assert d["n_gpus"] == 1 and g["backend"] == "nccl" and g["world_size"] == 1
    assert d["config"]["bit_error_rate_frame0"] == 0.0
    assert g["counts_only"]["bit_errors_all_ranks"] == 0 and g["counts_only"]["bits_all_ranks"] > 0
