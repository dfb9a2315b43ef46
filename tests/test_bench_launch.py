"""bench.py's launcher (no GPU): `python bench.py --gpus N` without WORLD_SIZE must start N ranks itself (the driver's command
form) and relay rank 0's JSON line; a --gpus / WORLD_SIZE disagreement and a node with too few GPUs must fail loudly.
--dry-launch keeps the ranks on CPU tensors over gloo with a fake produce step (control flow only)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MASTER_ADDR"] = "127.0.0.1"
    return env


def _run(*args, env=None, timeout=300):
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=timeout, env=env or _env())


@pytest.mark.parametrize("cfg,gather", [("cfg2", "auto"), ("cfg4", "auto"), ("cfg2", "direct"), ("cfg4", "direct")])
def test_gpus_2_spawns_two_ranks(cfg, gather):
    r = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-launch", "--config", cfg, "--gather", gather)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world_size"] == 2
    assert d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert "dry-launch" in d["data"]
    if cfg == "cfg4":
        assert d["config"]["sub_batches_per_step"] == 8


def test_single_rank_dry_launch_needs_no_launcher():
    r = _run("--steps", "2", "--warmup", "0", "--dry-launch")
    assert r.returncode == 0, r.stderr[-3000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_world_size_disagreement_fails_loudly():
    env = _env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = _run("--gpus", "4", "--dry-launch", env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_too_few_gpus_fails_before_any_rank_starts():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("node has >= 2 GPUs")
    env = _env()
    env.pop("BENCH_REHEARSAL", None)
    r = _run("--gpus", "2", "--steps", "1", env=env)
    assert r.returncode != 0
    assert "exposes" in (r.stderr + r.stdout) and "{" not in r.stdout


def test_cfg4_is_baseline_config_3():
    import bench
    c = bench.CONFIGS["cfg4"]
    assert (c["nfft"], c["cp"], c["Kd"], c["mod"]) == (2048, 144, 1200, "64QAM")
    # 8 batches of ~1 Mi symbols per GPU; 8 GPUs -> 64 Mi symbols
    assert c["batches"] == 8 and abs(c["batches"] * c["frames"] * c["n_sym"] * 8 - 64 * 2 ** 20) < 64 * 240 * 8
