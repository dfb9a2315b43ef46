"""The CPU-baseline child of bench.py on a tiny workload (CPU only): all four legs run, the line carries what SURVEY 8(d) /
BASELINE.md section 3 ask for (host cores, usable cores, >= 3 timed repetitions of warmed workers, whole frames of >= 240 symbols
for the reference-structure leg, the plain-C scalar leg and the same C on every usable core)."""
import json
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT
from oracle import cpu_baseline as cb
from oracle import ofdm_oracle as orc


def test_usable_cores_respects_affinity_and_override(monkeypatch):
    host, use = cb.usable_cores()
    assert 1 <= use <= host == (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert use <= len(os.sched_getaffinity(0))
    monkeypatch.setenv("BENCH_CPU_WORKERS", "3")
    assert cb.usable_cores()[1] == 3
    assert cb.sample_frames_wanted() == max(16, 3 * cb.FRAMES_PER_WORKER)


def test_cpu_baseline_child_runs_all_legs(tmp_path):
    N, cp, Kd, n_sym = 64, 16, 60, 240
    rng = np.random.default_rng(1)
    frames = []
    for _ in range(4):
        bits = rng.integers(0, 2, 180 * Kd * 2).astype(np.uint8)
        tx = orc.tx_modulate(bits, N, cp, N - 2, Kd, n_sym)
        frames.append((tx + 0.01 * (rng.standard_normal(len(tx)) + 1j * rng.standard_normal(len(tx)))).astype(np.complex64))
    path = str(tmp_path / "sample.npy")
    np.save(path, np.stack(frames * 4))                       # 16 frames
    env = dict(os.environ, BENCH_CPU_WORKERS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), path,
                        json.dumps(dict(nfft=N, cp=cp, Kd=Kd, snr_db=30.0)), "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["kind"] == "port" and d["value"] > 0 and d["unit"] == "Msamples/s"
    assert d["host_cores"] == (os.cpu_count() or 1) and d["usable_cores"] == 2 and d["vectorised_cores"] == 2
    assert d["vectorised_reps"] >= 3 and d["vectorised_frames_per_worker"] >= 8 and d["vectorised_value"] > 0
    assert "240 symbols" in d["sample"] and "disjoint" in d["vectorised_sample"]
    assert d["c_scalar_value"] and d["c_scalar_cores"] == 1
    assert d["c_parallel_value"] and d["c_parallel_cores"] == 2 and "OpenMP" in d["c_parallel_sample"]
