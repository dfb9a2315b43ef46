"""csrc/fft_core.hpp executed lane-by-lane on the CPU (host compilation by hipcc, no GPU):
pins the radix passes, twiddles and padded LDS exchange layouts against numpy.fft."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT, relerr

SRC = os.path.join(ROOT, "tests", "host", "fft_core_host_test.cpp")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("fftcore") / "fft_core_host_test")
    subprocess.run(["hipcc", "-O2", "-std=c++17", "-fno-slp-vectorize", "-o", out, SRC], check=True)
    return out


@pytest.mark.parametrize("n", [64, 128, 256, 512, 1024, 2048, 4096])
def test_fft_core_matches_numpy(exe, tmp_path, n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    x.tofile(fin)
    assert subprocess.run([exe, str(n), fin, fout]).returncode == 0
    got = np.fromfile(fout, dtype=np.complex64)
    ref = np.fft.fft(x.astype(np.complex128))
    assert relerr(got, ref) < 1e-6     # fp32 FFT; north-star tolerance is 1e-5
    # impulse -> flat spectrum, exact
    e = np.zeros(n, np.complex64)
    e[1] = 1
    e.tofile(fin)
    subprocess.run([exe, str(n), fin, fout], check=True)
    got = np.fromfile(fout, dtype=np.complex64)
    assert relerr(got, np.exp(-2j * np.pi * np.arange(n) / n)) < 5e-7
