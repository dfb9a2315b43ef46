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


@pytest.fixture(scope="module")
def exe_swz8(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("fftcore8") / "fft_core_host_test")
    subprocess.run(["hipcc", "-O2", "-std=c++17", "-fno-slp-vectorize", "-DOFDM_B_SWZ_RL8=1", "-o", out, SRC], check=True)
    return out


def test_fft_core_with_rotated_rows_of_8_matches_numpy(exe_swz8, tmp_path):
    n = 2048
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    x.tofile(fin)
    assert subprocess.run([exe_swz8, str(n), fin, fout]).returncode == 0
    assert relerr(np.fromfile(fout, dtype=np.complex64), np.fft.fft(x.astype(np.complex128))) < 1e-6


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


def _b_index(N, c, n, swz8=False):
    """csrc/fft_core.hpp: b_index<N> (restated): rows of RL elements, padded to RL+1 or -- 4096-pt, and 2048-pt when built with
    OFDM_B_SWZ_RL8 -- unpadded with the column rotated by the row."""
    rl = N // 256
    if rl == 16:
        return c * rl + ((n + c + (c >> 4)) & (rl - 1))
    if rl == 8 and swz8:
        return c * 8 + ((n + (c >> 1) + (c >> 4)) & 7)
    return c * (rl + 1) + n


@pytest.mark.parametrize("n,swz8", [(512, False), (1024, False), (2048, False), (2048, True), (4096, False)])
def test_exchange_b_layout_is_bank_conflict_free(n, swz8):
    """The gfx950 rules the layouts are built for (MI355X_MICROARCH.md, LDS): ds_write_b64 is processed in 16-lane groups over
    32 eight-byte slots (128 B), ds_read_b64 in 32-lane groups over 256 B.  Pass 1 writes element (row k1*16+k0, column n2) from
    lane n2*16+k0; the last pass reads (row c = lane + T*j, column n).  Every group must hit distinct slots, and the map must be
    a bijection onto [0, LDS_B)."""
    T, rl = n // 16, n // 256
    nc = n // rl
    used = set()
    for k1 in range(16):
        for g in range(T // 16):                               # a 16-lane group: fixed n2 = g, k0 = 0..15
            slots = {_b_index(n, k1 * 16 + k0, g, swz8) % 16 for k0 in range(16)}
            assert len(slots) == 16, (n, "write", k1, g)
        used |= {_b_index(n, k1 * 16 + k0, n2, swz8) for k0 in range(16) for n2 in range(rl)}
    lds_b = nc * rl if (rl == 16 or swz8) else nc * (rl + 1)
    assert len(used) == 256 * rl and max(used) < lds_b
    C = 16 // rl
    for j in range(C):
        for col in range(rl):
            for g in range(T // 32):                           # a 32-lane group of consecutive lanes
                slots = {_b_index(n, (32 * g + r) + T * j, col, swz8) % 32 for r in range(32)}
                assert len(slots) == 32, (n, "read", j, col, g)
