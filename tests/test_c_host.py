"""The drop-in boundary is a C ABI: a plain-C program (examples/c_host.c; no Python, no torch) includes include/ofdm_mi355x.h,
links libofdm_mi355x.so and drives TX -> channel -> RX on device buffers."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "examples", "c_host.c")
EXE = os.path.join(ROOT, "examples", "c_host")
LIBDIR = os.path.join(ROOT, "lte-gnu-radio-code_amd", "ofdm_mi355x")


def _build():
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    if not os.path.exists(os.path.join(LIBDIR, "libofdm_mi355x.so")):
        pytest.skip("library not built")
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-std=c11", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE, "-L", LIBDIR,
           "-lofdm_mi355x", "-lm", "-Wl,-rpath," + LIBDIR]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_header_is_plain_c_and_the_library_links():
    """include/ofdm_mi355x.h compiles as C11 with -Wall -Werror and every entry point the example uses resolves at link time."""
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_c_host_round_trip_on_the_gpu():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bit errors 0 /" in r.stdout
