"""The C-ABI library loads without a GPU and exports every function include/ofdm_mi355x.h declares,
the ctypes prototype table covers exactly that set, and compute calls fail loudly (no CPU fallback)."""
import os
import re

import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "ofdm_mi355x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(ofdm_[a-z0-9_]+)\s*\(", txt))


def test_library_exports_the_whole_header():
    import ofdm_mi355x
    from ofdm_mi355x import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built (run __graft_entry__.build())")
    lib = ofdm_mi355x.load()
    declared = _declared()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), name
    assert set(_lib.PROTOTYPES) == declared
    assert lib.ofdm_abi_version() == 1


def test_c_frame_partition_equals_the_python_rule():
    """ofdm_shard_frames (hosts that are not Python) against ofdm_mi355x.dist.shard_frames: host arithmetic, no device."""
    import ctypes as C
    import ofdm_mi355x
    from ofdm_mi355x import _lib, dist
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    lib = ofdm_mi355x.load()
    first, count = C.c_int64(-1), C.c_int64(-1)
    for total, world in ((4369 * 8, 8), (12, 4), (7, 1), (0, 3), (34952, 2)):
        for rank in range(world):
            assert lib.ofdm_shard_frames(total, world, rank, C.byref(first), C.byref(count)) == 0
            assert (first.value, count.value) == dist.shard_frames(total, world, rank)
    assert lib.ofdm_shard_frames(10, 4, 0, C.byref(first), C.byref(count)) != 0          # ValueError in Python
    with pytest.raises(ValueError):
        dist.shard_frames(10, 4, 0)
    assert lib.ofdm_shard_frames(8, 4, 4, C.byref(first), C.byref(count)) != 0
    assert lib.ofdm_shard_frames(8, 4, 0, None, C.byref(count)) != 0


def test_no_silent_cpu_fallback():
    import torch
    import ofdm_mi355x
    from ofdm_mi355x import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ofdm_mi355x.OfdmError):
        ofdm_mi355x.RxEngine(8, 64, 16, 62, (1, 3), 60, 100)       # no device -> error, never a NumPy path


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ofdm_mi355x import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.OfdmLibraryError):
        _lib.load()


def test_safe_pickle_rejects_code_execution(tmp_path):
    import pickle
    import numpy as np
    from ofdm_mi355x.safe_pickle import UnsafePickleError, load_ndarray, loads_ndarray

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned",))

    with pytest.raises(UnsafePickleError):
        loads_ndarray(pickle.dumps(Evil()))
    with pytest.raises(UnsafePickleError):
        loads_ndarray(pickle.dumps({"a": 1}))
    for proto in (2, 3, 4, 5):
        a = (np.arange(12).reshape(3, 4) * (1 + 2j)).astype(np.complex128)
        p = tmp_path / ("a%d.pckl" % proto)
        p.write_bytes(pickle.dumps(a, protocol=proto))
        assert np.array_equal(load_ndarray(str(p)), a)
    f = np.asfortranarray(np.arange(6, dtype=np.int32).reshape(2, 3))
    assert np.array_equal(loads_ndarray(pickle.dumps(f, protocol=2)), f)


def test_product_tree_never_touches_the_oracle_or_the_reference():
    """The oracle is test infrastructure: nothing under the package directory may import it, and nothing shipped (package,
    bench.py, __graft_entry__.py) may read /root/reference at run time."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "lte-gnu-radio-code_amd")
    offenders = []
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".c")):
                text = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", text, re.M) or re.search(r"#\s*include[^\n]*oracle", text):
                    offenders.append(os.path.join(dp, f))
                if re.search(r"open\([^)]*/root/reference|load[^(]*\([^)]*/root/reference", text):
                    offenders.append(os.path.join(dp, f) + " (reads the reference)")
    for f in ("bench.py", "__graft_entry__.py"):
        text = open(os.path.join(root, f)).read()
        for m in re.finditer(r"/root/reference[^\s\"')]*", text):
            line = text[:m.start()].count("\n") + 1
            src = text.splitlines()[line - 1]
            if not src.lstrip().startswith("#") and "os.path.isdir" not in src and "exists" not in src:
                offenders.append("%s:%d %s" % (f, line, src.strip()))
    assert not offenders, offenders
