"""CPU-only: the C-ABI library loads, exports every symbol include/kmx.h declares, and fails
loudly (no fallback) when there is no GPU."""
import ctypes, os, re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "kmx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(kmx_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    so = os.path.join(ROOT, "kmtricks_amd", "libkmx.so")
    assert os.path.exists(so), "libkmx.so not built (run __graft_entry__.build())"
    lib = ctypes.CDLL(so)
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"libkmx.so does not export {s}"


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from kmtricks_amd import lib
    with pytest.raises(lib.KmxError, match="no HIP device"):
        lib.Context(0)
