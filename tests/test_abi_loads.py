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


def test_product_never_reaches_the_oracle():
    """The oracle is test infrastructure: nothing under kmtricks_amd/ or include/ may include, link, dlopen or execute anything under
    oracle/ (bench.py may, in its cpu_baseline legs only -- that is checked by reading where it uses `orc`)."""
    import subprocess
    bad = []
    for top in ("kmtricks_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if not f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".c", "Makefile")): continue
                txt = open(os.path.join(d, f), errors="replace").read()
                if re.search(r"kmx_oracle|liborc|oracle/|tests\.orc|import orc", txt): bad.append(os.path.join(d, f))
    assert not bad, f"product files that mention the oracle: {bad}"
    # the built library and the driver link neither of the oracle's objects
    for name in ("libkmx.so", "kmx"):
        path = os.path.join(ROOT, "kmtricks_amd", name)
        if not os.path.exists(path): continue
        out = subprocess.run(["ldd", path], capture_output=True, text=True).stdout
        assert "oracle" not in out, f"{name} links the oracle:\n{out}"
