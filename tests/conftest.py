import os, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle is test infrastructure: (re)build it when the .so is missing
    so = os.path.join(ROOT, "oracle", "libkmx_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    # the product binaries are built in-tree by __graft_entry__.build(); build them here if a fresh
    # checkout has none (hipcc cross-compiles without a GPU)
    if not (os.path.exists(os.path.join(ROOT, "kmtricks_amd", "libkmx.so")) and os.path.exists(os.path.join(ROOT, "kmtricks_amd", "kmx"))):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_first():
    """torch bundles its own HIP runtime: when a test process uses both torch.cuda and libkmx, torch has to
    initialise the device first (as bench.py does), otherwise torch reports 'No HIP GPUs are available'."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
