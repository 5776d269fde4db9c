"""time of kmx_copy_to_host per piece: device -> page-locked host memory (kmx_alloc_pinned), pieces of several sizes"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kmtricks_amd import lib
ctx = lib.Context(0)
src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for mb in (8, 32, 128):
    n = mb << 20
    bufs = [lib._lib.kmx_alloc_pinned(n) for _ in range(4)]
    for rep in range(2):
        t0 = time.perf_counter(); k = 0
        for off in range(0, 1 << 30, n):
            rc = lib._lib.kmx_copy_to_host(ctx._h, bufs[k % 4], src.data_ptr() + off, n); k += 1
            assert rc == 0
        dt = time.perf_counter() - t0
    print(f"{mb} MB pieces: {dt / k * 1e3:.3f} ms per piece, {(1 << 30) / dt / 1e9:.1f} GB/s")
    for b in bufs: lib._lib.kmx_free_pinned(b)
