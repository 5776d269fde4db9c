#!/usr/bin/env python3
"""Round 5: what the two persistent kernels of the column-blocked pair lose on a part of the chip.
configs[2]'s lists from the count stage (built once), then a batch per setting of KMX_COLS_GRID (workgroups of k_merge_cols) /
KMX_SPARSE_CUS (CUs' worth of k_cols_sparse's ORD build): HIP events around each kernel (kmx_result_kernel_parts_ms)."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=0)
ap.add_argument("--parts", type=int, default=32)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--wl", default="count")
ap.add_argument("--settings", default="")
a = ap.parse_args()
import torch
import bench
from kmtricks_amd import lib, shard
ctx = lib.Context(0); ctx.set_profiling(True)
k = 31 if a.wl == "count" else 63      # (count63: count rows with 128-bit keys)
kw = (k + 31) // 32
mode = lib.MODE_COUNT if a.wl in ("count", "count63") else lib.MODE_PA
rec_min = 2 if a.wl in ("count", "count63") else 1
N = a.samples or (1000 if a.wl == "count" else 500)
parts = shard.partitions_of_rank(a.parts, 1, 0)
store, lists = bench.gen_counted(ctx, lib, N, k, 5_000_000, 0.001, 256, parts, 20240601, True)
tasks = ctx.prepare([dict(lists=ls, key_words=kw, soft_min=[1] * N, rec_min=rec_min, share_min=0, mode=mode) for ls in lists])

def run(n):
    out = []
    prev = None
    for _ in range(n):
        cur = ctx.merge_dev(tasks)
        if prev is not None:
            prev.wait(); out.append((prev.kernel_ms(),) + prev.kernel_parts_ms()); prev.free()
        prev = cur
    prev.wait(); out.append((prev.kernel_ms(),) + prev.kernel_parts_ms()); prev.free()
    return out

for _ in range(3):
    run(1)
settings = [s for s in a.settings.split(";") if s] or ["", "KMX_COLS_GRID=192", "KMX_COLS_GRID=128", "KMX_SPARSE_CUS=192", "KMX_SPARSE_CUS=128", "KMX_SPARSE_CUS=96", "KMX_SPARSE_CUS=64"]
for fo in (True, False):
    ctx.set_file_order(fo)
    run(2)
    for st in settings:
        kv = dict(x.split("=") for x in st.split(",") if x)
        for kk, vv in kv.items():
            os.environ[kk] = vv
        run(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = run(a.steps)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps * 1e3
        for kk in kv:
            del os.environ[kk]
        m = [sum(x[i] for x in r) / len(r) for i in range(3)]
        print(json.dumps({"wl": a.wl, "file_order": fo, "setting": st or "default", "pair_ms": round(m[0], 3), "cols_ms": round(m[1], 3), "sparse_ms": round(m[2], 3), "step_ms": round(dt, 3)}), flush=True)
    if not fo:
        break
