#!/usr/bin/env python3
"""Does a launch of the file-order pair ever take far longer than the others?  The lists of configs[4] (or configs[2]) once, then a FRESH
context per iteration (arena sizing, first-batch re-runs and all) and a few batches each; every launch's HIP-event times are kept.
Prints the slowest launches and every one beyond 10x the median."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import argparse, faulthandler
faulthandler.dump_traceback_later(75, repeat=True)
ap = argparse.ArgumentParser()
ap.add_argument("--wl", default="pa63"); ap.add_argument("--order", type=int, default=1); ap.add_argument("--fresh", type=int, default=1); ap.add_argument("--budget", type=float, default=240); ap.add_argument("--iters", type=int, default=150); ap.add_argument("--steps", type=int, default=8)
a = ap.parse_args()
import torch, bench
from kmtricks_amd import lib, shard
ctx0 = lib.Context(0)
k = 31 if a.wl == "count" else 63
kw = (k + 31) // 32
mode = lib.MODE_COUNT if a.wl == "count" else lib.MODE_PA
rec_min = 2 if a.wl == "count" else 1
N = 1000 if a.wl == "count" else 500
parts = shard.partitions_of_rank(32, 1, 0)
store, lists = bench.gen_counted(ctx0, lib, N, k, 5_000_000, 0.001, 256, parts, 20240601, False)
td = [dict(lists=ls, key_words=kw, soft_min=[1] * N, rec_min=rec_min, share_min=0, mode=mode) for ls in lists]
allms = []
t00 = time.perf_counter()
ctx = None
for it in range(a.iters):
    if ctx is None or a.fresh:
        ctx = lib.Context(0); ctx.set_profiling(True); ctx.set_file_order(bool(a.order))
        tasks = ctx.prepare(td)
    ti = time.perf_counter()
    prev = None
    for s in range(a.steps):
        cur = ctx.merge_dev(tasks)
        if prev is not None:
            prev.wait(); allms.append((prev.kernel_ms(),) + prev.kernel_parts_ms() + (it, s - 1)); prev.free()
        prev = cur
    prev.wait(); allms.append((prev.kernel_ms(),) + prev.kernel_parts_ms() + (it, a.steps - 1)); prev.free()
    if a.fresh: ctx.close()
    print(f"it {it} {time.perf_counter() - ti:.3f} s, launches so far {len(allms)}, slowest {max(x[0] for x in allms):.1f} ms", flush=True)
    if time.perf_counter() - t00 > a.budget: break
ms = sorted(x[0] for x in allms); med = ms[len(ms) // 2]
slow = [x for x in allms if x[0] > 10 * med]
print(json.dumps({"wl": a.wl, "launches": len(allms), "median_ms": round(med, 3), "max_ms": round(ms[-1], 3), "p99_ms": round(ms[int(len(ms) * 0.99)], 3),
                  "beyond_10x_median": [dict(pair_ms=round(x[0], 1), cols_ms=round(x[1], 1), sparse_ms=round(x[2], 1), iteration=x[3], step=x[4]) for x in slow]}))
