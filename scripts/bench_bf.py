#!/usr/bin/env python3
"""BASELINE configs[1]: 100 synthetic 5 Mbp samples, k=31, hash:bf:bin, bloom 1e8, 32 partitions on one
MI355X: times k_merge_bf over all 32 partitions (inputs resident in HBM) and checks one partition
against the oracle.  Prints a JSON line (not the driver's bench; see bench.py)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from kmtricks_amd import lib

N, P, BLOOM = 100, 32, 100_000_000
W = ((BLOOM + P - 1) // P + 63) // 64 * 64          # 3 125 056 (hash.hpp:31-40)
per_list = int(5e6 / P)                              # k-mers of a sample in a partition
dev = torch.device("cuda", 0)
ctx = lib.Context(0); ctx.set_profiling(True)
g = torch.Generator(device=dev); g.manual_seed(20240601)
parts, total = [], 0
for p in range(P):
    lo = W * p
    # shared ancestor hashes + sample-private ones (same sharing model as bench.py), hashed into the window
    pool = torch.randint(lo, lo + W, (per_list,), generator=g, device=dev, dtype=torch.int64)
    recs, offs = [], [0]
    for i in range(N):
        keep = torch.rand(per_list, generator=g, device=dev) < 0.969
        priv = torch.randint(lo, lo + W, (int(per_list * 0.031),), generator=g, device=dev, dtype=torch.int64)
        h = torch.unique(torch.cat([pool[keep], priv]))                  # sorted, distinct (colliding k-mers are summed)
        r = torch.empty((h.numel(), 3), device=dev, dtype=torch.int32)
        r[:, :2] = h.view(torch.int32).view(-1, 2); r[:, 2] = torch.randint(2, 12, (h.numel(),), generator=g, device=dev, dtype=torch.int32)
        recs.append(r); offs.append(offs[-1] + h.numel())
    rec = torch.cat(recs); parts.append((rec, offs)); total += rec.shape[0]
torch.cuda.synchronize()
tasks = []
for p, (rec, offs) in enumerate(parts):
    base = rec.data_ptr()
    tasks.append(dict(lists=[(base + 12 * offs[i], offs[i + 1] - offs[i]) for i in range(N)], key_words=1, soft_min=[1] * N,
                      rec_min=1, share_min=0, mode=lib.MODE_BF, lower=W * p, upper=W * (p + 1) - 1))
prep = ctx.prepare(tasks)
kms = []
for it in range(6):
    t0 = time.perf_counter(); res = ctx.merge_dev(prep); res.wait(); dt = time.perf_counter() - t0
    if it >= 2: kms.append((res.kernel_ms(), dt))
    algo = sum(res.algo_bytes(t) for t in range(P))
    if it == 5:
        import orc
        rec, offs = parts[3]; h = rec.cpu().numpy()
        lists = [(np.ascontiguousarray(h[offs[i]:offs[i+1], :2]).view(np.uint64).reshape(-1), np.ascontiguousarray(h[offs[i]:offs[i+1], 2]).view(np.uint32)) for i in range(N)]
        body, rows, st = orc.merge_matrix(lists, 1, [1] * N, 1, 0, orc.MODE_BF, W * 3, W * 4 - 1)
        ok = res.body(3) == body and np.array_equal(res.stats(3), st)
    res.free()
k = sum(x[0] for x in kms) / len(kms); w = sum(x[1] for x in kms) / len(kms)
print(json.dumps({"workload": "configs[1]: 100 samples, hash:bf:bin, bloom 1e8, 32 partitions, 1 GPU", "records": total,
                  "kernel_ms": k, "wall_ms": w * 1e3, "kmers_per_s": total / w, "algo_bytes": algo,
                  "achieved_GBps": algo / (k * 1e-3) / 1e9, "frac_of_8TBps": algo / (k * 1e-3) / 8e12, "bit_exact_vs_oracle_partition3": bool(ok)}))
