#!/usr/bin/env python3
"""kmx_count_reads_dev on configs[2]'s sample (30 Mbases), the bases sent ahead (kmx_reads_upload) as `kmx pipeline` does, or from the host blob"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from kmtricks_amd import lib
import orc
ap = argparse.ArgumentParser()
ap.add_argument("--genome", type=float, default=5e6); ap.add_argument("--partitions", type=int, default=256)
ap.add_argument("--reps", type=int, default=6); ap.add_argument("--ahead", type=int, default=1); ap.add_argument("--raw", type=int, default=0)
ap.add_argument("--kmer-size", type=int, default=31)
a = ap.parse_args()
rng = np.random.default_rng(20240601)
G, L, COV, K, M, P = int(a.genome), 150, 6, a.kmer_size, 10, a.partitions
genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=G)
comp = np.zeros(256, np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
n_reads = G * COV // L
starts = rng.integers(0, G - L, n_reads)
reads = genome[starts[:, None] + np.arange(L)[None, :]]
rc = rng.random(n_reads) < 0.5
reads[rc] = comp[reads[rc]][:, ::-1]
blob = reads.tobytes(); offs = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L))
ctx = lib.Context(0)
rep = orc.repart_static(M, P)
store = lib.Store(0)
kw = dict(ahead=bool(a.ahead), raw=bool(a.raw), sparse=bool(a.raw))
for _ in range(3):
    ctx.count_reads_dev((blob, offs), K, M, rep, P, 2, [store], **kw)
ts = []
for _ in range(a.reps):
    t0 = time.perf_counter(); lists, nk, _ = ctx.count_reads_dev((blob, offs), K, M, rep, P, 2, [store], **kw); ts.append(time.perf_counter() - t0)
print(json.dumps({"bases": n_reads * L, "kmers": sum(nk), "distinct_solid": sum(n for _, n in lists), "partitions": P, "ahead": a.ahead, "raw": a.raw,
                  "wall_ms_median_incl_python": sorted(ts)[len(ts) // 2] * 1e3}))
store.close(); ctx.close()
