#!/usr/bin/env python3
"""The count stage by itself, as `kmx pipeline` runs it: one synthetic sample (G bp genome, 150-bp error-free reads at 6x, k = 31,
m = 10, P static partitions) through kmx_count_reads_dev (split + count in one call, results left in HBM) REPS times.  Prints the
wall clock per call (host buffers in: the reads' upload is inside) and the roofline of the stage on SURVEY 8d's bytes
(B = super-k-mer stream bytes + 12 per distinct solid k-mer).  Run under `rocprofv3 --kernel-trace --stats` for the kernels'
shares (scripts/profile_count_stage.sh)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from kmtricks_amd import lib
import orc

ap = argparse.ArgumentParser()
ap.add_argument("--genome", type=float, default=5e6); ap.add_argument("--partitions", type=int, default=32)
ap.add_argument("--skip-streams", action="store_true"); ap.add_argument("--resident", type=int, default=1); ap.add_argument("--reps", type=int, default=10); ap.add_argument("--hash", action="store_true"); ap.add_argument("--kmer-size", type=int, default=31)
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
W = 3125056 if a.hash else 0
store = lib.Store(0)
handle = ctx.upload_reads(blob, offs) if a.resident else None      # (round 6: the bases resident in HBM, as the pipeline has them when the call starts)
for _ in range(3):
    ctx.count_reads_dev((blob, offs), K, M, rep, P, 2, [store], window=W, resident=handle)      # warm-up at full size (module load, pools)
ts = []
for _ in range(a.reps):
    t0 = time.perf_counter(); lists, nk, _ = ctx.count_reads_dev((blob, offs), K, M, rep, P, 2, [store], window=W, resident=handle); ts.append(time.perf_counter() - t0)
if handle: ctx.release_reads(handle)
distinct = sum(n for _, n in lists)
if a.skip_streams:      # (profiled runs: only the calls above -- the super-k-mer bytes come from an unprofiled run's line)
    nb, same = 0, True
else:
    got = ctx.count_reads((blob, offs), K, M, rep, P, 2, window=W, streams=True)      # (the path that writes the record streams: SURVEY 8d's S_p)
    nb = sum(len(s) for s in got[2])
    same = all(n == len(got[0][p][1]) for p, (_, n) in enumerate(lists))
t = sorted(ts)[len(ts) // 2]
print(json.dumps({"genome": G, "reads": n_reads, "bases": n_reads * L, "kmers": sum(nk), "superk_bytes": nb, "distinct_solid": distinct, "partitions": P, "hash": bool(a.hash),
                  "count_reads_dev_ms_median": t * 1e3, "count_reads_dev_ms_min": min(ts) * 1e3, "Gbases_per_s": n_reads * L / t / 1e9, "Gkmers_per_s": sum(nk) / t / 1e9,
                  "algorithmic_bytes": nb + 12 * distinct, "same_counts_as_count_reads": bool(same)}))
store.close(); ctx.close()
