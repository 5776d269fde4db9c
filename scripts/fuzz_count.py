#!/usr/bin/env python3
"""Randomised cross-check of the split + count path against the oracle: random k, m, partitions, hard-min, hash mode, read lengths
(short reads, reads shorter than k, long reads, N's, low-complexity stretches, duplicated reads), through kmx_count_reads (streams
asked for: the record-stream decode), kmx_count_reads (no streams: the packed-bases decode) and kmx_count_reads_dev + statistics.
`--cases N --seed S --scale X`; prints one JSON line; exit code 1 on the first difference."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from kmtricks_amd import lib
import orc

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60); ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--scale", type=int, default=1, help="times as many reads per case (partitions of many buckets)")
ap.add_argument("--kmers", choices=["narrow", "wide", "all"], default="narrow", help="k <= 63, k = 64 ... 127 (keys of two to four words), or both")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
ctx = lib.Context(0)
done = 0
for case in range(a.cases):
    ks = ([12, 15, 20, 21, 27, 31, 32, 33, 40, 47, 55, 63] if a.kmers != "wide" else []) + ([64, 65, 80, 95, 96, 97, 111, 127] if a.kmers != "narrow" else [])
    k = int(rng.choice(ks))
    m = int(rng.integers(4, min(15, k) + 1)) if k < 20 else int(rng.choice([7, 8, 10, 11, 12]))
    P = int(rng.choice([1, 2, 3, 8, 16, 37, 64, 256]))
    hard_min = int(rng.choice([1, 1, 2, 3]))
    hashed = bool(rng.integers(0, 2))
    W = int(rng.choice([6400, 100003, 1 << 20]))
    style = int(rng.integers(0, 5))
    nreads = int(rng.integers(1, 1500)) * a.scale
    reads = []
    for _ in range(nreads):
        L = int(rng.choice([rng.integers(0, k + 3), rng.integers(k, 200), 150, 150, rng.integers(200, 700), rng.integers(1000, 3000) if style == 4 else 100]))
        s = rng.choice(list("ACGT"), size=L, p=[0.4, 0.1, 0.1, 0.4] if style == 1 else None)
        if style == 2 and L > 40:
            s[10:10 + L // 3] = "A"          # low complexity: the m-mer value with the AA rule, long super-k-mers
        s[rng.random(L) < (0.01 if style == 3 else 0.001)] = "N"
        reads.append("".join(s))
    if style in (0, 3):
        reads = reads + reads[: len(reads) // 2]      # repeats: counts above 1, the splitter samples' strata
    if os.environ.get("KMX_FUZZ_VERBOSE"):
        print(f"case {case}: k={k} m={m} P={P} hard_min={hard_min} hashed={hashed} W={W} style={style} reads={len(reads)}", file=sys.stderr, flush=True)
    lut = orc.minimizer_lut(m)
    rep = orc.repart_static(m, P)
    exp = orc.superk_partition(reads, k, m, lut, rep, P)
    got_s, nk_s, streams, info = ctx.count_reads(reads, k, m, rep, P, hard_min, window=W if hashed else 0, streams=True)
    got_d, nk_d, none, _ = ctx.count_reads(reads, k, m, rep, P, hard_min, window=W if hashed else 0)
    store = lib.Store(0)
    lists, nk_v, raw = ctx.count_reads_dev(reads, k, m, rep, P, hard_min, [store], window=W if hashed else 0, raw=True, sparse=bool(rng.integers(0, 2)))
    kw = 1 if hashed else (k + 31) // 32
    for p in range(P):
        ek, ec = (orc.count_hash(exp[p][0], k, W, p, hard_min) if hashed else orc.count_kmer(exp[p][0], k, hard_min))
        ok = streams[p] == exp[p][0] and nk_s[p] == exp[p][1] and nk_d[p] == exp[p][1] and nk_v[p] == exp[p][1]
        ok = ok and np.array_equal(got_s[p][0], ek) and np.array_equal(got_s[p][1], ec) and np.array_equal(got_d[p][0], ek) and np.array_equal(got_d[p][1], ec)
        rk, rc = ctx.read_list(lists[p][0], lists[p][1], kw)
        ok = ok and np.array_equal(rk.reshape(ek.shape), ek) and np.array_equal(rc, ec)
        if not ok:
            print(json.dumps({"failed_case": case, "seed": a.seed, "k": k, "m": m, "P": P, "hard_min": hard_min, "hashed": hashed, "style": style, "reads": len(reads), "partition": p}))
            sys.exit(1)
    epin, ems, emk, _ = orc.superk_stats(reads, k, m, lut, rep, P)
    pr, ms, mk, nsk = raw
    pr = pr.reshape(P, 5, 256).astype(np.uint64)
    if not (np.array_equal(pr.reshape(P, 1280), epin[:, 2:]) and np.array_equal(ms, ems) and np.array_equal(mk, emk) and nsk == int(ems.sum())):
        print(json.dumps({"failed_case": case, "what": "PartiInfo statistics", "seed": a.seed, "k": k, "m": m, "P": P, "style": style, "reads": len(reads)}))
        sys.exit(1)
    store.close()
    done += 1
print(json.dumps({"cases": done, "seed": a.seed, "all_equal_to_oracle": True}))
