"""Randomised parity of the super-k-mer split and the counting stage against the oracle: k, minimizer size, partition
count, read lengths from below k to 20 kb, N-rich / low-complexity / lower-case reads, hard-min, k-mer and hash mode."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from kmtricks_amd import lib
import orc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = lib.Context(0)
for case in range(n_cases):
    k = rng.choice([12, 20, 21, 25, 31, 32, 33, 47, 63]); m = rng.choice([x for x in (5, 7, 8, 10, 11, 12) if x < k - 1])
    P = rng.choice([1, 3, 4, 16, 37])
    reads = []
    for _ in range(rng.choice([1, 20, 300])):
        L = rng.choice([0, 1, k - 1, k, k + 1, 63, 64, 65, 100, 150, 151, 1000, 20000]) if rng.random() < 0.5 else rng.randrange(0, 400)
        kind = rng.random()
        if kind < 0.6: s = "".join(rng.choice("ACGT") for _ in range(L))
        elif kind < 0.7: s = "".join(rng.choice("ACGTN") for _ in range(L))
        elif kind < 0.8: s = (rng.choice(["A", "AC", "ACG", "AAT", "ACGT"]) * (L // 2 + 1))[:L]
        elif kind < 0.9: s = "".join(rng.choice("acgtACGT") for _ in range(L))
        else: s = "".join(rng.choice("ACGT") if rng.random() > 0.02 else "N" for _ in range(L))
        reads.append(s)
    lut = orc.minimizer_lut(m); rep = orc.repart_static(m, P)
    print(f"case {case}: k={k} m={m} P={P} reads={len(reads)} bases={sum(map(len, reads))} ...", flush=True)
    exp = orc.superk_partition(reads, k, m, lut, rep, P)
    got = ctx.superk_partition(reads, k, m, rep, P)
    for p in range(P):
        if not (got[p][1] == exp[p][1] and got[p][0] == exp[p][0]):
            print("MISMATCH superk partition", p); sys.exit(1)
    streams = [exp[p][0] for p in range(P)]
    hm = rng.choice([1, 2, 3]); W = rng.choice([64, 6400, 1000003])
    gb = ctx.count_batch(streams, k, hm); gh = ctx.count_batch(streams, k, hm, window=W, partitions=list(range(P)))
    for p in range(P):
        ek, ec = orc.count_kmer(streams[p], k, hm); eh, ehc = orc.count_hash(streams[p], k, W, p, hm)
        if not (np.array_equal(ek, gb[p][0]) and np.array_equal(ec, gb[p][1]) and np.array_equal(eh, gh[p][0]) and np.array_equal(ehc, gh[p][1])):
            print("MISMATCH count partition", p); sys.exit(1)
        if p == 0:
            a, b = ctx.count_kmer(streams[p], k, hm); c, d = ctx.count_hash(streams[p], k, W, p, hm)
            if not (np.array_equal(ek, a) and np.array_equal(ec, b) and np.array_equal(eh, c) and np.array_equal(ehc, d)):
                print("MISMATCH single-stream count"); sys.exit(1)
print("all", n_cases, "cases equal the oracle")
