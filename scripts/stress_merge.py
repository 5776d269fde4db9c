"""Randomised parity of kmx_merge against the oracle over everything the merge takes: 64- and 128-bit keys, count / PA /
Bloom (bf, bfc) rows, soft-min, recurrence-min, share-min (rescue), list counts from 1 to 1100, similar and unrelated
lists, empty lists; the kernel is libkmx's own choice or forced (argv[3] = rows | pivot | cols)."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from synth import synth_lists, synth_hash_lists
from kmtricks_amd import lib
import orc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
if len(sys.argv) > 3: os.environ["KMX_MERGE_KERNEL"] = sys.argv[3]
else: os.environ.pop("KMX_MERGE_KERNEL", None)
ctx = lib.Context(0)
for case in range(n_cases):
    mode = rng.choice([lib.MODE_COUNT, lib.MODE_COUNT, lib.MODE_PA, lib.MODE_BF, lib.MODE_BFC])
    N = rng.choice([1, 2, 3, 7, 31, 64, 100, 130, 255, 513, 640, 1100])
    rec_min = rng.choice([1, 1, 2, 3, 6]); share = rng.choice([0, 0, 0, 1, 2, 4])
    soft = [rng.choice([1, 1, 2, 4]) for _ in range(N)]
    lower = upper = 0; bitw = rng.choice([1, 2, 3, 8])
    if mode in (lib.MODE_BF, lib.MODE_BFC):
        kw = 1; window = rng.choice([64, 1000, 20000]); lower = rng.choice([0, window * 3]); upper = lower + window - 1
        lists = synth_hash_lists(rng.randrange(1 << 30), N, lower, window, rng.choice([0.02, 0.3, 0.9]), count_max=rng.choice([2, 20]))
        desc = f"window={window}"
    else:
        kw = rng.choice([1, 1, 2])
        pool = rng.choice([10, 200, 3000]) if N > 200 else rng.choice([10, 200, 3000, 20000])
        p = rng.choice([0.99, 0.9, 0.5, 0.1]); priv = int(pool * rng.choice([0, 0.02, 0.3]))
        lists = synth_lists(rng.randrange(1 << 30), N, pool, p, priv, kw=kw, key_bits=62 if kw == 1 else 100, count_max=rng.choice([2, 6, 60]), ragged=rng.random() < 0.3)
        desc = f"pool={pool} p={p} priv={priv}"
    print(f"case {case}: mode={mode} N={N} kw={kw} rec_min={rec_min} share={share} {desc} ...", flush=True)
    eb, er, es = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], kw, soft, rec_min, share, mode, lower, upper, bitw)
    body, rows, stats = ctx.merge(lists, kw, soft, rec_min, share, mode, lower, upper, bitw)
    if not (rows == er and body == eb and np.array_equal(stats, es)):
        print("MISMATCH rows", rows, er, "body", body == eb, "stats", np.array_equal(stats, es)); sys.exit(1)
print("all", n_cases, "cases equal the oracle")
