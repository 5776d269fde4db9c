"""Randomised parity of the column-blocked merge against the oracle: list counts, partition sizes, similarity, private
k-mers, recurrence-min, soft-min, count / PA rows, few work items (many tiles each) or many.  KMX_MERGE_KERNEL=cols is
forced, so a case the kernel does not suit exercises the hand-back chain instead."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from synth import synth_lists
from kmtricks_amd import lib
import orc

if len(sys.argv) > 3 and sys.argv[3] == "auto": os.environ.pop("KMX_MERGE_KERNEL", None)      # libkmx chooses (and backs off)
else: os.environ["KMX_MERGE_KERNEL"] = "cols"
big = "big" in sys.argv[3:]
KW = 2 if "kw2" in sys.argv[3:] else 1      # 128-bit keys (k >= 32): merge_cols_k2.hip
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = lib.Context(0)
kernels = {}
for case in range(n_cases):
    N = rng.choice([9, 64, 127, 128, 129, 200, 256, 257, 384, 511, 600, 777, 1000, 1024, 1500])
    pool = rng.choice([40, 300, 1500, 6000, 15000]) if N <= 600 else rng.choice([40, 300, 1500, 5000])
    if big: N = rng.choice([300, 600, 1000]); pool = rng.choice([20000, 40000])      # work items of many tiles
    p = rng.choice([0.999, 0.97, 0.9, 0.6, 0.2]) if not big else rng.choice([0.999, 0.97, 0.93, 0.85])
    priv = int(pool * (rng.choice([0.0, 0.01, 0.03, 0.1, 0.4]) if not big else rng.choice([0.0, 0.01, 0.03, 0.08])))
    rec_min = rng.choice([0, 1, 1, 2, 2, 2, 3, 5, 9])
    share = rng.choice([0, 0, 0, 1, max(1, rec_min), rec_min + 2, int(N * p * 0.5)])      # (above max(1, recurrence-min): count rows through the pair + k_share_fix, PA rows through k_merge_rows)
    mode = rng.choice([lib.MODE_COUNT, lib.MODE_COUNT, lib.MODE_PA])
    os.environ["KMX_ITEMS_PER_SLOT"] = rng.choice(["1", "3"])
    lists = synth_lists(rng.randrange(1 << 30), N, pool, p, priv, kw=KW, key_bits=62 if KW == 1 else rng.choice([66, 72, 126]),
                        count_max=rng.choice([2, 5, 50, 300, 70000]), ragged=rng.random() < 0.2)
    if KW == 1 and rng.random() < 0.3:      # a list with a long run of keys nobody else has
        i = rng.randrange(N); k, c = lists[i]
        if len(k):
            lo = int(k[len(k) // 2, 0]); run = (np.arange(1, 400, dtype=np.uint64) + np.uint64(lo)).reshape(-1, 1)
            run = run[~np.isin(run[:, 0], k[:, 0])]
            k2 = np.concatenate([k, run]); c2 = np.concatenate([c, np.full(len(run), 3, np.uint32)])
            o = np.argsort(k2[:, 0]); lists[i] = (np.ascontiguousarray(k2[o]), np.ascontiguousarray(c2[o]))
    soft = [rng.choice([1, 1, 2, 3]) for _ in range(N)]
    print(f"case {case}: N={N} pool={pool} p={p} priv={priv} rec_min={rec_min} share={share} mode={mode} ...", flush=True)
    eb, er, es = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], KW, soft, rec_min, share, mode)
    body, rows, stats = ctx.merge(lists, KW, soft, rec_min, share, mode)
    ok = rows == er and body == eb and np.array_equal(stats, es)
    print(f"case {case}: N={N} pool={pool} p={p} priv={priv} rec_min={rec_min} share={share} mode={mode} rows={rows} {'ok' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        sys.exit(1)
print("all", n_cases, "cases equal the oracle")
