"""What one outlier sample does to the column-blocked merge: a cohort of 300 similar lists plus one unrelated list of 1x / 3x / 6x
their size (KMX_TRACE shows the kernel per batch and the hand-back reasons).  Results equal the oracle's either way."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from synth import synth_lists
from kmtricks_amd import lib
import orc
os.environ["KMX_TRACE"] = "1"
ctx = lib.Context(0)
for mult in (1, 3, 6):
    lists = synth_lists(11, 300, 20000, 0.97, 200, kw=1)
    rng = np.random.default_rng(5)
    k = np.unique(rng.integers(0, 1 << 62, 20000 * mult, dtype=np.uint64)).reshape(-1, 1)
    lists[137] = (k, rng.integers(1, 9, len(k), dtype=np.uint32))
    print(f"--- outlier list of {mult}x the cohort's list size", flush=True)
    eb, er, es = orc.merge_matrix([(a.reshape(-1), c) for a, c in lists], 1, [1] * 300, 2, 0, lib.MODE_COUNT)
    body, rows, stats = ctx.merge(lists, 1, [1] * 300, 2, 0, lib.MODE_COUNT)
    print("equal to the oracle:", rows == er and body == eb and np.array_equal(stats, es), "rows", rows, flush=True)
