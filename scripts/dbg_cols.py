import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from synth import synth_lists
from kmtricks_amd import lib
import orc
os.environ["KMX_MERGE_KERNEL"] = "cols"; os.environ["KMX_TRACE"] = "1"
ctx = lib.Context(0)
for N, pool in ((600, 3000), (1000, 3000), (1000, 300), (1024, 3000), (768, 3000), (520, 3000)):
    lists = synth_lists(11 + N, N, pool, 0.97, int(pool * 0.03), kw=1)
    body, rows, stats = ctx.merge(lists, 1, [1] * N, 2, 0, lib.MODE_COUNT)
    eb, er, es = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], 1, [1] * N, 2, 0, orc.MODE_COUNT)
    print("N", N, "pool", pool, "rows", rows, er, "ok", body == eb and np.array_equal(stats, es), flush=True)
