import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from kmtricks_amd import lib
ctx = lib.Context(0)
import time
for k in (63, 31):
    t0 = time.perf_counter()
    store, lists = bench.gen_counted(ctx, lib, 3, k, 5000000, 0.001, 256, list(range(32)), 20240601, False)
    print("k", k, "3 samples", time.perf_counter() - t0, "s; sizes", sorted(n for _, n in [l[0] for l in lists])[-3:], flush=True)
