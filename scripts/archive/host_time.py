"""Host-side cost of the submit path on the bench workload: time inside kmx_merge_dev, and inside wait + free
(two batches in flight, as bench.py runs them)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from kmtricks_amd import lib
N, P = 1000, 32
dev = torch.device("cuda", 0)
shared = int(5e6 / 256); pp = (1 - 0.001) ** 31; npriv = int(round(shared * (1 - pp)))
parts = [bench.gen_partition(torch, dev, 20240601 + g, N, shared, pp, npriv) for g in range(P)]
torch.cuda.synchronize()
ctx = lib.Context(0)
tasks = ctx.prepare([dict(lists=[(rec.data_ptr() + 12 * offs[i], offs[i + 1] - offs[i]) for i in range(N)], key_words=1, soft_min=[1] * N,
                          rec_min=2, share_min=0, mode=lib.MODE_COUNT, rows_hint=shared + 4096) for rec, offs in parts])
for _ in range(3):
    r = ctx.merge_dev(tasks); r.wait(); r.free()
ts, tw = [], []
prev = None
t00 = time.perf_counter()
for _ in range(10):
    t0 = time.perf_counter(); cur = ctx.merge_dev(tasks); t1 = time.perf_counter(); ts.append(t1 - t0)
    if prev is not None:
        t0 = time.perf_counter(); prev.wait(); t1 = time.perf_counter(); prev.free(); t2 = time.perf_counter(); tw.append((t1 - t0, t2 - t1))
    prev = cur
prev.wait(); prev.free()
print("step ms", (time.perf_counter() - t00) / 10 * 1e3)
print("merge_dev ms", [round(x * 1e3, 3) for x in ts])
print("wait, free ms", [(round(a * 1e3, 3), round(b * 1e3, 3)) for a, b in tw])
