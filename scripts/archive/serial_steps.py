"""The bench batch run serially (submit, wait, free; nothing in flight beside it): under rocprofv3 --stats this gives the
duration of every libkmx kernel of the step when it has the GPU to itself."""
import os, sys
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
for _ in range(8):
    r = ctx.merge_dev(tasks); r.wait(); r.free()
print("kernel", r.kernel() if False else "done")
