"""PA rows (kmer:pa:bin) of the bench cohort: step time per COUNT/PA kernel (KMX_MERGE_KERNEL)."""
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
ctx = lib.Context(0); ctx.set_profiling(True)
tasks = ctx.prepare([dict(lists=[(rec.data_ptr() + 12 * offs[i], offs[i + 1] - offs[i]) for i in range(N)], key_words=1, soft_min=[1] * N,
                          rec_min=2, share_min=0, mode=lib.MODE_PA, rows_hint=shared + 4096) for rec, offs in parts])
for kern in ("rows", "pivot", "cols"):
    os.environ["KMX_MERGE_KERNEL"] = kern
    for _ in range(2):
        r = ctx.merge_dev(tasks); r.wait(); r.free()
    t0 = time.perf_counter(); prev = None; kms = []
    for _ in range(5):
        cur = ctx.merge_dev(tasks)
        if prev is not None: prev.wait(); kms.append(prev.kernel_ms()); prev.free()
        prev = cur
    prev.wait(); name = prev.kernel(); kms.append(prev.kernel_ms()); prev.free()
    print(kern, name, "step ms", round((time.perf_counter() - t0) / 5 * 1e3, 3), "kernel ms", round(sum(kms) / len(kms), 3), flush=True)
