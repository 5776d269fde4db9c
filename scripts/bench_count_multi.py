#!/usr/bin/env python3
"""kmx_count_reads_dev_multi against kmx_count_reads_dev: S synthetic samples (G bp genome, 150-bp reads at 6x, k = 31, m = 10, P
static partitions) per call, wall clock per SAMPLE (host buffers in, counts resident in HBM out)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, ctypes as C
from kmtricks_amd import lib
import orc
ap = argparse.ArgumentParser()
ap.add_argument("--genome", type=float, default=1e6); ap.add_argument("--partitions", type=int, default=256); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--raw", action="store_true", help="with the PartiInfo<5> statistics (sparse form, pinned buffers)")
ap.add_argument("--threads", type=int, default=1, help="host threads, each with a context of its own, calling at the same time (as the count workers of kmx pipeline do)")
a = ap.parse_args()
G, L, COV, K, M, P = int(a.genome), 150, 6, 31, 10, a.partitions
rng = np.random.default_rng(1)
comp = np.zeros(256, np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
def sample(seed):
    r = np.random.default_rng(seed)
    genome = r.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=G)
    n = G * COV // L
    st = r.integers(0, G - L, n)
    reads = genome[st[:, None] + np.arange(L)[None, :]]
    return reads.tobytes(), (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
import threading
rep = np.ascontiguousarray(orc.repart_static(M, P), dtype=np.uint16)
_lib = lib._lib; _vp = C.c_void_p
out = {}
for S in (1, 2, 4, 8):
    smp = [sample(100 + i) for i in range(S)]
    res = [None] * a.threads
    def work(tid):
        ctx = lib.Context(0); store = lib.Store(0)
        bp = (C.c_char_p * S)(*[b for b, _ in smp]); op = (_vp * S)(*[o.ctypes.data for _, o in smp]); ns = (C.c_uint64 * S)(*[len(o) - 1 for _, o in smp])
        sp = (_vp * 1)(store._h)
        lists, nk = (lib.KmxList * (S * P))(), (C.c_uint64 * (S * P))()
        info = np.zeros((S * P, 2), np.uint64)
        rws = None; keep = []
        if a.raw:
            rws = (lib.KmxSuperkRaw * S)()
            for i in range(S):
                pr = np.zeros(P * 1280, np.uint32); q = np.zeros((4 ** M, 3), np.uint32); keep += [pr, q]
                rws[i] = lib.KmxSuperkRaw(pr.ctypes.data, None, None, 0, q.ctypes.data, 4 ** M, 0)
        def call():
            rc = _lib.kmx_count_reads_dev_multi(ctx._h, S, bp, op, ns, K, M, rep.ctypes.data, P, 0, 0, 2, sp, 1, lists, nk, info.ctypes.data, rws)
            assert rc == 0, rc
        for _ in range(3): call()
        bar.wait()
        t0 = time.perf_counter()
        for _ in range(a.reps): call()
        res[tid] = (time.perf_counter() - t0) / a.reps
        bar.wait()
        store.close(); ctx.close()
    bar = threading.Barrier(a.threads)
    th = [threading.Thread(target=work, args=(t,)) for t in range(a.threads)]
    for t in th: t.start()
    for t in th: t.join()
    # per sample, as the stage sees it: a call's time over the samples all threads finished in it
    out[f"S={S}"] = round(max(res) / (S * a.threads) * 1e3, 4)
print(json.dumps({"genome": G, "partitions": P, "raw_stats": a.raw, "threads": a.threads, "ms_per_sample": out}))
