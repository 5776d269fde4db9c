#!/usr/bin/env python3
"""Full-size cross-check on the bench workloads, on the lists the bench times (`--lists counted`: the product's count stage,
kmx_count_reads_dev; `random`: the round-1 generator): BASELINE configs[2] (`count`: 1000 samples, k = 31, count rows,
recurrence-min 2, 32 partitions) or configs[4] (`pa63`: 500 samples, k = 63, PA rows, recurrence-min 1).  The matrix bodies and
statistics produced by k_merge_cols (+ k_cols_sparse), k_merge_pivot (64-bit keys) and k_merge_rows are compared byte for byte
(sha256 per partition), rows are checked ascending, the column-blocked result must have come from k_merge_cols with rows out of
k_cols_sparse, and the first partition is compared with the oracle.  Reference semantics: merge.hpp:183-286.
Run as a script (one JSON line) or through tests/test_merge_gpu.py::test_bench_workload_full_size_parity."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def verify(workload="count", lists_kind="counted", N=None, P=32, genome=5e6, d=0.001, total_parts=256, oracle_parts=1):
    import numpy as np
    import torch
    import bench
    from kmtricks_amd import lib
    import orc
    k, rec_min, mode, omode = (31, 2, lib.MODE_COUNT, orc.MODE_COUNT) if workload == "count" else (63, 1, lib.MODE_PA, orc.MODE_PA)
    N = N or (1000 if workload == "count" else 500)
    kw = (k + 31) // 32
    rb = 8 * kw + 4
    dev = torch.device("cuda", 0)
    ctx = lib.Context(0)
    keep = []
    if lists_kind == "counted":
        store, lists = bench.gen_counted(ctx, lib, N, k, int(genome), d, total_parts, list(range(P)), 20240601, False)
        keep.append(store)
    else:
        shared = int(genome / total_parts); pp = (1.0 - d) ** k; npriv = int(round(shared * (1.0 - pp)))
        parts = [bench.gen_partition(torch, dev, 20240601 + g, N, shared, pp, npriv, kw) for g in range(P)]
        keep.append(parts)
        lists = [[(rec.data_ptr() + rb * offs[i], offs[i + 1] - offs[i]) for i in range(N)] for rec, offs in parts]
        torch.cuda.synchronize()
    tasks = [dict(lists=ls, key_words=kw, soft_min=[1] * N, rec_min=rec_min, share_min=0, mode=mode) for ls in lists]
    row_bytes = 8 * kw + (4 * N if workload == "count" else (N + 7) // 8)
    out = {}
    prev = os.environ.get("KMX_MERGE_KERNEL")
    try:
        for kern in (("rows", "pivot", "cols") if kw == 1 else ("rows", "cols")):
            os.environ["KMX_MERGE_KERNEL"] = kern
            res = ctx.merge_dev(tasks); res.wait()
            hs, rows, sparse = [], [], 0
            for t in range(P):
                body = res.body(t); st = res.stats(t)
                hs.append(hashlib.sha256(body).hexdigest() + hashlib.sha256(st.tobytes()).hexdigest())
                rows.append(res.rows(t)); sparse += res.sparse_rows(t)
                if kern != "rows":
                    m = np.frombuffer(body, np.uint8).reshape(res.rows(t), row_bytes)
                    keys = m[:, :8 * kw].copy().view(np.uint64).reshape(-1, kw)
                    hi, lo = keys[:, kw - 1], keys[:, 0]
                    assert np.all((hi[1:] > hi[:-1]) | ((hi[1:] == hi[:-1]) & (lo[1:] > lo[:-1]))), "rows not ascending"
                if t < 2 and res.body_from_arena(t) != body:      # (the arena and the order of its rows: what the pipeline's writer works from)
                    raise AssertionError(f"{kern}: partition {t}: arena + row order do not give the body")
                if t < oracle_parts and kern != "rows":
                    hl = [ctx.read_list(ptr, n, kw) for ptr, n in lists[t]]
                    eb, er, es = orc.merge_matrix(hl, kw, [1] * N, rec_min, 0, omode)
                    assert er == res.rows(t) and eb == body and np.array_equal(es, st), f"{kern}: partition {t} differs from the oracle"
            out[kern] = (res.kernel(), hs, rows, sparse)
            res.free()
    finally:
        if prev is None: os.environ.pop("KMX_MERGE_KERNEL", None)
        else: os.environ["KMX_MERGE_KERNEL"] = prev
    assert out["rows"][0] == "k_merge_rows" and out["cols"][0] == "k_merge_cols", [out[x][0] for x in out]
    if "pivot" in out and d <= 0.001: assert out["pivot"][0] == "k_merge_pivot"
    if lists_kind == "counted": assert out["cols"][3] > 0, "no row came out of k_cols_sparse"
    same = all(out[x][1] == out["rows"][1] for x in out)
    rep = {"workload": workload, "lists": lists_kind, "partitions": P, "samples": N, "rows_total": int(sum(out["rows"][2])), "rows_from_k_cols_sparse": int(out["cols"][3]),
           "all_kernels_equal_sha256": bool(same), "kernels": [out[x][0] for x in out], "subst_rate": d, "partitions_equal_to_oracle": oracle_parts, "keys_ascending": True}
    ctx.close()
    for x in keep:
        if hasattr(x, "close"): x.close()
    assert same, rep
    return rep


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["count", "pa63"], default="count")
    ap.add_argument("--lists", choices=["counted", "random"], default="counted")
    ap.add_argument("--subst-rate", type=float, default=float(os.environ.get("KMX_VERIFY_D", "0.001")))
    a = ap.parse_args()
    print(json.dumps(verify(a.workload, a.lists, d=a.subst_rate)))
