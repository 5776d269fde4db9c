#!/usr/bin/env python3
"""Full-size cross-check on the bench workload (BASELINE configs[2], 32 partitions x 1000 samples on one GPU): the
matrix bodies and statistics produced by k_merge_cols, k_merge_pivot and k_merge_rows are compared byte for byte (sha256 per
partition), rows are checked ascending, and the first partition is compared with the oracle."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
from kmtricks_amd import lib
import orc

N, P, G, d, K = 1000, 32, 5e6, float(os.environ.get("KMX_VERIFY_D", "0.001")), 31
dev = torch.device("cuda", 0)
shared = int(G / 256); pp = (1.0 - d) ** K; npriv = int(round(shared * (1.0 - pp)))
parts = [bench.gen_partition(torch, dev, 20240601 + g, N, shared, pp, npriv) for g in range(P)]
torch.cuda.synchronize()
ctx = lib.Context(0)
tasks = [dict(lists=[(rec.data_ptr() + 12 * offs[i], offs[i + 1] - offs[i]) for i in range(N)], key_words=1, soft_min=[1] * N,
              rec_min=2, share_min=0, mode=lib.MODE_COUNT, rows_hint=shared + 4096) for rec, offs in parts]
out = {}
for kern in ("rows", "pivot", "cols"):
    os.environ["KMX_MERGE_KERNEL"] = kern
    res = ctx.merge_dev(tasks); res.wait()
    hs, rows = [], []
    for t in range(P):
        body = res.body(t); st = res.stats(t)
        hs.append(hashlib.sha256(body).hexdigest() + hashlib.sha256(st.tobytes()).hexdigest())
        rows.append(res.rows(t))
        if kern != "rows":
            m = np.frombuffer(body, np.uint8).reshape(res.rows(t), 8 + 4 * N)
            keys = m[:, :8].copy().view(np.uint64).ravel()
            assert np.all(keys[1:] > keys[:-1])
        if t == 0 and kern != "rows":
            rec, offs = parts[0]; h = rec.cpu().numpy()
            lists = [(np.ascontiguousarray(h[offs[i]:offs[i + 1], :2]).view(np.uint64).reshape(-1),
                      np.ascontiguousarray(h[offs[i]:offs[i + 1], 2]).view(np.uint32)) for i in range(N)]
            eb, er, es = orc.merge_matrix(lists, 1, [1] * N, 2, 0, orc.MODE_COUNT)
            assert er == res.rows(0) and eb == body and np.array_equal(es, st)
    out[kern] = (res.kernel(), hs, rows)
    res.free()
if d == 0.001:      # (more divergent cohorts: the forced kernels may hand tasks down -- the bodies must agree all the same)
    assert out["rows"][0] == "k_merge_rows" and out["pivot"][0] == "k_merge_pivot" and out["cols"][0] == "k_merge_cols", [out[k][0] for k in out]
same = out["rows"][1] == out["pivot"][1]
same_cols = out["rows"][1] == out["cols"][1]
print(json.dumps({"partitions": P, "samples": N, "rows_total": int(sum(out["pivot"][2])), "pivot_equals_rows_sha256": bool(same),
                  "cols_equals_rows_sha256": bool(same_cols), "kernels": [out[k][0] for k in ("rows", "pivot", "cols")], "subst_rate": d, "partition0_equals_oracle": True, "keys_ascending": True}))
assert same and same_cols
