#!/usr/bin/env python3
"""bench.py -- k-mers merged/s of the MI355X-native kmtricks merge (libkmx, C ABI).

Workload = BASELINE.json configs[2]: 1000 synthetic samples, k=31, kmer:count:bin
(`--hard-min 2 --recurrence-min 2 --soft-min 1`), 256 minimizer partitions of a 5 Mbp
ancestor genome with substitution rate 0.001.  Partitions are
independent, so they shard over GPUs with no collective: every rank merges
`--partitions-per-gpu` partitions per step (32 = the 8-GPU sharding of the 256-partition
job; weak scaling).  A step = one kmx_merge_dev batch over this rank's partitions, inputs
(sorted per-sample count lists = .kmer file bodies) already resident in HBM.

The per-partition count lists are generated directly (the FASTQ -> super-k-mer -> count
stages are not part of the timed merge stage): each partition has G/P shared ancestor
k-mers, present in a sample with probability (1-d)^k, plus the sample's private k-mers
created by its substitutions; counts ~ 2 + Poisson-like coverage.  Random data, seeded.

Prints ONE JSON line on rank 0.
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def gen_partition(torch, dev, seed, n_samples, shared, p_present, n_private):
    """-> (records int32[R,3] on dev, offsets list[n_samples+1])  AoS: key lo, key hi, count"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    INF = (1 << 62)
    pool = torch.randint(0, 1 << 62, (shared,), generator=g, device=dev, dtype=torch.int64)
    priv = torch.randint(0, 1 << 62, (n_samples, n_private), generator=g, device=dev, dtype=torch.int64)
    mask = torch.rand((n_samples, shared), generator=g, device=dev) < p_present
    keys = torch.where(mask, pool.unsqueeze(0).expand(n_samples, shared), torch.full((), INF, device=dev, dtype=torch.int64))
    keys = torch.cat([keys, priv], dim=1)
    keys, _ = torch.sort(keys, dim=1)
    valid = keys < INF
    # drop (astronomically unlikely) duplicates inside a list to keep lists strictly ascending
    dup = torch.zeros_like(valid)
    dup[:, 1:] = keys[:, 1:] == keys[:, :-1]
    valid &= ~dup
    n_i = valid.sum(dim=1)
    flat = keys[valid]
    R = flat.numel()
    counts = torch.randint(2, 12, (R,), generator=g, device=dev, dtype=torch.int32)
    rec = torch.empty((R, 3), device=dev, dtype=torch.int32)
    rec[:, :2] = flat.view(torch.int32).view(R, 2)
    rec[:, 2] = counts
    offs = [0] + torch.cumsum(n_i, 0).tolist()
    return rec, offs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--partitions-per-gpu", type=int, default=32)
    ap.add_argument("--total-partitions", type=int, default=256)
    ap.add_argument("--genome", type=float, default=5e6)
    ap.add_argument("--subst-rate", type=float, default=0.001)
    ap.add_argument("--kmer-size", type=int, default=31)
    ap.add_argument("--rec-min", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from kmtricks_amd import lib, shard
    ctx = lib.Context(local)
    ctx.set_profiling(True)

    N, P = a.samples, a.partitions_per_gpu
    shared = int(a.genome / a.total_partitions)
    p_present = (1.0 - a.subst_rate) ** a.kmer_size
    n_private = int(round(shared * (1.0 - p_present)))
    parts, total_recs = [], 0
    # weak scaling: the job has P * world partitions, partition g belongs to rank g mod world
    for g in shard.partitions_of_rank(P * world, world, rank):
        rec, offs = gen_partition(torch, dev, 20240601 + g, N, shared, p_present, n_private)
        parts.append((rec, offs))
        total_recs += rec.shape[0]
    torch.cuda.synchronize()

    def make_tasks():
        tasks = []
        for rec, offs in parts:
            base = rec.data_ptr()
            lists = [(base + 12 * offs[i], offs[i + 1] - offs[i]) for i in range(N)]
            tasks.append(dict(lists=lists, key_words=1, soft_min=[1] * N, rec_min=a.rec_min, share_min=0,
                              mode=lib.MODE_COUNT, rows_hint=shared + 4096))
        return tasks

    tasks = ctx.prepare(make_tasks())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    kernel_ms, algo_bytes, rows_out, kernel_name = [], 0, 0, ""

    def run(n, record):
        """n steps = n kmx_merge_dev batches; batch i+1 is submitted before batch i is waited for (the host
        prepares the next batch while the GPU merges the current one, as the pipeline driver does with
        consecutive partition batches); every batch is waited for before the clock stops."""
        nonlocal algo_bytes, rows_out

        def finish(res):
            nonlocal algo_bytes, rows_out, kernel_name
            res.wait()
            if record:
                kernel_ms.append(res.kernel_ms())
                algo_bytes = sum(res.algo_bytes(t) for t in range(P))
                rows_out = sum(res.rows(t) for t in range(P))
                kernel_name = res.kernel()
            res.free()

        prev = None                      # two batches in flight (double buffering: no new device blocks)
        for _ in range(n):
            cur = ctx.merge_dev(tasks)
            if prev is not None:
                finish(prev)
            prev = cur
        if prev is not None:
            finish(prev)

    run(a.warmup, False)
    barrier()
    t0 = time.perf_counter()
    run(a.steps, True)
    barrier()
    dt = time.perf_counter() - t0
    dt, job_recs = shard.reduce_job(dist if world > 1 else None, dev, dt, float(total_recs))

    if rank == 0:
        ms_step = dt / a.steps * 1e3
        value = job_recs * a.steps / dt
        kms = sum(kernel_ms) / max(1, len(kernel_ms))
        achieved = algo_bytes / (kms * 1e-3) / 1e9 if kms > 0 else None
        out = {
            "metric": "k-mers merged/s (merge stage, sum over partitions of input records / time)",
            "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 keys / u32 counts (integer)", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: {N} samples, k={a.kmer_size}, kmer:count:bin, "
                                   f"recurrence-min {a.rec_min}, {P} of {a.total_partitions} partitions per GPU "
                                   f"(G={a.genome:.0f} bp, d={a.subst_rate})",
                       "records_per_step_per_gpu": total_recs, "rows_out_per_step_per_gpu": rows_out,
                       "parallelism": f"partitions sharded over {world} GPU(s), no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": (achieved / 8000.0) if achieved else None, "traffic": pmc_traffic(a, N, P, kernel_name),
                         "kernel": kernel_name + ("" if kernel_name == "k_merge_cols" else "<1,0>"), "kernel_ms": kms, "algo_bytes_per_launch": algo_bytes,
                         # streaming read rate of this access pattern measured on an MI355X (profiles/r01_h_fetch_calibration.txt)
                         "measured_stream_peak": 5654.0, "frac_of_measured": (achieved / 5654.0) if achieved else None},
        }
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(parts, N, a.rec_min)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def pmc_traffic(a, N, P, kernel):
    """HBM bytes per launch of the merge kernel from the committed rocprofv3 --pmc passes
    (FETCH_SIZE + WRITE_SIZE, profiles/merge_pmc.json) -- only when they were taken on this workload."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "merge_pmc.json")))
    except Exception:
        return None
    key = f"configs[2] {N}x{P} G={a.genome:.0e} d={a.subst_rate} rec_min={a.rec_min}".replace("e+0", "e")
    d = d.get(kernel, {})
    return d["fetch_bytes"] + d["write_bytes"] if d.get("workload") == key else None


def cpu_baseline(parts, N, rec_min, budget_s=12.0):
    """The oracle (a port of the reference's KmerMerger linear-scan merge, merge.hpp:183-260) timed on one
    host core over a bounded sample: whole partitions of the workload until ~budget_s seconds of CPU work."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import subprocess
    so = os.path.join(ROOT, "oracle", "libkmx_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    import orc
    total_dt, total_n, total_rows, used = 0.0, 0, 0, 0
    for rec, offs in parts:
        h = rec.cpu().numpy()
        lists = []
        for i in range(N):
            r = h[offs[i]:offs[i + 1]]
            keys = np.ascontiguousarray(r[:, :2]).view(np.uint64).reshape(-1)
            lists.append((keys, np.ascontiguousarray(r[:, 2]).view(np.uint32)))
        t0 = time.perf_counter()
        body, rows, stats = orc.merge_matrix(lists, 1, [1] * N, rec_min, 0, orc.MODE_COUNT)
        total_dt += time.perf_counter() - t0
        total_n += int(offs[-1]); total_rows += rows; used += 1
        if total_dt >= budget_s:
            break
    return {"value": total_n / total_dt, "unit": "k-mers/s", "cores": 1, "kind": "port",
            "sample": f"{used} of the {len(parts)} partitions of the workload ({total_n} input records, {total_rows} rows out), "
                      f"oracle linear-scan merge, {total_dt:.1f} s"}


if __name__ == "__main__":
    main()
