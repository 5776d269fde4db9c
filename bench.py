#!/usr/bin/env python3
"""bench.py -- k-mers merged/s of the MI355X-native kmtricks merge (libkmx, C ABI).

Default workload = BASELINE.json configs[2]: 1000 synthetic samples, k=31, kmer:count:bin
(`--hard-min 2 --recurrence-min 2 --soft-min 1`), 256 minimizer partitions (static repartition) of a
5 Mbp ancestor genome with substitution rate 0.001.  Partitions are independent, so
they shard over GPUs with no collective: every rank merges `--partitions-per-gpu` partitions per step
(32 = the 8-GPU sharding of the 256-partition job; weak scaling).  A step = one kmx_merge_dev batch over
this rank's partitions, inputs (sorted per-sample count lists = .kmer file bodies) resident in HBM.

Where the lists come from (`--lists`):
  counted (default)  the product's own count stage: every sample's genome (the ancestor with its substitutions) goes
                     through kmx_superk_partition + kmx_count_batch, and the (k-mer, count) lists of this rank's
                     partitions are what the merge is timed on -- the key distribution of real minimizer partitions
                     (setup, not timed: about a minute for 1000 samples);
  random             uniform random 62-bit keys with the same sharing model (G/P shared k-mers present with
                     probability (1-d)^k, the rest private) -- the round-1 generator, seconds to set up.

Other workloads (`--workload`), each with its own roofline line:
  bf     configs[1]: 100 samples, hash:bf:bin, bloom 1e8, 32 partitions on one GPU
  pa63   configs[4]: 500 samples, k=63 (128-bit keys), kmer:pa:bin, 32 of 256 partitions per GPU and step
  bft    configs[3]: 2500 samples, hash:bft:bin --soft-min 2 --share-min 1 (merge + on-device transpose; with more than
         one rank the per-sample Bloom rows are exchanged with one all-to-all over RCCL, kmtricks_amd/shard.py)

  pipeline  the metric's second half: `kmx pipeline` end to end (FASTA files in, matrix files out) on configs[2]'s cohort at the
         size the GPU box's disk takes (1000 samples x 1 Mbp, 256 partitions), wall clock and k-mers merged/s end to end, with
         the oracle's split + count + merge on the host cores over a bounded sample of the same files beside it
  all    (default on one GPU) the headline count workload first -- its fields are the line's own --, then count_stage, bf, bft, pa63,
         count_200 and pipeline: their lines go into "workloads" / "pipeline" of the ONE JSON line

Prints ONE JSON line on rank 0.
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# ---------------------------------------------------------------------------------------------- generators
def gen_partition(torch, dev, seed, n_samples, shared, p_present, n_private, kw=1):
    """random keys -> (records int32[R, 2*kw+1] on dev, offsets list[n_samples+1])  AoS: key words (low first), count"""
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
    dup = torch.zeros_like(valid)                     # (astronomically unlikely) duplicates inside a list
    dup[:, 1:] = keys[:, 1:] == keys[:, :-1]
    valid &= ~dup
    n_i = valid.sum(dim=1)
    flat = keys[valid]
    R = flat.numel()
    counts = torch.randint(2, 12, (R,), generator=g, device=dev, dtype=torch.int32)
    rec = torch.zeros((R, 2 * kw + 1), device=dev, dtype=torch.int32)
    if kw == 1:
        rec[:, :2] = flat.view(torch.int32).view(R, 2)
    else:
        # 128-bit keys that still ascend: the random value is the HIGH word (most significant first, kmer.hpp:262-268),
        # the low word a mix of it
        rec[:, 2:4] = flat.view(torch.int32).view(R, 2)
        rec[:, :2] = (flat * 0x1E3779B97F4A7C15 + 12345).view(torch.int32).view(R, 2)
    rec[:, 2 * kw] = counts
    offs = [0] + torch.cumsum(n_i, 0).tolist()
    return rec, offs


def gen_hash_partition(torch, dev, g, N, per_list, lo, W, p_shared=0.969):
    """hash-mode lists of one partition: window hashes are uniform in [lo, lo + W) whatever the k-mers are"""
    pool = torch.randint(lo, lo + W, (per_list,), generator=g, device=dev, dtype=torch.int64)
    recs, offs = [], [0]
    for _ in range(N):
        keep = torch.rand(per_list, generator=g, device=dev) < p_shared
        priv = torch.randint(lo, lo + W, (int(per_list * (1 - p_shared)),), generator=g, device=dev, dtype=torch.int64)
        h = torch.unique(torch.cat([pool[keep], priv]))                  # sorted, distinct (colliding k-mers are summed)
        r = torch.empty((h.numel(), 3), device=dev, dtype=torch.int32)
        r[:, :2] = h.view(torch.int32).view(-1, 2)
        r[:, 2] = torch.randint(1, 12, (h.numel(),), generator=g, device=dev, dtype=torch.int32)
        recs.append(r); offs.append(offs[-1] + h.numel())
    return torch.cat(recs), offs


def xxh64_u32(v):
    """XXH64(&v, 4, seed 0) of a numpy uint32 array (static repartition, repartition.hpp:45-56)"""
    import numpy as np
    P1, P2, P3, P5 = (np.uint64(x) for x in (0x9E3779B185EBCA87, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9, 0x27D4EB2F165667C5))
    with np.errstate(over="ignore"):
        h = P5 + np.uint64(4)
        h = h ^ (v.astype(np.uint64) * P1)
        h = ((h << np.uint64(23)) | (h >> np.uint64(41))) * P2 + P3
        h ^= h >> np.uint64(33); h *= P2; h ^= h >> np.uint64(29); h *= P3; h ^= h >> np.uint64(32)
    return h


def gen_counted(ctx, lib, N, k, genome, d, total_parts, my_parts, seed, log, all_lists=None):
    """Lists produced by the product's count stage: sample i = ancestor genome with i.i.d. substitutions at rate d (PCG64 seeds of
    SURVEY 8d), given twice so that every k-mer passes --hard-min 2; split with the static repartition of `total_parts`
    partitions and counted in one call with the results left in HBM (kmx_count_reads_dev into a kmx_store: what `kmx pipeline`
    does); only this rank's partitions are used.  -> (store, lists[j][i] = (device pointer, records) for partition my_parts[j])"""
    import numpy as np
    m = 10
    table = (xxh64_u32(np.arange(4 ** m, dtype=np.uint32)) % np.uint64(total_parts)).astype(np.uint16)
    anc = np.random.Generator(np.random.PCG64(seed)).integers(0, 4, genome, dtype=np.uint8)
    letters = np.frombuffer(b"ACGT", np.uint8)
    store = lib.Store(ctx.device, limit_bytes=200 << 30)
    lists = [[] for _ in my_parts]
    L = 2000
    starts = np.arange(0, genome - k + 1, L)
    win = starts[:, None] + np.arange(L + k - 1)[None, :]                # (the last window runs into a tail of N's: no k-mers there)
    offs = (np.arange(2 * len(starts) + 1, dtype=np.uint64) * np.uint64(L + k - 1))
    t0 = time.perf_counter()
    for i in range(N):
        rng = np.random.Generator(np.random.PCG64(seed + 1 + i))
        gsm = anc.copy()
        pos = np.nonzero(rng.random(genome) < d)[0]
        gsm[pos] = (gsm[pos] + rng.integers(1, 4, len(pos), dtype=np.uint8)) & 3
        # the genome as overlapping 2 kb windows (every k-mer exactly once; a wave per read in the split), twice
        seq = np.concatenate([letters[gsm], np.full(L, ord("N"), np.uint8)])[win].tobytes()
        ls, _, _ = ctx.count_reads_dev((seq + seq, offs), k, m, table, total_parts, 2, [store])
        for j, p in enumerate(my_parts):
            lists[j].append(ls[p])
        if all_lists is not None:      # (every partition of the job: the whole-job figure of a one-GPU run)
            for p in range(total_parts):
                all_lists[p].append(ls[p])
        if log and (i + 1) % 100 == 0:
            print(f"[bench] counted lists: {i + 1}/{N} samples, {time.perf_counter() - t0:.1f} s, {store.used() / 1e9:.1f} GB resident", file=sys.stderr, flush=True)
    return store, lists


# ---------------------------------------------------------------------------------------------- merge workloads
def merge_workload(env, a, wl, lists_kind, want_cpu):
    """One `--workload` of the merge stage: builds its lists on this rank's GPU, times a.steps batches, returns the JSON fields
    (on rank 0; None elsewhere)."""
    torch, dist, lib, shard = env["torch"], env["dist"], env["lib"], env["shard"]
    rank, world, local, dev = env["rank"], env["world"], env["local"], env["dev"]
    import numpy as np
    ctx = lib.Context(local)
    ctx.set_profiling(True)
    # COUNT / PA: the line's own step runs the library's default -- rows written at their final place (kmx_set_file_order on: the arena
    # IS the matrix body, the ascending row stream the reference writes, merge.hpp:262-272).  The pair's speed with the rows left where
    # the kernels put them (the headline of rounds 1-4) is timed right behind it and reported in roofline.rows_left_in_arena.
    # count_200: configs[2]'s count rows for a cohort of 200 samples (round 6: every real cohort below 257 samples is k_merge_rows' -- its
    # build for up to 256 lists, merge_rows_small.hip); the same code path as "count" otherwise
    small = wl == "count_200"
    wide = wl == "count_k96"      # round 6: a cohort of 1000 samples at k = 96 (Kmer<96>, three-word keys: the reference's KMER_LIST "32 64 96 128")
    if small or wide:
        wl = "count"
    defaults = {"count": (200 if small else 1000, 32, 96 if wide else 31, 2), "bf": (100, 32, 31, 1), "pa63": (500, 32, 63, 1), "bft": (2500, 4, 31, 1)}[wl]
    N = defaults[0] if small else (a.samples or defaults[0])
    P = a.partitions_per_gpu or defaults[1]
    k = defaults[2]
    rec_min = a.rec_min if a.rec_min >= 0 else defaults[3]
    kw = (k + 31) // 32
    total_parts = a.total_partitions if wl != "bf" else 32
    my_parts = shard.partitions_of_rank(P * world, world, rank)        # weak scaling: the job has P * world partitions
    genome = int(a.genome)
    shared = genome // total_parts
    p_present = (1.0 - a.subst_rate) ** k
    n_private = int(round(shared * (1.0 - p_present)))
    keep, tasks_d, label, mode = [], [], "", lib.MODE_COUNT      # keep: whatever owns the lists' device memory
    job_lists = None                                              # (one GPU, headline workload: the lists of EVERY partition of the job)
    host_lists = None                                             # j -> [(keys, counts)] of partition my_parts[j] on the host (cpu_baseline)
    W = 0
    rb = 8 * kw + 4
    if wl in ("count", "pa63"):
        mode = lib.MODE_COUNT if wl == "count" else lib.MODE_PA
        if lists_kind == "counted":
            job_lists = [[] for _ in range(total_parts)] if (wl == "count" and not small and not wide and world == 1 and a.whole_job) else None
            store, lists = gen_counted(ctx, lib, N, k, genome, a.subst_rate, total_parts, my_parts, 20240601, rank == 0, job_lists)
            keep.append(store)
            def host_lists(j):
                return [ctx.read_list(ptr, n, kw) for ptr, n in lists[j]]
        else:
            parts = [gen_partition(torch, dev, 20240601 + g, N, shared, p_present, n_private, kw) for g in my_parts]
            keep.append(parts)
            lists = [[(rec.data_ptr() + rb * offs[i], offs[i + 1] - offs[i]) for i in range(N)] for rec, offs in parts]
            def host_lists(j):
                rec, offs = parts[j]
                h = rec.cpu().numpy().view(np.uint32)
                return [(np.ascontiguousarray(h[offs[i]:offs[i + 1], :2 * kw]).view(np.uint64).reshape(-1, kw), np.ascontiguousarray(h[offs[i]:offs[i + 1], 2 * kw])) for i in range(N)]
        for ls in lists:
            tasks_d.append(dict(lists=ls, key_words=kw, soft_min=[1] * N, rec_min=rec_min, share_min=a.share_min, mode=mode))      # (no rows_hint: libkmx sizes the arenas from the batches it has seen)
        label = (f"BASELINE configs[{2 if wl == 'count' else 4}]{' at k = 96' if wide else ''}: {N} samples, k={k}, kmer:{'count' if wl == 'count' else 'pa'}:bin, recurrence-min {rec_min}, "
                 + (f"share-min {a.share_min}, " if a.share_min else "")
                 + f"{P} of {total_parts} partitions per GPU (G={genome} bp, d={a.subst_rate}), lists: "
                 + ("count stage output resident in HBM (kmx_count_reads_dev)" if lists_kind == "counted" else "random 62-bit keys"))
    else:
        bloom = int(a.bloom) if a.bloom else (100_000_000 if wl == "bf" else 1_000_000_000)
        W = ((bloom + total_parts - 1) // total_parts + 63) // 64 * 64      # hash.hpp:31-40
        g = torch.Generator(device=dev); g.manual_seed(20240601 + rank)
        mode = lib.MODE_BF if wl == "bf" else lib.MODE_BFT
        smin, share = (1, 0) if wl == "bf" else (2, 1)
        parts = []
        for p in my_parts:
            rec, offs = gen_hash_partition(torch, dev, g, N, shared, W * p, W)
            parts.append((rec, offs))
            base = rec.data_ptr()
            tasks_d.append(dict(lists=[(base + 12 * offs[i], offs[i + 1] - offs[i]) for i in range(N)], key_words=1, soft_min=[smin] * N,
                                rec_min=rec_min, share_min=share, mode=mode, lower=W * p, upper=W * (p + 1) - 1))
        keep.append(parts)
        def host_lists(j):
            rec, offs = parts[j]
            h = rec.cpu().numpy().view(np.uint32)
            return [(np.ascontiguousarray(h[offs[i]:offs[i + 1], :2]).view(np.uint64).reshape(-1), np.ascontiguousarray(h[offs[i]:offs[i + 1], 2])) for i in range(N)]
        label = (f"BASELINE configs[{1 if wl == 'bf' else 3}]: {N} samples, k=31, hash:{wl}:bin, bloom {bloom:.0e} / {total_parts} partitions "
                 f"(window {W} bits), soft-min {smin}, share-min {share}, {P} partitions per GPU and step (G={genome} bp)")
    total_recs = sum(n for d_ in tasks_d for _, n in d_["lists"])
    sync = env.get("sync", torch.cuda.synchronize)
    empty_cache = env.get("empty_cache", torch.cuda.empty_cache)
    sync()
    empty_cache()           # (the generators' scratch goes back to the device: libkmx allocates beside torch)
    tasks = ctx.prepare(tasks_d)

    def barrier():
        sync()
        if world > 1:
            dist.barrier()
        sync()

    kernel_ms, tr_ms, kernel_name = [], [], ""
    algo_bytes = rows_out = 0
    xbuf = None
    if wl == "bft" and world > 1:
        n8 = (N + 7) // 8 * 8
        xbuf = [torch.empty((n8, W // 8), dtype=torch.uint8, device=dev) for _ in range(P)]

    def run(n, record, batches=None):
        """n steps = n kmx_merge_dev batches; batch i+1 is submitted before batch i is waited for (the host
        prepares the next batch while the GPU merges the current one, as the pipeline driver does with
        consecutive partition batches); every batch is waited for before the clock stops.
        batches: a list of prepared task batches run once each instead (the whole job of a one-GPU run)."""
        nonlocal algo_bytes, rows_out

        def finish(res):
            nonlocal algo_bytes, rows_out, kernel_name
            res.wait()
            if xbuf is not None:      # per-sample Bloom rows to their owners: the one collective of the path (RCCL all-to-all)
                for t in range(P):
                    res.body_to_device(t, xbuf[t].data_ptr(), xbuf[t].numel())
                shard.bloom_exchange(dist, xbuf, N, P * world, world, rank)
            if record:
                kernel_ms.append(res.kernel_ms())
                if wl == "bft":
                    tr_ms.append(res.transpose_ms())
                algo_bytes = sum(res.algo_bytes(t) for t in range(P))
                rows_out = sum(res.rows(t) for t in range(P))
                kernel_name = res.kernel()
            res.free()

        prev = None                      # two batches in flight (double buffering: no new device blocks)
        for it in range(n if batches is None else len(batches)):
            cur = ctx.merge_dev(tasks if batches is None else batches[it])
            if prev is not None:
                finish(prev)
            prev = cur
        if prev is not None:
            finish(prev)

    # setup, not warm-up: libkmx sizes its row arenas from the batches it has completed (the first ones of a cohort run twice, and
    # the first arenas of the final size are fresh hipMallocs): three batches, one at a time, before the W warm-up steps
    for _ in range(3):
        run(1, False)
    run(a.warmup, False)
    barrier()
    t0 = time.perf_counter()
    run(a.steps, True)
    barrier()
    dt = time.perf_counter() - t0
    dt, job_recs = shard.reduce_job(dist if world > 1 else None, dev, dt, float(total_recs))
    # ---- on one GPU, the WHOLE job: all partitions of configs[2] (8 batches of 32 on one MI355X), rows in file order ----
    whole_job = None
    if job_lists is not None and rank == 0:
        per = P
        jb = [ctx.prepare([dict(lists=job_lists[p], key_words=kw, soft_min=[1] * N, rec_min=rec_min, share_min=a.share_min, mode=mode)
                           for p in range(b0, min(total_parts, b0 + per))]) for b0 in range(0, total_parts, per)]
        run(0, False, jb)      # (once untimed: the context's pool then holds the blocks two batches in flight need)
        sync()
        if os.environ.get("KMX_TRACE_ALLOC"):
            print("[bench] whole job: the timed pass starts", file=sys.stderr, flush=True)
        t2 = time.perf_counter()
        run(0, False, jb)
        sync()
        dt_job = time.perf_counter() - t2
        recs_job = sum(n for ls in job_lists for _, n in ls)
        whole_job = {"partitions": total_parts, "batches": len(jb), "records": recs_job, "wall_ms": dt_job * 1e3, "value": recs_job / dt_job,
                     "what": f"every partition of the job merged on this one GPU, {len(jb)} batches of {per}, two in flight, rows in file order, results left in HBM"}
        del jb
    # ---- the same steps with the rows left where the kernels put them (kmx_set_file_order off: the row keys' rows, then the rows out of
    #      k_cols_sparse, each list ascending -- NOT the matrix body; the headline of rounds 1-4, kept for continuity) ----
    arena = None
    gather_ms = order_ms = None
    if wl in ("count", "pa63") and hasattr(ctx, "set_file_order"):
        ctx.set_file_order(False)
        head_ms, head_name, head_algo, head_rows = list(kernel_ms), kernel_name, algo_bytes, rows_out
        kernel_ms.clear()
        for _ in range(3):
            run(1, False)
        run(a.warmup, False)
        barrier()
        t1 = time.perf_counter()
        run(a.steps, True)
        barrier()
        dt_ar = time.perf_counter() - t1
        dt_ar, _ = shard.reduce_job(dist if world > 1 else None, dev, dt_ar, float(total_recs))
        ar_ms = sum(kernel_ms) / max(1, len(kernel_ms))
        arena = {"kernel_ms": ar_ms, "ms_per_step": dt_ar / a.steps * 1e3, "value": job_recs * a.steps / dt_ar, "kernel": kernel_name,
                 "frac": (algo_bytes / (ar_ms * 1e-3) / 1e9 / 8000.0) if ar_ms > 0 else None,
                 "traffic": pmc_traffic(wl, lists_kind, N, P, kernel_name)}
        # a consumer of THAT result who wants the body in file order on the device (kmx_result_body_dev) pays a device-to-device pass once
        # per result (k_cols_offsets + k_cols_gather); one who places the rows itself needs only their order (kmx_result_copy_order)
        if hasattr(lib, "_lib"):
            import numpy as _np
            res = ctx.merge_dev(tasks); res.wait()
            sync()
            tg = time.perf_counter()
            for t in range(P):      # (queued for every task first, as a writer does: the passes run back to back)
                ctx._check(lib._lib.kmx_result_prepare_body(res._h, t), "kmx_result_prepare_body")
            for t in range(P):
                res.body_dev(t)
            sync()
            gather_ms = (time.perf_counter() - tg) * 1e3
            res.free()
            res = ctx.merge_dev(tasks); res.wait()
            bufs = [_np.zeros(max(1, res.rows(t)), _np.uint32) for t in range(P)]
            sync()
            tg = time.perf_counter()
            for t in range(P):
                ctx._check(lib._lib.kmx_result_copy_order(res._h, t, bufs[t].ctypes.data), "kmx_result_copy_order")
            order_ms = (time.perf_counter() - tg) * 1e3
            res.free()
            arena["file_order_gather_ms"] = gather_ms; arena["row_order_ms"] = order_ms
            arena["frac_with_gather"] = (algo_bytes / ((ar_ms + gather_ms) * 1e-3) / 1e9 / 8000.0) if ar_ms > 0 else None
        kernel_ms[:] = head_ms; kernel_name = head_name; algo_bytes = head_algo; rows_out = head_rows
        ctx.set_file_order(True)

    out = None
    if rank == 0:
        ms_step = dt / a.steps * 1e3
        value = job_recs * a.steps / dt
        kms = sum(kernel_ms) / max(1, len(kernel_ms))
        # BFT: the launch priced is merge + transposes (the roofline's algorithmic bytes are records in + transposed matrix out)
        if wl == "bft" and tr_ms and min(tr_ms) >= 0:      # (builds that transpose in a second pass)
            kms += sum(tr_ms) / len(tr_ms)
        achieved = algo_bytes / (kms * 1e-3) / 1e9 if kms > 0 else None
        kname = kernel_name + (" + k_cols_sparse" if kernel_name == "k_merge_cols" else "")
        out = {
            "metric": "k-mers merged/s (merge stage, sum over partitions of input records / time)",
            "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("u128" if kw == 2 else "u64") + " keys / u32 counts (integer)", "data": "synthetic",
            "config": {"workload": label,
                       "records_per_step_per_gpu": total_recs, "rows_out_per_step_per_gpu": rows_out,
                       "parallelism": f"partitions sharded over {world} GPU(s), " + ("per-sample Bloom rows exchanged by one RCCL all-to-all" if xbuf is not None else "no collective")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": (achieved / 8000.0) if achieved else None,
                         "traffic": pmc_traffic(wl, lists_kind, N, P, kernel_name + (":file_order" if wl in ("count", "pa63") else "")),
                         "kernel": kname, "kernel_ms": kms, "algo_bytes_per_launch": algo_bytes,
                         "rows": "at their final place (kmx_set_file_order on, the library's default: the arena is the matrix body)" if wl in ("count", "pa63") else "the matrix body",
                         "whole_job": whole_job, "rows_left_in_arena": arena,
                         # streaming read rate of this access pattern measured on an MI355X (profiles/r01_h_fetch_calibration.txt)
                         "measured_stream_peak": 5654.0, "frac_of_measured": (achieved / 5654.0) if achieved else None},
        }
        if want_cpu:
            out["cpu_baseline"] = cpu_baseline(host_lists, len(tasks_d), N, kw, tasks_d, mode, with_io=(wl == "count"))
    ctx.close()
    for x in keep:
        if hasattr(x, "close"):
            x.close()
    del keep, tasks, tasks_d
    empty_cache()
    return out


# ---------------------------------------------------------------------------------------------- count stage
def count_stage_workload(env, a):
    """The count stage by itself, as `kmx pipeline` runs it for configs[2]: one synthetic sample (5 Mbp genome, 150-bp error-free reads at
    6x = 30 Mbases, k = 31, m = 10, 256 static partitions) through kmx_count_reads_dev (split + count in one call, the sample's bases sent
    ahead with kmx_reads_upload as the pipeline sends them, results left in HBM).  Live: device time per call from HIP events on the context's own stream, wall clock
    per call.  Roofline: B = super-k-mer bytes + 12 per distinct solid k-mer over the kernels' time -- the stage is a chain
    of ~25 instruction-bound kernels, so next to it, per kernel, what share of the chip's instruction issue it uses (from the committed
    rocprofv3 passes, profiles/count_stage_kernels.json: scripts/r6_count_stage_profile.sh)."""
    torch, lib = env["torch"], env["lib"]
    import numpy as np
    rng = np.random.default_rng(20240601)
    G, L, COV, K, M, P = 5_000_000, 150, 6, 31, 10, a.total_partitions
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=G)
    comp = np.zeros(256, np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
    n_reads = G * COV // L
    starts = rng.integers(0, G - L, n_reads)
    reads = genome[starts[:, None] + np.arange(L)[None, :]]
    rc = rng.random(n_reads) < 0.5
    reads[rc] = comp[reads[rc]][:, ::-1]
    blob = reads.tobytes(); offs = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L))
    table = (xxh64_u32(np.arange(4 ** M, dtype=np.uint32)) % np.uint64(P)).astype(np.uint16)
    ctx = lib.Context(env["local"]); store = lib.Store(env["local"])
    st = torch.cuda.ExternalStream(ctx.stream() if callable(ctx.stream) else ctx.stream, device=env["dev"])
    # the sample's bases are resident in HBM when the timed region starts (kmx_reads_upload: what `kmx pipeline` does with the NEXT sample
    # while this one is counted); the same call with the bases handed over as a host buffer (30 MB over PCIe inside the call) beside it
    handle = ctx.upload_reads(blob, offs)      # (the offsets behind the bases in the page-locked block, as the pipeline's reader leaves them: no staging copy inside the call)
    for _ in range(3):
        ctx.count_reads_dev((blob, offs), K, M, table, P, 2, [store], resident=handle)
    reps, dev_ms, wall, dev_ms_up = 10, [], [], []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(st)
        lists, nk, _ = ctx.count_reads_dev((blob, offs), K, M, table, P, 2, [store], resident=handle)
        e1.record(st); e1.synchronize(); wall.append(time.perf_counter() - t0); dev_ms.append(e0.elapsed_time(e1))
    ctx.release_reads(handle)
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        ctx.count_reads_dev((blob, offs), K, M, table, P, 2, [store])
        e1.record(st); e1.synchronize(); dev_ms_up.append(e0.elapsed_time(e1))
    got = ctx.count_reads((blob, offs), K, M, table, P, 2, streams=True)
    sk_bytes = sum(len(x) for x in got[2]); distinct = sum(n for _, n in lists)
    B = sk_bytes + 12 * distinct
    d_ms, w_ms = sorted(dev_ms)[reps // 2], sorted(wall)[reps // 2] * 1e3
    out = {"metric": "count stage: bases counted/s (one sample per call, results resident in HBM)", "value": n_reads * L / (w_ms * 1e-3), "unit": "bases/s",
           "ms_per_sample_wall": w_ms, "ms_per_sample_device": d_ms, "ms_per_sample_device_bases_from_host": sorted(dev_ms_up)[len(dev_ms_up) // 2], "kmers_per_s": sum(nk) / (w_ms * 1e-3), "higher_is_better": True, "data": "synthetic", "dtype": "u64 keys / u32 counts (integer)",
           "config": {"workload": f"configs[2]'s sample: {G} bp genome, {n_reads} reads of {L} bp ({n_reads * L} bases), k={K}, m={M}, {P} static partitions, --hard-min 2; "
                                  f"{sum(nk)} k-mers, {sk_bytes} super-k-mer bytes, {distinct} distinct solid k-mers"},
           "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0, "algo_bytes_per_sample": B, "achieved": B / (d_ms * 1e-3) / 1e9, "frac": B / (d_ms * 1e-3) / 1e9 / 8000.0,
                        "kernel": "the stage's kernels (one walk over the reads, counting sort of the super-k-mer descriptors, decode, partition-local sample sort, wave sort + run lengths, "
                                  "compaction into the store) between two HIP events on the context's stream, the sample's bases resident in HBM; one read-back at the call's end, no library call",
                        "note": "instruction-bound kernels: see kernels[].issue_frac"}}
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "count_stage_kernels.json")))
        out["roofline"]["kernels_us_per_sample_profiled"] = prof["kernels_us_per_sample"]
        out["roofline"]["launches_per_sample"] = prof["launches_per_sample"]
        out["roofline"]["library_us_per_sample"] = prof["library_us_per_sample"]
        out["roofline"]["frac_of_kernel_time_profiled"] = prof["hbm_frac_algorithmic"]
        out["roofline"]["kernels"] = [{kk: r[kk] for kk in ("kernel", "calls_per_sample", "us_per_sample", "issue_frac", "hbm_frac")} for r in prof["kernels"][:6]]
        out["roofline"]["source"] = "profiles/count_stage_kernels.json"
    except Exception:
        pass
    store.close(); ctx.close()
    return out


# ---------------------------------------------------------------------------------------------- end to end
def _pipeline_make_sample(args):
    """one sample of SURVEY 8d's cohort as a FASTA file of error-free 150-bp reads at the given coverage, random strands"""
    import numpy as np
    s, G, d, cov, tmp = args
    L = 150
    ref = np.random.Generator(np.random.PCG64(20240601)).integers(0, 4, G, dtype=np.uint8)
    rng = np.random.Generator(np.random.PCG64(20240601 + 1 + s))
    g = ref
    pos = np.nonzero(rng.random(G) < d)[0]
    g[pos] = (g[pos] + rng.integers(1, 4, len(pos), dtype=np.uint8)) & 3
    n_reads = G * cov // L
    starts = rng.integers(0, G - L, n_reads)
    reads = g[starts[:, None] + np.arange(L)[None, :]]
    rc = rng.random(n_reads) < 0.5
    comp = np.array([2, 3, 0, 1], np.uint8)           # A0 C1 T2 G3 -> T G A C
    reads[rc] = comp[reads[rc]][:, ::-1]
    letters = np.frombuffer(b"ACTG", np.uint8)
    path = os.path.join(tmp, f"S{s:04d}.fa")
    lines = np.empty((n_reads, L + 4), np.uint8)
    lines[:, 0] = ord(">"); lines[:, 1] = ord("r"); lines[:, 2] = ord("\n"); lines[:, 3:3 + L] = letters[reads]; lines[:, 3 + L] = ord("\n")
    lines.tofile(path)
    return path


def pipeline_workload(a, n_gpus=1, genome=None, tmp_root=None, cpu=True):
    """`kmx pipeline` end to end on SURVEY 8d's cohort: FASTA files on local disk in, the run directory (matrices) out; wall clock
    around the process.  Then the oracle (split + count + merge, a port of the reference's CPU path) over a bounded sample of the
    same files on the host cores.  genome / tmp_root: another genome size, files in another place (the full-size run keeps its
    30 GB of FASTA and ~100 GB of matrices in a RAM file system: the box's disk holds 79 GB)."""
    import shutil, subprocess, tempfile
    import multiprocessing
    Pool = multiprocessing.get_context("spawn").Pool      # (not fork: this process may hold a HIP runtime and tens of GB of mappings)
    S, G, P, k = a.pipeline_samples, int(genome or a.pipeline_genome), a.total_partitions, 31
    tmp = tempfile.mkdtemp(prefix="kmx_bench_", dir=tmp_root or a.tmp)
    try:
        nproc = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        t0 = time.perf_counter()
        with Pool(min(64, nproc)) as pool:
            paths = pool.map(_pipeline_make_sample, [(s, G, a.subst_rate, 6, tmp) for s in range(S)], chunksize=4)
        gen_s = time.perf_counter() - t0
        with open(os.path.join(tmp, "in.fof"), "w") as fof:
            for s, pth in enumerate(paths):
                fof.write(f"S{s:04d}: {pth}\n")
        run = os.path.join(tmp, "run")
        threads = min(64, nproc)
        cmd = [os.path.join(ROOT, "kmtricks_amd", "kmx"), "pipeline", "--file", os.path.join(tmp, "in.fof"), "--run-dir", run, "--kmer-size", str(k),
               "--mode", "kmer:count:bin", "--hard-min", "2", "--recurrence-min", "2", "--nb-partitions", str(P), "--static-repart",
               "-t", str(threads), "--gpus", str(n_gpus)]
        # the bench's warm-up, as its W untimed steps in front of the K timed ones: the same command once, untimed.  The boxes are fresh
        # virtual machines whose memory the host backs on first touch: the first job that writes 92 GB of matrices into the RAM file system
        # and page-locks its buffers is 0.3-1 s slower in EACH stage than every later one (reader threads 33 s of CPU against 15, the merge
        # stage's writes 2.8-3.2 s against 1.8), and a fresh box's HBM is cleared on first touch as well (--pipeline-warmup-samples n: over
        # the first n samples only; 0: none -- both the warm-up and the first run's penalty are then in the line's "warmup" field)
        warm_n = (S if a.pipeline_warmup_samples < 0 else min(S, a.pipeline_warmup_samples)) if genome is not None else 0
        if warm_n > 0:
            wfof = os.path.join(tmp, "warm.fof")
            with open(wfof, "w") as fof:
                for s, pth in enumerate(paths[:warm_n]):
                    fof.write(f"S{s:04d}: {pth}\n")
            cw = list(cmd); cw[cw.index("--file") + 1] = wfof; cw[cw.index("--run-dir") + 1] = run + "_warm"
            subprocess.run(cw, capture_output=True, text=True)
            shutil.rmtree(run + "_warm", ignore_errors=True)
        os.sync()
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True)
        wall = time.perf_counter() - t0
        line = [l for l in r.stderr.splitlines() if l.startswith("[kmx pipeline]")]
        if r.returncode != 0 or not line:
            return {"error": r.stderr[-1500:]}
        d = json.loads(line[-1][len("[kmx pipeline] "):])
        out_bytes = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(os.path.join(run, "matrices")) for f in fs)
        shutil.rmtree(run, ignore_errors=True)
        out = {"metric": "end-to-end wall-clock of kmx pipeline (FASTA in, matrices out)", "value": wall, "unit": "s", "higher_is_better": False,
               "kmers_merged_per_s_end_to_end": d["merge_records"] / wall, "Mbases_per_s_end_to_end": d["bases"] / wall / 1e6,
               "n_gpus": n_gpus, "data": "synthetic",
               "config": {"workload": f"BASELINE configs[2] end to end: {S} samples x {G} bp (d={a.subst_rate}, 150-bp reads at 6x, plain FASTA {'in ' + tmp_root + ' (RAM)' if tmp_root else 'on local disk'}), k=31, "
                                      f"kmer:count:bin --hard-min 2 --recurrence-min 2, {P} partitions, static repartition, {threads} host threads",
                          "command": " ".join(cmd[1:]), "bases": d["bases"], "kmers": d["kmers"], "merge_records": d["merge_records"], "matrix_bytes": out_bytes},
               "stages": {kk: d[kk] for kk in ("setup_wall_s", "count_wall_s", "merge_wall_s", "total_s", "read_s", "count_s", "merge_io_s", "merge_s", "gpu_workers", "resident_samples") if kk in d},
               "fasta_generation_s": gen_s}
        if warm_n > 0:
            out["warmup"] = f"one untimed run of the same command over {'the cohort' if warm_n == S else 'the first ' + str(warm_n) + ' samples'} in front of the timed one"
        elif genome is not None:
            out["warmup"] = "none: the timed run is the first job of its size on this box"
        if genome is None:
            # the same files at k = 96 (Kmer<128>, keys of three words: the reference's default KMER_LIST reaches 128) -- the
            # split by k_superk_wide, the decode by k_superk_decode_wide, the word-by-word sort, k_merge_rows<3>
            try:
                cmdw = list(cmd); cmdw[cmdw.index("--kmer-size") + 1] = "96"
                os.sync()
                t0 = time.perf_counter()
                rw = subprocess.run(cmdw, capture_output=True, text=True)
                wallw = time.perf_counter() - t0
                linew = [l for l in rw.stderr.splitlines() if l.startswith("[kmx pipeline]")]
                if rw.returncode != 0 or not linew:
                    out["wide_k"] = {"error": rw.stderr[-800:]}
                else:
                    dw = json.loads(linew[-1][len("[kmx pipeline] "):])
                    out["wide_k"] = {"k": 96, "key_words": 3, "value": wallw, "unit": "s", "higher_is_better": False, "Mbases_per_s_end_to_end": dw["bases"] / wallw / 1e6,
                                     "kmers": dw["kmers"], "merge_records": dw["merge_records"],
                                     "stages": {kk: dw[kk] for kk in ("setup_wall_s", "count_wall_s", "merge_wall_s", "total_s", "resident_samples") if kk in dw}}
                shutil.rmtree(run, ignore_errors=True)
            except Exception as e:
                out["wide_k"] = {"error": repr(e)}
        if not a.no_cpu_baseline and cpu:
            # the host's share: the oracle over a bounded sample of the same files, in a process of its own (a plain Python process
            # forks its workers cheaply; this one may hold a HIP runtime and tens of GB of mappings)
            try:
                out["cpu_baseline"] = pipeline_cpu_baseline(paths[:a.pipeline_cpu_samples] if a.pipeline_cpu_samples else paths, k, 10, P, 2, 2, S)
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def physical_cores():
    """one hardware thread per physical core of this process's affinity set (SMT siblings counted once)"""
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    seen, n = set(), 0
    for c in cpus:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib); n += 1
    return max(1, n), len(cpus)


def pipeline_cpu_baseline(paths, k, m, P, hard_min, rec_min, S_total):
    """cpu_baseline of the end-to-end workload: oracle/kmx_oracle_pipeline -- plain C and pthreads around the oracle, no Python in
    the loop: whole samples through the oracle's split + count, one task per sample on a pool of T threads (T = physical cores, at
    most one sample each), then one oracle merge task per partition on the same pool, in memory.  The per-core rate of the split +
    count leg (CPU seconds of the workers) must agree with one thread working alone on a sample within 2x -- else the figure is
    a harness artefact, not the CPU path (round 3's was: 0.42 Mbases/s per core under 256 Python processes)."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "kmx_oracle_pipeline")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    phys, logical = physical_cores()
    T = min(phys, len(paths))
    r = subprocess.run([exe, str(k), str(m), str(P), str(hard_min), str(rec_min), str(T)] + list(paths[:T]), capture_output=True, text=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    cw = d["split_count_wall_s"] + d["merge_wall_s"]
    ratio = d["single_core_Mbases_per_s"] / max(1e-9, d["split_count_Mbases_per_s_per_core"])
    out = {"value": d["merge_records"] / cw, "unit": "k-mers merged/s end to end", "cores": T, "kind": "port", "host_cores": logical, "physical_cores": phys,
           "wall_s_per_sample_set": cw, "split_count_s": d["split_count_wall_s"], "merge_s": d["merge_wall_s"],
           "split_count": {"Mbases_per_s_per_core": d["split_count_Mbases_per_s_per_core"], "single_core_Mbases_per_s": d["single_core_Mbases_per_s"],
                           "single_over_pooled": ratio, "Mbases_per_s": d["bases"] / d["split_count_wall_s"] / 1e6},
           "merge": {"Mrecords_per_s_per_core": d["merge_Mrecords_per_s_per_core"], "threads": d["merge_threads"]},
           "sample": f"{T} of the {S_total} samples' FASTA files: oracle/kmx_oracle_pipeline (C, pthreads): split + count of a whole sample per thread ({T} threads = one "
                     f"per physical core, host has {logical} hardware threads), then one oracle merge task per partition ({P}) on the same pool; {d['merge_records']} records "
                     f"merged, {d['rows']} rows; in memory, no intermediate files (the reference writes and re-reads super-k-mer and count files)"}
    if not (0.5 <= ratio <= 2.0):
        out["warning"] = f"pooled per-core rate and single-core rate differ by {ratio:.2f}x: the pooled figure is not the CPU path's"
    return out


# ---------------------------------------------------------------------------------------------- main
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["auto", "all", "count", "count_200", "count_k96", "bf", "pa63", "bft", "pipeline", "count_stage"], default="auto")
    ap.add_argument("--lists", choices=["counted", "random"], default="counted")
    ap.add_argument("--samples", type=int, default=0)
    ap.add_argument("--partitions-per-gpu", type=int, default=0)
    ap.add_argument("--total-partitions", type=int, default=256)
    ap.add_argument("--genome", type=float, default=5e6)
    ap.add_argument("--subst-rate", type=float, default=0.001)
    ap.add_argument("--rec-min", type=int, default=-1)
    ap.add_argument("--share-min", type=int, default=0)
    ap.add_argument("--bloom", type=float, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-whole-job", dest="whole_job", action="store_false", help="skip the one-GPU whole-job figure of the headline workload")
    ap.add_argument("--pipeline-samples", type=int, default=1000)
    ap.add_argument("--pipeline-genome", type=float, default=1e6)
    ap.add_argument("--pipeline-cpu-samples", type=int, default=0, help="samples of the end-to-end cpu_baseline (0: one per host core)")
    ap.add_argument("--pipeline-warmup-samples", type=int, default=-1, help="the full-size end-to-end run: samples of the untimed warm-up run in front of it (-1: the whole cohort, 0: none)")
    ap.add_argument("--tmp", default=None, help="directory for the end-to-end workload's files (default: the system's temporary directory)")
    ap.add_argument("--tmp-ram", default="/dev/shm", help="RAM file system for the end-to-end workload at full size (5 Mbp genomes: 130 GB of files)")
    ap.add_argument("--no-full-size", dest="full_size", action="store_false", help="skip the end-to-end run at G = 5 Mbp")
    ap.add_argument("--no-multi-gpu-pipeline", dest="multi_gpu_pipeline", action="store_false", help="several ranks: skip the `kmx pipeline --gpus N` point behind the merge steps")
    return ap.parse_args(argv)


def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    wl = a.workload
    if wl == "auto":      # one GPU: every workload in one invocation; several ranks (the scaling runs): the headline workload
        wl = "all" if world == 1 else "count"
    if wl == "pipeline":
        if rank == 0:
            print(json.dumps(pipeline_workload(a, a.gpus if world == 1 else 1)))
        return

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # (an N > 1 line is printed only by the N ranks it names: --gpus, WORLD_SIZE and the process group must agree)
    seen = dist.get_world_size() if dist is not None else 1
    if a.gpus != world or seen != world:
        if rank == 0:
            print(json.dumps({"error": f"--gpus {a.gpus}, WORLD_SIZE {world} and the process group ({seen} ranks) disagree: no line", "n_gpus": a.gpus}))
        if dist is not None:
            dist.destroy_process_group()
        sys.exit(2)
    from kmtricks_amd import lib, shard
    env = dict(torch=torch, dist=dist, lib=lib, shard=shard, rank=rank, world=world, local=local, dev=dev)
    out = run_workloads(a, wl, env)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_workloads(a, wl, env):
    """the merge workloads of one invocation on this rank; -> the JSON line's fields on rank 0 (None elsewhere).  env: torch, dist
    (None on one rank), lib (the ctypes binding of libkmx), shard, rank, world, local, dev (+ sync / empty_cache overrides: the CPU
    test of this plumbing runs it under a two-rank gloo group with a stand-in for lib)"""
    rank, world = env["rank"], env["world"]
    want_cpu = not a.no_cpu_baseline and world == 1      # (the host baseline is a 1-GPU line: with more ranks the others would wait at the barrier for it)
    extras, pipe, head = {}, None, None
    if wl == "all":
        t_all = time.perf_counter()
        # the headline workload first (round 6): on device memory nothing has been carved up yet -- behind the other workloads and the
        # end-to-end runs the same 20 steps read 3 % slower (4.95 against 4.79 ms on one box, the same build minutes apart)
        head = merge_workload(env, a, "count", a.lists, want_cpu)
        if rank == 0:
            print(f"[bench] workload count (the line's own) done, {time.perf_counter() - t_all:.0f} s", file=sys.stderr, flush=True)
        for w in ("count_stage", "bf", "bft", "pa63", "count_200"):
            try:
                extras[w] = count_stage_workload(env, a) if w == "count_stage" else merge_workload(env, a, w, "counted", want_cpu)
            except Exception as e:      # (a side line must not cost the headline one)
                extras[w] = {"error": repr(e)}
            if rank == 0:
                print(f"[bench] workload {w} done, {time.perf_counter() - t_all:.0f} s", file=sys.stderr, flush=True)
        if rank == 0:
            try:
                pipe = pipeline_workload(a)
            except Exception as e:
                pipe = {"error": repr(e)}
            print(f"[bench] workload pipeline done, {time.perf_counter() - t_all:.0f} s", file=sys.stderr, flush=True)
            # ... and at SURVEY 8d's stated size (G = 5 Mbp: 30 GB of FASTA in, ~100 GB of count matrices out -- more than the box's
            # disk): in a RAM file system when the box has one with room for it
            if a.full_size and isinstance(pipe, dict) and "error" not in pipe:
                try:
                    import shutil as _sh
                    big = a.tmp_ram if os.path.isdir(a.tmp_ram) else None
                    free = _sh.disk_usage(big).free if big else 0
                    need = int(a.pipeline_samples * 5e6 * (6 * 1.03 + 22))      # FASTA + matrices (~21 bytes per base pair of genome here)
                    if big and free > need * 1.2:
                        pipe["full_size"] = pipeline_workload(a, genome=5e6, tmp_root=big, cpu=False)
                    else:
                        pipe["full_size"] = {"skipped": f"no RAM file system with {need / 1e9:.0f} GB free at {a.tmp_ram} ({free / 1e9:.0f} GB)"}
                except Exception as e:
                    pipe["full_size"] = {"error": repr(e)}
                print(f"[bench] workload pipeline at 5 Mbp done, {time.perf_counter() - t_all:.0f} s", file=sys.stderr, flush=True)
        wl = "count"
    if wl == "count_stage":
        return count_stage_workload(env, a) if rank == 0 else None
    out = head if head is not None else merge_workload(env, a, wl, a.lists, want_cpu)
    if world > 1 and a.multi_gpu_pipeline and wl == "count" and env.get("run_pipeline", True):
        # several ranks (the scaling runs): the PRODUCT's own multi-GPU path gets a point too -- `kmx pipeline --gpus <world>`, one
        # process driving every GPU of the node (partitions p -> GPU p mod G, count lists to the merging GPU's store over xGMI),
        # started by rank 0 once every rank has let go of its device memory; the other ranks wait at the barrier
        env["dist"].barrier()
        if rank == 0:
            try:
                pipe = pipeline_workload(a, n_gpus=world, cpu=False)
            except Exception as e:
                pipe = {"error": repr(e)}
        env["dist"].barrier()
    per_rank = None
    if world > 1:
        per_rank = rank_probe(env)      # (every rank takes part: a collective)
    if rank == 0:
        # what the scaling runs need to be read without a second look: the world the ranks saw, and the one-GPU line of the last round
        # the driver recorded (BENCH_r*.json at the repository's root), so that N ranks x that value is at hand
        out["ranks"] = {"world_size": world, "backend": ("nccl (RCCL)" if world > 1 else None),
                        "devices": env["torch"].cuda.device_count() if hasattr(env["torch"], "cuda") else None}
        if per_rank is not None:
            out["ranks"]["per_rank"] = per_rank
        try:
            import glob
            last = sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json")))[-1]
            d = json.load(open(last)); pv = d.get("parsed", d)
            out["ranks"]["one_gpu_line"] = {"file": os.path.basename(last), "value": pv.get("value"), "ms_per_step": pv.get("ms_per_step"), "n_gpus": pv.get("n_gpus")}
        except Exception:
            pass
        if extras:
            out["workloads"] = extras
        if pipe is not None:
            out["pipeline"] = pipe
    return out


def rank_probe(env):
    """several ranks: what each of them bound and saw, and the path's one collective on real links -- a small hash:bft exchange
    (kmtricks_amd/shard.py: per-sample Bloom rows, one all-to-all) with rows whose content names (partition, sample), checked on the
    receiving side.  -> a list of one record per rank (gathered on every rank), or the error it died of."""
    torch, dist, shard, rank, world, local, dev = (env[k] for k in ("torch", "dist", "shard", "rank", "world", "local", "dev"))
    rec = {"rank": rank, "local_rank": local, "world_size_seen": dist.get_world_size() if dist is not None else 1}
    try:
        if hasattr(torch, "cuda") and dev.type == "cuda":
            pr = torch.cuda.get_device_properties(local)
            rec["device"] = {"index": local, "name": pr.name, "gcn_arch": getattr(pr, "gcnArchName", None), "cus": pr.multi_processor_count, "hbm_GB": round(pr.total_memory / 1e9, 1)}
        n_samples, n_parts, row_bytes = 8 * world + 3, 2 * world, 256
        mine = shard.partitions_of_rank(n_parts, world, rank)
        rows8 = (n_samples + 7) // 8 * 8
        mats = []
        for p in mine:      # row s of partition p: bytes (p * 131 + s * 7 + j) % 251
            sidx = torch.arange(rows8, device=dev).view(-1, 1); j = torch.arange(row_bytes, device=dev).view(1, -1)
            mats.append(((p * 131 + sidx * 7 + j) % 251).to(torch.uint8))
        got = shard.bloom_exchange(dist, mats, n_samples, n_parts, world, rank)
        my_s = shard.samples_of_rank(n_samples, world, rank)
        ok = True
        for a_, s_ in enumerate(my_s):
            for p in range(n_parts):
                exp = ((p * 131 + s_ * 7 + torch.arange(row_bytes, device=dev)) % 251).to(torch.uint8)
                ok = ok and bool(torch.equal(got[a_, p], exp))
        rec["bloom_exchange"] = {"ok": ok, "bytes_sent": len(mine) * n_samples * row_bytes, "bytes_received": int(got.numel()), "samples": n_samples, "partitions": n_parts}
    except Exception as e:      # (reported, not fatal: the line's own figure does not depend on it)
        rec["error"] = repr(e)
    out = [None] * world
    if dist is not None and world > 1:
        dist.all_gather_object(out, rec)
    else:
        out = [rec]
    return out


def pmc_traffic(wl, lists, N, P, kernel):
    """HBM bytes per launch of the merge kernel from the committed rocprofv3 --pmc passes
    (FETCH_SIZE + WRITE_SIZE, profiles/merge_pmc.json) -- only when they were taken on this workload."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "merge_pmc.json")))
    except Exception:
        return None
    d = d.get(f"{wl}:{lists if wl in ('count', 'pa63') else 'uniform'}:{N}x{P}", {}).get(kernel, {})
    return d["fetch_bytes"] + d["write_bytes"] if d else None


def cpu_baseline(host_lists, n_parts, N, kw, tasks_d, mode, with_io=False):
    """The oracle (a port of the reference's KmerMerger / HashMerger linear-scan merge, merge.hpp:183-260, 441-517, 575-644) on the
    host cores with the reference's task granularity -- one merge task per partition handed to a pool of threads
    (task_scheduler.hpp:381-417) -- over ALL partitions of the step, on min(partitions, cores) threads.  with_io: a second figure
    with the files around it, as KmerMergeTask::exec has them (task.hpp:690-743): every task reads its N count files
    (counts/partition_<p>/<id>.kmer) and writes its matrix file.  The checker timed as a baseline; never part of the measured GPU path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import shutil, subprocess, tempfile
    from concurrent.futures import ThreadPoolExecutor
    so = os.path.join(ROOT, "oracle", "libkmx_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    import orc
    omode = {0: orc.MODE_COUNT, 1: orc.MODE_PA, 2: orc.MODE_BF, 3: orc.MODE_BFC, 4: orc.MODE_BFT}[mode]
    nproc = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n_thr = max(1, min(nproc, n_parts))

    def merge(j, lists):
        d = tasks_d[j]
        t0 = time.perf_counter()
        body, rows, stats = orc.merge_matrix(lists, kw, d["soft_min"], d["rec_min"], d["share_min"], omode, d.get("lower", 0), d.get("upper", 0))
        return time.perf_counter() - t0, rows, body

    all_lists = [host_lists(j) for j in range(n_parts)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(n_thr) as ex:          # (ctypes releases the GIL: the merges run in parallel)
        res = list(ex.map(lambda j: merge(j, all_lists[j])[:2], range(n_parts)))
    wall = time.perf_counter() - t0
    recs = sum(len(c) for ls in all_lists for _, c in ls)
    out = {"value": recs / wall, "unit": "k-mers/s", "cores": n_thr, "kind": "port",
           "host_cores": nproc, "per_core": recs / sum(r[0] for r in res),
           "sample": f"all {n_parts} partitions of the step ({recs} input records, {sum(r[1] for r in res)} rows out), one merge task per "
                     f"partition on a pool of {n_thr} threads (host has {nproc} cores), oracle linear-scan merge in memory, {wall:.1f} s wall, "
                     f"{sum(r[0] for r in res):.1f} s of CPU"}
    if with_io:
        tmp = tempfile.mkdtemp(prefix="kmx_cpu_")
        try:
            rbw = 2 * kw + 1
            def write_part(j):
                os.makedirs(os.path.join(tmp, f"partition_{j}"))
                for i, (kk, cc) in enumerate(all_lists[j]):
                    rec = np.empty((len(cc), rbw), np.uint32)
                    rec[:, :2 * kw] = np.ascontiguousarray(kk, dtype=np.uint64).reshape(len(cc), kw).view(np.uint32)
                    rec[:, 2 * kw] = cc
                    with open(os.path.join(tmp, f"partition_{j}", f"{i}.kmer"), "wb") as f:
                        f.write(b"\0" * 41); rec.tofile(f)
            with ThreadPoolExecutor(n_thr) as ex:
                list(ex.map(write_part, range(n_parts)))
            del all_lists
            os.sync()
            def task(j):      # KmerMergeTask::exec: count files in, matrix file out
                lists = []
                for i in range(N):
                    rec = np.fromfile(os.path.join(tmp, f"partition_{j}", f"{i}.kmer"), np.uint32, offset=41).reshape(-1, rbw)
                    lists.append((np.ascontiguousarray(rec[:, :2 * kw]).view(np.uint64).reshape(-1, kw), np.ascontiguousarray(rec[:, 2 * kw])))
                _, rows, body = merge(j, lists)
                with open(os.path.join(tmp, f"matrix_{j}.count"), "wb") as f:
                    f.write(b"\0" * 45); f.write(body)
                return rows
            t0 = time.perf_counter()
            with ThreadPoolExecutor(n_thr) as ex:
                rows = list(ex.map(task, range(n_parts)))
            wall_io = time.perf_counter() - t0
            out["with_io"] = {"value": recs / wall_io, "unit": "k-mers/s", "wall_s": wall_io,
                              "sample": f"the same {n_parts} tasks, each reading its {N} count files and writing its matrix file ({sum(rows)} rows) on the box's local disk"}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return out


if __name__ == "__main__":
    main()
