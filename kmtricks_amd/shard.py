"""Partition sharding of the count/merge stages over the GPUs of one node.

Partitions are independent units (disjoint minimizer sets / disjoint hash windows, reference
include/kmtricks/gatb/sorting_count.hpp:356 and task_scheduler.hpp:381-417: one merge task per
partition), so partition p simply belongs to rank p mod G and no data-path collective exists.
The helpers below are the only cross-rank logic: the round-robin map and the reduction of the
per-rank timing/volume into the job figure (max time, summed records)."""


def partitions_of_rank(n_partitions: int, world: int, rank: int):
    """Round-robin: partition p -> rank p mod world."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    return list(range(rank, n_partitions, world))


def rank_of_partition(p: int, world: int) -> int:
    return p % world


def reduce_job(dist, device, seconds: float, records: float):
    """-> (max seconds over ranks, records summed over ranks); identity when dist is None."""
    if dist is None:
        return seconds, records
    import torch
    t = torch.tensor([seconds], device=device, dtype=torch.float64)
    r = torch.tensor([records], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(r, op=dist.ReduceOp.SUM)
    return float(t.item()), float(r.item())


# ---- the one exchange on the path: per-sample Bloom filters (hash:bft:bin) ---------------------------------------
# After the merge + transpose (KMX_MODE_BFT) rank g holds, for each of its partitions p (p mod G == g), a matrix of
# round_up8(N) rows x W/8 bytes whose row s is sample s's slice of its final filter; the filter of sample s is the
# concatenation of its rows over p = 0..P-1 (reference include/kmtricks/howde_utils.hpp:133-187, which gathers them
# through the matrix files on disk).  With the matrices resident in HBM the gather is one all-to-all over RCCL:
# sample s belongs to rank s mod G, message src -> dst = (partitions of src) x (samples of dst) x W/8 bytes
# (SURVEY.md section 8e).

def samples_of_rank(n_samples: int, world: int, rank: int):
    """Round-robin: sample s -> rank s mod world."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    return list(range(rank, n_samples, world))


def bloom_exchange(dist, mats, n_samples: int, n_partitions: int, world: int, rank: int, group=None):
    """mats: this rank's transposed partition matrices (torch uint8 [rows >= n_samples, W/8], one per partition
    rank, rank + world, ... in ascending order, all on one device).  Returns a uint8 tensor
    [len(samples_of_rank), n_partitions, W/8]: entry [j, p] is the slice of partition p in the filter of this
    rank's j-th sample (flatten the last two axes = the filter's bit vector, partitions in order).
    dist: torch.distributed (backend nccl = RCCL on the GPUs, gloo on CPU), or None / world 1 for the local case."""
    import torch
    mine_p = partitions_of_rank(n_partitions, world, rank)
    if len(mats) != len(mine_p):
        raise ValueError(f"rank {rank} owns {len(mine_p)} partitions, got {len(mats)} matrices")
    if not mats:
        raise ValueError("a rank without partitions cannot take part (use world <= n_partitions)")
    row_bytes = mats[0].shape[1]
    dev = mats[0].device
    my_s = samples_of_rank(n_samples, world, rank)
    out = torch.empty((len(my_s), n_partitions, row_bytes), dtype=torch.uint8, device=dev)
    if dist is None or world == 1:
        for j, p in enumerate(mine_p):
            out[:, p, :] = mats[j][:n_samples]
        return out
    # send buffer: for dst in ranks, for my partitions (ascending), the rows of dst's samples (ascending)
    chunks, in_split = [], []
    for dst in range(world):
        ns = len(samples_of_rank(n_samples, world, dst))
        for m in mats:
            chunks.append(m[dst:n_samples:world])
        in_split.append(len(mats) * ns)
    send = torch.cat(chunks, dim=0).contiguous()
    out_split = [len(partitions_of_rank(n_partitions, world, src)) * len(my_s) for src in range(world)]
    recv = torch.empty((sum(out_split), row_bytes), dtype=torch.uint8, device=dev)
    dist.all_to_all_single(recv, send, output_split_sizes=out_split, input_split_sizes=in_split, group=group)
    off = 0
    for src in range(world):
        ps = partitions_of_rank(n_partitions, world, src)
        blk = recv[off:off + out_split[src]].view(len(ps), len(my_s), row_bytes)
        out[:, ps, :] = blk.permute(1, 0, 2)
        off += out_split[src]
    return out
