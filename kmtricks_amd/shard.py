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
