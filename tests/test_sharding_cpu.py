"""World-size-2 gloo test (CPU) of the multi-GPU logic: round-robin partition map and the
job-level reduction bench.py uses (max time over ranks, records summed over ranks)."""
import os, socket, sys
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmtricks_amd import shard


def test_round_robin_is_a_partition_of_the_set():
    for P, G in ((256, 8), (256, 1), (7, 3), (4, 8)):
        seen = []
        for r in range(G):
            mine = shard.partitions_of_rank(P, G, r)
            assert all(shard.rank_of_partition(p, G) == r for p in mine)
            seen += mine
        assert sorted(seen) == list(range(P))
    with pytest.raises(ValueError):
        shard.partitions_of_rank(4, 2, 2)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.partitions_of_rank(9, world, rank)
    # every rank "merges" its partitions: 100 + p records each, taking rank-dependent time
    recs = float(sum(100 + p for p in mine))
    secs = 0.5 + rank
    t, r = shard.reduce_job(dist, torch.device("cpu"), secs, recs)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    q.put((rank, t, r, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_reduction():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    for rank, t, r, gathered in res:
        assert t == 1.5                                  # max over ranks
        assert r == float(sum(100 + p for p in range(9)))  # summed over ranks
        assert sorted(gathered[0] + gathered[1]) == list(range(9))
        assert gathered[0] == [0, 2, 4, 6, 8] and gathered[1] == [1, 3, 5, 7]
