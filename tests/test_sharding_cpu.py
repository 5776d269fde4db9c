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


def _bloom_worker(rank, world, port, q, N, P, RB):
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # matrix of partition p: row s, byte b = a value every rank can recompute
    def mat(p):
        s = torch.arange((N + 7) // 8 * 8, dtype=torch.int64).view(-1, 1)
        b = torch.arange(RB, dtype=torch.int64).view(1, -1)
        return ((s * 131 + p * 17 + b * 7) % 251).to(torch.uint8)
    mats = [mat(p) for p in shard.partitions_of_rank(P, world, rank)]
    out = shard.bloom_exchange(dist, mats, N, P, world, rank)
    mine = shard.samples_of_rank(N, world, rank)
    ok = out.shape == (len(mine), P, RB)
    for j, s in enumerate(mine):
        for p in range(P):
            ok = ok and bool(torch.equal(out[j, p], mat(p)[s]))
    q.put((rank, ok, mine))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("N,P,RB", [(5, 4, 24), (11, 7, 8), (2, 2, 16)])
def test_two_rank_gloo_bloom_exchange(N, P, RB):
    """the per-sample Bloom-row all-to-all of hash:bft:bin (howde_utils.hpp:133-187 layout): after the exchange rank
    s mod 2 holds, for each of its samples, the rows of ALL partitions in partition order"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bloom_worker, args=(r, 2, port, q, N, P, RB)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    assert all(ok for _, ok, _ in res)
    assert sorted(x for _, _, mine in res for x in mine) == list(range(N))


def test_bloom_exchange_single_rank():
    mats = [torch.full((8, 4), p, dtype=torch.uint8) for p in range(3)]
    out = shard.bloom_exchange(None, mats, 5, 3, 1, 0)
    assert out.shape == (5, 3, 4) and all(int(out[:, p].max()) == p and int(out[:, p].min()) == p for p in range(3))


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
