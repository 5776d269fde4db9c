"""CPU-only: bench.py's rank plumbing under a two-rank gloo group with a stand-in for the ctypes binding of libkmx -- argument
parsing, the partition -> rank map, the two-batches-in-flight loop, the rank-0 reduction (max time, summed records) and, for
`--workload bft`, the call order of the one exchange (body_to_device of every partition, then the all-to-all of
kmtricks_amd/shard.py).  Nothing here computes a merge: the first run on an 8-GPU node must not fail for a reason this finds."""
import ctypes, json, os, socket, sys
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _StubResult:
    def __init__(self, owner, n_tasks):
        self.o, self.n = owner, n_tasks
    def wait(self): self.o.log.append("wait")
    def kernel_ms(self): return 1.0
    def transpose_ms(self): return -1.0
    def algo_bytes(self, t): return 1000
    def rows(self, t): return 8
    def kernel(self): return "k_stub"
    def free(self): self.o.log.append("free")
    def body_to_device(self, t, ptr, n):
        ctypes.memset(ptr, (self.o.rank * 16 + t) & 0xFF, n)
        self.o.log.append(f"body{t}")


class _StubContext:
    def __init__(self, device): self.device, self.log, self.rank = device, [], int(os.environ.get("KMX_STUB_RANK", "0"))
    def set_profiling(self, on): pass
    def prepare(self, tasks): return (tasks, len(tasks))
    def merge_dev(self, prep):
        self.log.append("submit")
        return _StubResult(self, prep[1])
    def close(self): _StubLib.last_log = list(self.log)


class _StubLib:
    MODE_COUNT, MODE_PA, MODE_BF, MODE_BFC, MODE_BFT = 0, 1, 2, 3, 4
    Context = _StubContext
    last_log = []


def _worker(rank, world, port, q, argv, wl):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port); os.environ["KMX_STUB_RANK"] = str(rank)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from kmtricks_amd import shard
    a = bench.parse_args(argv)
    env = dict(torch=torch, dist=dist, lib=_StubLib, shard=shard, rank=rank, world=world, local=0, dev=torch.device("cpu"), run_pipeline=False,
               sync=lambda: None, empty_cache=lambda: None)
    out = bench.run_workloads(a, wl, env)
    q.put((rank, out, _StubLib.last_log))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("wl,extra", [("count", ["--lists", "random"]), ("bft", [])])
def test_bench_rank_plumbing_two_ranks(wl, extra):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    argv = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--samples", "6", "--partitions-per-gpu", "2", "--total-partitions", "4",
            "--genome", "4000", "--bloom", "4096", "--no-cpu-baseline"] + extra
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, argv, wl)) for r in range(2)]
    for p in procs: p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda x: x[0])
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    (r0, out0, log0), (r1, out1, log1) = res
    assert out1 is None and out0 is not None
    json.dumps(out0)                                       # the line must serialise
    assert out0["n_gpus"] == 2 and out0["steps"] == 3 and out0["warmup"] == 1 and out0["scaling"] == "weak"
    assert out0["value"] > 0 and out0["ms_per_step"] > 0 and "cpu_baseline" not in out0
    assert out0["roofline"]["kernel"] == "k_stub" and out0["roofline"]["algo_bytes_per_launch"] == 2000
    assert "2 GPU(s)" in out0["config"]["parallelism"]
    for log in (log0, log1):
        # 3 set-up batches one at a time, then warm-up and timed steps with two batches in flight: submit, submit, wait ...
        assert log.count("submit") == 3 + 1 + 3 and log.count("wait") == log.count("submit") == log.count("free")
        timed = log[-(3 * (3 + (2 if wl == "bft" else 0))):] if wl == "bft" else log[-9:]
        assert timed[0] == "submit" and timed[1] == "submit"      # the second batch is submitted before the first is waited for
        if wl == "bft":                                           # every partition's body goes to the exchange buffer before the batch is freed
            i = log.index("wait")
            assert log[i + 1:i + 3] == ["body0", "body1"] and log[i + 3] == "free"
    if wl == "bft":
        assert "all-to-all" in out0["config"]["parallelism"]
    # round 6: the line names what every rank bound and saw, and the path's one collective has crossed the group once with checked content
    pr = out0["ranks"]["per_rank"]
    assert out0["ranks"]["world_size"] == 2 and [x["rank"] for x in pr] == [0, 1] and all(x["world_size_seen"] == 2 for x in pr)
    assert all("error" not in x and x["bloom_exchange"]["ok"] and x["bloom_exchange"]["bytes_sent"] > 0 for x in pr)
