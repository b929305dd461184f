"""The N>1 path: index split + gather by global index, exercised with 2 gloo processes on CPU."""
import os
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from genomeworks_amd.multi_gpu import shard_range
    for n in (0, 1, 7, 1024, 1000003):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from genomeworks_amd.multi_gpu import run_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    units = ["w%04d" % i for i in range(37)]
    seen = []

    def process(chunk, lo):
        seen.append((lo, len(chunk)))
        return [u[::-1] + ":%d" % (lo + k) for k, u in enumerate(chunk)]

    res = run_sharded(units, process)
    dist.barrier()
    q.put((rank, seen, res))
    dist.destroy_process_group()


def test_two_rank_gloo_gather_is_order_independent():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort()
    (r0, seen0, res0), (r1, seen1, res1) = got
    assert seen0 == [(0, 19)] and seen1 == [(19, 18)]
    assert res1 is None
    assert res0 == [("w%04d" % i)[::-1] + ":%d" % i for i in range(37)]  # same as a single-rank run
