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


def test_balanced_partition_is_exact_deterministic_and_balanced():
    import random
    from genomeworks_amd.multi_gpu import balanced_partition, poa_window_cost, pair_cost
    rng = random.Random(3)
    costs = [rng.choice([1, 5, 40, 300]) * rng.randrange(1, 50) for _ in range(997)]
    for world in (1, 2, 3, 8):
        parts = balanced_partition(costs, world)
        assert sorted(i for p in parts for i in p) == list(range(len(costs)))  # every unit exactly once
        assert all(p == sorted(p) for p in parts)
        assert parts == balanced_partition(list(costs), world)                # same on every rank
        loads = [sum(costs[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(costs)                          # LPT bound
    assert balanced_partition([], 4) == [[], [], [], []]
    assert poa_window_cost(["ACGT" * 10] * 3) == 2 * 40 * 256 and poa_window_cost(["ACGT"]) == 0
    assert poa_window_cost(["ACGT" * 10, "ACG"], band_width=0) == 40 * 3
    assert pair_cost("ACGT", "AC") == 6
    with pytest.raises(ValueError):
        balanced_partition([1], 0)


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
    # cost-balanced split of the same units: results still land by global index
    costs = [(i * 7919) % 13 + 1 for i in range(len(units))]
    res_b = run_sharded(units, lambda chunk, idx: [u[::-1] + ":%d" % i for u, i in zip(chunk, idx)], costs=costs)
    dist.barrier()
    q.put((rank, seen, (res, res_b)))
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
    assert res1 == (None, None)
    want = [("w%04d" % i)[::-1] + ":%d" % i for i in range(37)]  # same as a single-rank run
    assert res0[0] == want and res0[1] == want
