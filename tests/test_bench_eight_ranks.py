"""bench.py's multi-rank logic with world = 8 (gloo, CPU): the device layer is replaced by a double that answers from the
committed oracle golden, so what runs here is exactly what has never met eight GPUs -- rendezvous, the weak headline with
every rank checking its own output, both strong-scaling records (cost-balanced split, gather by global index over the gloo
host group, per-block golden digests), the scalar reductions and the single JSON line of rank 0."""
import json
import os
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeBatch:
    """Stands in for genomeworks_amd.cudapoa.CudaPoaBatch: consensus of a window = the golden row of the window whose first read it is."""
    by_first_read = None

    def __init__(self, *a, **k):
        self.idx = []

    def add_poa_group(self, reads):
        self.idx.append(FakeBatch.by_first_read[reads[0]])
        return 0, [0] * len(reads)

    def generate_poa(self):
        pass

    relaunch = generate_poa

    def relaunch_timed(self):
        return 1.0, 0.1

    def get_consensus_native(self, in_place=False):
        return len(self.idx)

    def total_cells(self):
        return sum(FakeBatch.rows[i]["cells"] for i in self.idx)

    def get_consensus(self):
        rows = [FakeBatch.rows[i] for i in self.idx]
        return [r["consensus"] for r in rows], [r["coverage"] for r in rows], [r["status"] for r in rows]


def _rank(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      GW_BENCH_RANKS_PER_DEVICE=str(world))  # all ranks on "device 0": bench.py then rendezvous over gloo
    import io
    import contextlib
    import torch
    import golden_io
    from genomeworks_amd import cudapoa, cuda, synthetic
    rows, _ = golden_io.config3_windows()
    FakeBatch.rows = rows
    FakeBatch.by_first_read = {synthetic.generate_window(1000 + w)[0].decode(): w for w in range(len(rows))}
    cudapoa.CudaPoaBatch = FakeBatch
    cuda.cuda_set_device = lambda d: None
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda d: None
    torch.cuda.synchronize = lambda *a: None
    import bench
    import tempfile
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--sub-configs", "none",
                "--record-dir", tempfile.mkdtemp(prefix="gw_bench_")]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    q.put((rank, buf.getvalue()))


def test_bench_line_with_eight_gloo_ranks_and_a_device_double():
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 200)
    procs = [ctx.Process(target=_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(got[r].strip() == "" for r in range(1, world))      # one line, from rank 0
    lines = [l for l in got[0].splitlines() if l.startswith("{")]
    # the sub-records first (one line each), the headline LAST and small enough for the driver's tail
    assert len(lines) == 3 and got[0].rstrip().splitlines()[-1] == lines[-1]
    assert len(lines[-1].encode()) < 4096
    line = json.loads(lines[-1])
    assert line["n_gpus"] == world and line["scaling"] == "weak"
    assert line["equals_oracle_golden"] is True                      # every rank hashed its own last step
    assert line["config"]["cells_per_gpu"] == 10990578176
    assert line["roofline"]["frac"] > 0 and line["roofline"]["bound"] == "hbm"
    assert line["sub_records"]["summary"]["strong_scaling_8x"]["golden"] == [1, 1]
    subs = {json.loads(l)["sub_record"]: json.loads(l)["record"] for l in lines[:-1]}
    s1, s8 = subs["strong_scaling"], subs["strong_scaling_8x"]
    full = json.load(open(os.path.join(ROOT, line["sub_records"]["file"])))
    assert full["sub_records"]["strong_scaling_8x"] == s8 and full["value"] == line["value"]
    assert s1["windows"] == 1024 and s1["equals_oracle_golden"] is True
    assert s8["windows"] == 8192 and s8["equals_oracle_golden"] is True
