"""bench.py's banded-aligner record (bench_aligner) end to end on the CPU: the aligner is a double that answers from the oracle,
so the record's flow -- fill, timed loops, the golden verdict outside the clock, the reductions -- runs without a GPU. The double
is test infrastructure only; the real record is produced by genomeworks_amd.cudaaligner.CudaAlignerBatch on the device."""
import importlib.util
import os

import numpy as np

import oracle_aligner as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_for_aligner_record", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


class FakeLib:
    @staticmethod
    def gw_aligner_add_alignment(handle, q, lq, t, lt, rq, rt):
        assert lq == len(q) and lt == len(t)
        handle.pairs.append((q, t))
        return 0


class FakeAligner:
    """The calls bench_aligner makes on CudaAlignerBatch, answered by the oracle."""
    spoil = None  # (pair, field) to falsify in get_runs(); "raise" to fail there

    def __init__(self, max_bandwidth, **kwargs):
        self.bw, self._L, self._h, self.pairs, self.res = max_bandwidth, FakeLib, self, [], None

    def align_all(self):
        self.res = [A.align(q, t, self.bw) for q, t in self.pairs]

    def band_cells(self):
        return sum(r["cells"] for r in self.res)

    def relaunch_timed(self):
        return 1.0

    def reset(self):
        self.pairs, self.res = [], None

    def device_sync(self):
        return len(self.res), sum(len(r["runs"]) for r in self.res)

    def sync(self):
        return len(self.res)

    def get_runs(self):
        if FakeAligner.spoil == "raise":
            raise RuntimeError("no runs today")
        offs, ops, cnts = [0], [], []
        for r in self.res:
            for o, k in r["runs"]:
                ops.append(o)
                cnts.append(k)
            offs.append(len(ops))
        out = dict(offsets=np.array(offs, np.int64), ops=np.array(ops, np.int8), counts=np.array(cnts, np.int32),
                   status=np.array([r["status"] for r in self.res], np.int32), optimal=np.array([1 if r["optimal"] else 0 for r in self.res], np.int32))
        if FakeAligner.spoil is not None:
            pair, field = FakeAligner.spoil
            at = pair if field in ("status", "optimal") else int(out["offsets"][pair])
            out[field][at] = out[field][at] + 1 if field == "counts" else out[field][at] ^ 1
        return out


def run_record(bench, monkeypatch, cfg, pairs):
    from genomeworks_amd import cudaaligner
    monkeypatch.setattr(cudaaligner, "CudaAlignerBatch", FakeAligner)
    monkeypatch.setitem(cfg, "pairs", pairs)
    return bench.bench_aligner("record under test", cfg, 0, 1, 0, lambda: None, None, None, 2, 0.0)


def test_aligner_record_reports_the_golden_verdict(monkeypatch):
    bench = load_bench()
    for cfg, pairs in ((bench.CONFIG2, 24), (bench.CONFIG5, 2048)):
        FakeAligner.spoil = None
        rec = run_record(bench, monkeypatch, cfg, pairs)
        assert rec["equals_oracle_golden"] is True, rec["golden_compared"]
        assert rec["pairs"] == pairs and rec["value"] > 0 and rec["kernel_only"]["ms"] == 1.0
        assert str(pairs) in rec["golden_compared"]
        for field in ("counts", "ops", "optimal", "status"):
            FakeAligner.spoil = (pairs // 2, field)
            assert run_record(bench, monkeypatch, cfg, pairs)["equals_oracle_golden"] is False, field
        FakeAligner.spoil = "raise"
        rec = run_record(bench, monkeypatch, cfg, pairs)
        assert rec["equals_oracle_golden"] is None and "no runs today" in rec["golden_compared"]
    FakeAligner.spoil = None


def _rank(rank, world, port, q):
    import contextlib
    import io
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      GW_BENCH_RANKS_PER_DEVICE=str(world))
    import torch
    import golden_io
    import test_bench_eight_ranks as E
    from genomeworks_amd import cuda, cudaaligner, cudapoa, synthetic
    rows, _ = golden_io.config3_windows()
    E.FakeBatch.rows = rows
    E.FakeBatch.by_first_read = {synthetic.generate_window(1000 + w)[0].decode(): w for w in range(len(rows))}
    cudapoa.CudaPoaBatch = E.FakeBatch
    cudaaligner.CudaAlignerBatch = FakeAligner
    cuda.cuda_set_device = lambda d: None
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda d: None
    torch.cuda.synchronize = lambda *a: None
    import bench
    bench.CONFIG2["pairs"], bench.CONFIG5["pairs"] = 25, 3001  # odd counts: the ranks' ranges are uneven and not block aligned
    if rank == 1 and os.environ.get("GW_TEST_SPOIL_RANK1"):
        FakeAligner.spoil = (3, "ops")  # (an operation: seen by the edit distances, which the golden holds for every pair)
    import tempfile
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--sub-configs", "aligner",
                "--record-dir", tempfile.mkdtemp(prefix="gw_bench_")]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    q.put((rank, buf.getvalue()))


def _two_ranks(spoil):
    import json
    import torch.multiprocessing as mp
    world = 2
    if spoil:
        os.environ["GW_TEST_SPOIL_RANK1"] = "1"
    else:
        os.environ.pop("GW_TEST_SPOIL_RANK1", None)
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 29950 + (os.getpid() % 40) + (7 if spoil else 0)
        procs = [ctx.Process(target=_rank, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = dict(q.get(timeout=600) for _ in range(world))
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        os.environ.pop("GW_TEST_SPOIL_RANK1", None)
    assert got[1].strip() == ""
    lines = [l for l in got[0].splitlines() if l.startswith("{")]
    # every sub-record on a line of its own, the small headline line last (bench.emit)
    assert len(lines[-1].encode()) < 4096 and "metric" in json.loads(lines[-1])
    return {json.loads(l)["sub_record"]: json.loads(l)["record"] for l in lines[:-1]}


def test_aligner_records_with_two_gloo_ranks_reduce_the_verdict():
    """bench.py --gpus 2 --sub-configs aligner over gloo with the doubles: every rank compares its own range of pairs with the
    golden, the record says True only when all do (one falsified run on rank 1 turns it False)."""
    sub = _two_ranks(False)
    assert sub["configs[1]"]["equals_oracle_golden"] is True and sub["configs[4]"]["equals_oracle_golden"] is True
    assert sub["configs[1]"]["pairs"] == 25 and sub["configs[4]"]["pairs_per_gpu"] == 2048  # (rank 0: its range ends on the golden's block grid)
    sub = _two_ranks(True)
    assert sub["configs[1]"]["equals_oracle_golden"] is False and sub["configs[4]"]["equals_oracle_golden"] is False
