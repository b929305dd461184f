"""cudapoa::process_windows_multi_device on one GPU: several logical device shards x several batches per shard (host
threads, streams and Batch objects sharing one device and, per shard, one allocator -- the pattern of the reference's
multi-batch benchmark, cudapoa/benchmarks/multi_batch.hpp:55-60,165-176, and of its per-device workers,
cudamapper/src/main.cu:577-592). Results are placed by global window index and must equal the oracle goldens whatever
the worker layout."""
import pytest

import golden_io as G
import oracle_poa as O

pytestmark = pytest.mark.gpu


def config3(n):
    from genomeworks_amd import synthetic
    return [[r.decode() for r in synthetic.generate_window(1000 + w)] for w in range(n)]


@pytest.mark.parametrize("devices, batches", [((0,), 1), ((0, 0), 2), ((0, 0, 0), 1)])
def test_consensus_equals_golden_for_any_worker_layout(devices, batches):
    from genomeworks_amd import cudapoa
    rows, _ = G.config3_windows()
    n = 320
    windows = config3(n)
    # 0.6 GB per shard: about 150 windows of this shape per shard, split over its batches -> no layout holds all 320 at once
    out = cudapoa.process_windows_multi_device(windows, 32, 1024, devices=devices, batches_per_device=batches,
                                               memory_per_device=int(0.6e9), band_mode="static_band", max_nodes_per_graph=3072)
    assert out["status"] == [rows[w]["status"] for w in range(n)]
    bad = [w for w in range(n) if out["consensus"][w] != rows[w]["consensus"] or out["coverage"][w] != rows[w]["coverage"]]
    assert not bad, bad[:10]
    workers = len(devices) * batches
    assert set(out["worker"]) <= set(range(workers)) and min(out["worker"]) >= 0
    assert out["launches"] >= max(workers, 3) and out["seconds"] > 0
    if workers > 1:
        assert len(set(out["worker"])) > 1  # the work really was spread over the workers


def test_msa_mode_and_error_paths():
    from genomeworks_amd import cudapoa, synthetic
    windows = [[r.decode() for r in synthetic.generate_window(7000 + w, 120, 12, 8, 4, 4)] for w in range(9)]
    windows.append(["ACGT" * 100])  # longer than max_sequence_size: refused at add time, reported by status
    out = cudapoa.process_windows_multi_device(windows, 16, 256, devices=(0, 0), batches_per_device=1, memory_per_device=1 << 30,
                                               output_type="msa", band_mode="full_band")
    cfg = O.make_cfg(256, 16, 256, 0, output_mask=2)
    cfg.max_nodes_per_graph = 768
    cfg.matrix_sequence_dimension = 256
    O.lib().poa_cfg_select_types(cfg)
    with O.Workspace(cfg) as ws:
        for i, w in enumerate(windows[:9]):
            ref = ws.process(w)
            assert out["status"][i] == ref["status"] == 0 and out["msa"][i] == ref["msa"]
    assert out["status"][9] == cudapoa.empty_poa_group and out["msa"][9] == []
    with pytest.raises(RuntimeError, match="device id out of range"):
        cudapoa.process_windows_multi_device(windows[:1], 16, 256, devices=(99,))
    with pytest.raises(RuntimeError, match="batches_per_device"):
        cudapoa.process_windows_multi_device(windows[:1], 16, 256, devices=(0,), batches_per_device=0)
    empty = cudapoa.process_windows_multi_device([], 16, 256, devices=(0,))
    assert empty["status"] == [] and empty["launches"] == 0
