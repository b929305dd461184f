"""GPU parity against the committed oracle goldens of the BASELINE.json configs: EVERY unit (all 1024 config-3 windows,
all 10 000 config-2 pairs, all 1 000 000 config-5 pairs), through the C-ABI. The goldens come from the CPU oracle
(tests/golden/make_config_goldens.py); tests/test_config_goldens.py keeps them honest on CPU."""
import hashlib

import numpy as np
import pytest

import golden_io as G

pytestmark = pytest.mark.gpu


def test_config3_all_1024_windows_equal_the_golden():
    from genomeworks_amd import cudapoa, synthetic
    rows, _ = G.config3_windows()
    s = G.summary()["config3"]
    windows = [[r.decode() for r in synthetic.generate_window(s["first_seed"] + w)] for w in range(s["windows"])]
    b = cudapoa.CudaPoaBatch(s["max_seqs"], s["max_seq"], 8 << 30, output_type="consensus", band_mode="static_band",
                             alignment_band_width=s["band"], max_nodes_per_graph=3 * s["max_seq"])
    for w in windows:
        st, seq_st = b.add_poa_group(w)
        assert st == 0 and all(x == 0 for x in seq_st)
    b.generate_poa()
    cons, cov, status = b.get_consensus()
    assert len(cons) == 1024
    bad = [w for w in range(1024) if (status[w], cons[w], list(cov[w])) != (rows[w]["status"], rows[w]["consensus"], rows[w]["coverage"])]
    assert not bad, "windows that differ from the oracle golden: %s" % bad[:20]
    assert b.total_cells() == s["cells"]
    assert hashlib.sha256("\n".join(cons).encode()).hexdigest() == s["consensus_sha256"]
    # the steady-state loop of the benchmark (generate_poa + get_consensus on the same batch object) reproduces it
    b.generate_poa()
    cons2, cov2, status2 = b.get_consensus()
    assert cons2 == cons and cov2 == cov and status2 == status


def _run_pairs(cfg, n_pairs=None):
    from genomeworks_amd import cudaaligner, synthetic
    n = n_pairs or cfg["pairs"]
    pairs = synthetic.generate_pairs(cfg["seed"], n, cfg["length"], cfg["mut"], cfg["ins"], cfg["dele"])
    al = cudaaligner.CudaAlignerBatch(max_bandwidth=cfg["max_bandwidth"], max_device_memory_allocator_caching_size=24 << 30)
    add = al._L.gw_aligner_add_alignment
    for q, t in pairs:
        st = add(al._h, q, len(q), t, len(t), 0, 0)
        assert st == 0, st
    al.align_all()
    return al, pairs


def test_config2_all_10000_pairs_equal_the_golden():
    g, s = G.config2_pairs(), G.summary()["config2"]
    al, pairs = _run_pairs(s)
    assert al.band_cells() == s["band_cells"]
    dev = al.get_alignments_device()  # before sync_alignments(): the device-resident form of the same results
    r = al.get_runs()
    assert len(r["status"]) == 10000 and (r["status"] == 0).all() and (r["optimal"] == g["optimal"]).all()
    fp = G.run_fingerprints(r["offsets"], r["ops"], r["counts"])
    bad = np.nonzero(fp != g["fingerprint"])[0]
    assert len(bad) == 0, "pairs whose CIGAR differs from the oracle golden: %s" % bad[:20]
    assert (G.edit_distances(r["offsets"], r["ops"], r["counts"]) == g["edit_distance"]).all()
    # get_alignments_device(): alignment i (as added, via metadata) is stored back to front in [offsets[i], offsets[i+1])
    assert dev["n_alignments"] == 10000 and dev["total_length"] == len(r["ops"])
    idx = dev["metadata"] & ((1 << 27) - 1)
    assert sorted(idx.tolist()) == list(range(10000)) and ((dev["metadata"] >> 31) == 1).all()
    for k in (0, 1, 4999, 9999):
        i = int(idx[k])
        lo, hi = int(dev["cigar_offsets"][k]), int(dev["cigar_offsets"][k + 1])
        a, b = int(r["offsets"][i]), int(r["offsets"][i + 1])
        assert (dev["cigar_operations"][lo:hi][::-1] == r["ops"][a:b]).all()
        assert (dev["cigar_runlengths"][lo:hi][::-1] == r["counts"][a:b]).all()
    # the whole device form, vectorised: reversing each segment gives the forward runs of that alignment
    d_off = dev["cigar_offsets"].astype(np.int64)
    order = np.argsort(idx, kind="stable")
    seg_len = np.diff(d_off)[order]
    assert (seg_len == np.diff(r["offsets"])).all()


def test_config5_all_1000000_pairs_equal_the_golden():
    g, s = G.config5_pairs(), G.summary()["config5"]
    al, pairs = _run_pairs(s)
    assert al.band_cells() == s["band_cells"]
    r = al.get_runs()
    assert len(r["status"]) == s["pairs"] and (r["status"] == 0).all()
    assert (r["optimal"] == g["optimal"]).all()
    ed = G.edit_distances(r["offsets"], r["ops"], r["counts"])
    assert (ed == g["edit_distance"]).all()
    fp = G.run_fingerprints(r["offsets"], r["ops"], r["counts"])
    assert hashlib.sha256(fp.tobytes()).hexdigest() == s["fingerprint_sha256"]
    got = G.block_digests(fp, s["block"])
    bad = [k for k, (a, b) in enumerate(zip(got, g["block_sha"])) if a != str(b)]
    assert not bad, "blocks of 1024 pairs that differ from the oracle golden: %s" % bad[:10]


def test_config4_all_598_long_read_windows_equal_the_golden():
    """BASELINE configs[3]: every window of the long-read MSA set (32-bit scores and ids, HBM row tables, bands up to
    1536 columns, one wavefront per 256-column pass) through the size-class plan -- four batches of different shapes
    resident and running at once on host threads -- MSA rows hashed against the oracle's."""
    import importlib.util
    import json
    import os
    from genomeworks_amd import cudapoa
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_long_read_goldens", os.path.join(here, "make_long_read_goldens.py"))
    lr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lr)
    with open(os.path.join(here, "config4_long_reads.json")) as f:
        golden = json.load(f)
    windows, cfgs, groups = lr.plan()
    assert cfgs == golden["batch_configs"] and len(cfgs) >= 3
    plan = lr.size_plan(windows)
    out = cudapoa.process_windows_size_classes(windows, plan, memory_budget=golden["memory_budget_bytes"], output_type="msa",
                                               digest=lr.msa_digest)
    assert len(out["status"]) == golden["windows"] == 598
    bad = []
    for d in golden["windows_detail"]:
        w = d["w"]
        assert out["worker"][w] == d["cfg"]
        if out["status"][w] != d["status"] or (d["status"] == 0 and out["msa"][w] != d["msa_sha"]):
            bad.append(w)
    assert not bad, "windows whose MSA differs from the oracle golden: %s" % bad[:20]
    assert out["launches"] == len(cfgs) and 0 < out["compute_seconds"] <= out["seconds"]


def test_long_read_windows_through_the_sequential_multi_batch_loop():
    """The reference's own flow (get_multi_batch_sizes-style single plan, fills one after the other,
    cudapoa/src/main.cpp:197-326) on a part of the same set: same MSAs as the size-class run wherever both succeed."""
    import importlib.util
    import os
    from genomeworks_amd import cudapoa, multibatch, synthetic
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_long_read_goldens", os.path.join(here, "make_long_read_goldens.py"))
    lr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lr)
    import json
    with open(os.path.join(here, "config4_long_reads.json")) as f:
        golden = {d["w"]: d for d in json.load(f)["windows_detail"]}
    ids = list(range(0, 598, 9))
    windows = [synthetic.long_read_window(w, 32768) for w in ids]
    cfgs, groups = cudapoa.plan_multi_batch_sizes(windows, int(20e9), msa_flag=True, band_width=256, band_mode="adaptive_band",
                                                  adaptive_storage_factor=4.0)
    out = multibatch.run_plan(windows, cfgs, groups, int(20e9), output_type="msa", band_mode="adaptive_band", digest=lr.msa_digest)
    assert len(out["results"]) == len(ids) and out["launches"] >= 2
    cfg_of = {k: c for c, g in zip(cfgs, groups) for k in g}
    both = 0
    for k, w in enumerate(ids):
        got, st = out["results"][k]
        # the reference's binning gives a merged bin the read capacity of its first bin: a deeper window loses its last
        # reads there (exceeded_maximum_sequences_per_poa) and its MSA is that of the reads the batch accepted
        if len(out["accepted"][k]) < len(windows[k]):
            assert len(out["accepted"][k]) == cfg_of[k]["max_sequences_per_poa"]
            continue
        if st == 0 and golden[w]["status"] == 0:
            assert got == golden[w]["msa_sha"], w
            both += 1
    assert both >= len(ids) // 2


def test_long_read_prefixes_equal_the_reference_itself():
    """tests/golden/reference_simt_long_prefixes.json: windows of the two LARGEST size classes of configs[3] (reads of 7.6-30 kbp:
    32-bit scores and ids, HBM row tables, the adaptive band growing from 256 towards its 1536-column cap) cut to their first 4, 8
    or 12 reads and answered by the REFERENCE's own cudapoa library on the SIMT emulator
    (tests/golden/make_reference_simt_long_prefixes.py). None of them was among the 245 whole windows that
    reference_simt_config_check.json covered when they were cut (424 since). The HIP path, in a batch of the class's BatchConfig, gives the same statuses and MSA rows."""
    import importlib.util
    import json
    import os
    from genomeworks_amd import cudapoa, synthetic
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_long_read_goldens", os.path.join(here, "make_long_read_goldens.py"))
    lr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lr)
    with open(os.path.join(here, "reference_simt_long_prefixes.json")) as f:
        fixture = json.load(f)
    rows = fixture["windows"]
    assert len(rows) >= 60 and {r["cfg"] for r in rows} == {0, 1} and all(r["oracle_equal"] for r in rows)
    assert max(r["reads"] for r in rows) >= 8  # (sets in which the band has grown)
    bad = []
    for k, c in enumerate(fixture["batch_configs"]):
        mine = [r for r in rows if r["cfg"] == k]
        b = cudapoa.CudaPoaBatch.from_batch_config(c["max_sequence_size"], c["max_sequences_per_poa"], c["alignment_band_width"], "adaptive_band",
                                                   96 << 30, output_type="msa", adaptive_storage_factor=fixture["storage_factor"])
        got = b.batch_size
        assert (got.max_nodes_per_graph, got.matrix_sequence_dimension, got.max_consensus_size) == (
            c["max_nodes_per_graph"], c["matrix_sequence_dimension"], c["max_consensus_size"])
        for r in mine:
            st, _ = b.add_poa_group(synthetic.long_read_window(r["w"], 32768)[:r["reads"]])
            assert st == r["add_status"] == 0, (r["w"], st)
        b.generate_poa()
        n = b.get_msa_native()
        assert n == len(mine)
        for i, r in enumerate(mine):
            msa, status = b.collect_msa_one(i)
            if status != r["status"] or (status == 0 and (lr.msa_digest(msa) != r["msa_sha"] or len(msa) != r["msa_rows"])):
                bad.append((r["w"], r["reads"], status))
        del b
    assert not bad, "prefix windows where the HIP path differs from the reference: %s" % bad[:10]


def test_more_windows_than_simds_run_as_a_persistent_grid_and_equal_the_golden(monkeypatch):
    """A batch of more windows than the device holds at one wavefront per SIMD (gwhip_poa_args::work_counters): blocks take
    window after window from a device counter. 2300 windows (the 1024 metric windows cyclically) in the metric configuration and
    in the benchmarks' full band, twice each on the same Batch (the last block of a launch has to leave the counters at zero),
    against the committed goldens; then once more with one block per window (GWHIP_POA_PERSISTENT=0)."""
    import golden_io as G
    from genomeworks_amd import cudapoa, synthetic
    rows, _ = G.config3_windows()
    fgold = G.full_band_goldens()["fingerprint"]
    n = 2300
    windows = [[r.decode() for r in synthetic.generate_window(1000 + (w % 1024))] for w in range(n)]

    def run(make, check):
        b = make()
        for w in windows:
            st, _ = b.add_poa_group(w)
            assert st == 0, st
        for _ in range(2):
            b.generate_poa()
            cons, cov, status = b.get_consensus()
            assert len(cons) == n
            check(cons, cov, status)

    def check_static(cons, cov, status):
        bad = [w for w in range(n) if status[w] != rows[w % 1024]["status"] or cons[w] != rows[w % 1024]["consensus"]
               or list(cov[w]) != list(rows[w % 1024]["coverage"])]
        assert not bad, bad[:10]

    def check_full(cons, cov, status):
        fp = G.band_mode_fingerprints(cons, cov, status)
        assert all((fp[k:k + 1024] == fgold[:len(fp[k:k + 1024])]).all() for k in range(0, n, 1024))

    static = lambda: cudapoa.CudaPoaBatch(32, 1024, 24 << 30, output_type="consensus", band_mode="static_band", alignment_band_width=256,
                                          max_nodes_per_graph=3072)
    full = lambda: cudapoa.CudaPoaBatch(200, 1024, 48 << 30, output_type="consensus", band_mode="full_band", max_nodes_per_graph=3072,
                                        matrix_sequence_dimension=1024)
    run(static, check_static)
    run(full, check_full)
    monkeypatch.setenv("GWHIP_POA_PERSISTENT", "0")
    run(static, check_static)
