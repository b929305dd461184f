"""GPU parity tests: HIP path (through the C-ABI) vs the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

import oracle_poa as O

pytestmark = pytest.mark.gpu

BAND = {"full_band": 0, "static_band": 1, "adaptive_band": 2, "static_band_traceback": 3, "adaptive_band_traceback": 4}


def run_gpu(windows, band_mode, max_seq=1024, max_seqs=32, band_width=256, output_type="consensus", nodes=None, mem=8 << 30, **kw):
    from genomeworks_amd import cudapoa
    b = cudapoa.CudaPoaBatch(max_seqs, max_seq, mem, output_type=output_type, band_mode=band_mode,
                             alignment_band_width=band_width, max_nodes_per_graph=nodes or 3 * max_seq, **kw)
    for w in windows:
        st, seq_st = b.add_poa_group(w)
        assert st == 0 and all(s == 0 for s in seq_st)
    b.generate_poa()
    return b


def oracle_cfg(band_mode, max_seq=1024, max_seqs=32, band_width=256, output_mask=1, nodes=None):
    # mirror of the CudaPoaBatch construction above (explicit BatchConfig ctor)
    cfg = O.make_cfg(max_seq, max_seqs, band_width, BAND[band_mode], output_mask=output_mask)
    cfg.max_nodes_per_graph = nodes or 3 * max_seq
    if band_mode == "full_band":
        cfg.matrix_sequence_dimension = max_seq
    elif band_mode.startswith("static"):
        cfg.matrix_sequence_dimension = band_width + 8
    else:
        cfg.matrix_sequence_dimension = 2 * (band_width + 8)
    cfg.max_banded_pred_distance = 2 * band_width
    O.lib().poa_cfg_select_types(cfg)
    return cfg


_CONFIG3_CACHE = {}


def config3(n, first=1000):
    from genomeworks_amd import synthetic
    for w in range(n):
        if first + w not in _CONFIG3_CACHE:
            _CONFIG3_CACHE[first + w] = [r.decode() for r in synthetic.generate_window(first + w)]
    return [list(_CONFIG3_CACHE[first + w]) for w in range(n)]


@pytest.mark.parametrize("band_mode", list(BAND))
def test_consensus_bit_exact_vs_oracle(band_mode):
    windows = config3(6)
    b = run_gpu(windows, band_mode)
    cons, cov, status = b.get_consensus()
    cells_gpu = b.total_cells()
    cells_ref = 0
    with O.Workspace(oracle_cfg(band_mode)) as ws:
        for i, w in enumerate(windows):
            ref = ws.process(w)
            cells_ref += ref["cells"]
            assert status[i] == ref["status"], (i, status[i], ref["status"])
            if ref["status"] == 0:
                assert cons[i] == ref["consensus"], "window %d consensus differs" % i
                assert cov[i] == list(ref["coverage"]), "window %d coverage differs" % i
        assert ws.overflow_events() == 0  # precondition of the prefix-max scan
    assert cells_gpu == cells_ref


def test_graphs_match_oracle_node_and_edge_counts():
    windows = config3(3)
    b = run_gpu(windows, "static_band")
    b.get_consensus()
    graphs, status = b.get_graphs()
    with O.Workspace(oracle_cfg("static_band")) as ws:
        for i, w in enumerate(windows):
            ref = ws.process(w)
            assert graphs[i].number_of_nodes() == ref["node_count"]


def test_binding_graph_shape():
    # pygenomeworks/test/test_cudapoa_bindings.py:102-123
    b = run_gpu([["ACTGACTG", "ACTTACTG", "ACTCACTG"]], "full_band", max_seq=1024, max_seqs=10)
    graphs, status = b.get_graphs()
    assert graphs[0].number_of_nodes() == 10 and graphs[0].number_of_edges() == 11


def test_three_identical_reads():
    # cudapoa/tests/Test_CudapoaBatch.cu:155-205
    read = "A" * 1023
    for mode in BAND:
        b = run_gpu([[read, read, read]], mode, max_seqs=10)
        cons, cov, status = b.get_consensus()
        assert status == [0] and cons[0] == read and cov[0] == [3] * 1023


def test_add_poa_group_status_codes():
    # cudapoa/tests/Test_CudapoaBatch.cu:99-153
    from genomeworks_amd import cudapoa
    b = cudapoa.CudaPoaBatch(2, 256, 1 << 30, band_mode="full_band")
    st, seq = b.add_poa_group(["ACGT" * 4, "ACGT" * 4, "ACGT" * 4])
    assert st == 0 and seq == [0, 0, cudapoa.exceeded_maximum_sequences_per_poa]
    b.reset()
    assert b.total_poas == 0
    st, seq = b.add_poa_group(["A" * 300, "ACGT"])
    assert st == 0 and seq == [cudapoa.exceeded_maximum_sequence_size, 0]
    st, seq = b.add_poa_group(["A" * 300])
    assert st == cudapoa.empty_poa_group and seq == [cudapoa.exceeded_maximum_sequence_size]
    with pytest.raises(RuntimeError):
        cudapoa.CudaPoaBatch(2, 256, 0, band_mode="full_band")  # zero memory -> "Requires at least ..."
    with pytest.raises(ValueError):
        cudapoa.CudaPoaBatch(2, 64, 1 << 30, band_mode="full_band")  # band 256 > max_sequence_size (batch.cu:96-97)


def test_msa_bit_exact_vs_oracle():
    from genomeworks_amd import synthetic
    windows = [[r.decode() for r in synthetic.generate_window(7000 + w, 120, 12, 8, 4, 4)] for w in range(4)]
    b = run_gpu(windows, "full_band", max_seq=256, max_seqs=16, output_type="msa")
    msa, status = b.get_msa()
    with O.Workspace(oracle_cfg("full_band", 256, 16, output_mask=2)) as ws:
        for i, w in enumerate(windows):
            ref = ws.process(w)
            assert status[i] == ref["status"] == 0
            assert msa[i] == ref["msa"]
            assert [r.replace("-", "") for r in msa[i]] == w


@pytest.mark.parametrize("band_mode,band_width", [(m, w) for m in ("static_band", "adaptive_band", "static_band_traceback", "adaptive_band_traceback")
                                                  for w in (128, 256, 384, 512)])
def test_msa_through_every_packed_pass_bit_exact_vs_oracle(band_mode, band_width):
    """MSA output (the MSA instantiations of the graph-build kernel: sequence-begin and edge-coverage bookkeeping in the merge)
    through every packed forward pass -- bands 128 / 256, the two-pass bands 384 / 512, the traceback-buffer modes -- on metric
    windows and on short, deep and divergent ones: every row of every MSA equals the oracle's, and stripping the gaps gives back
    the reads."""
    from genomeworks_amd import synthetic
    windows = config3(6) + [[r.decode() for r in synthetic.generate_window(7100 + w, 700 + 40 * w, 10 + w, 40, 25, 25)] for w in range(6)]
    windows = [[r for r in w if len(r) < 1024] for w in windows]
    b = run_gpu(windows, band_mode, band_width=band_width, output_type="msa", mem=16 << 30)
    msa, status = b.get_msa()
    with O.Workspace(oracle_cfg(band_mode, band_width=band_width, output_mask=2)) as ws:
        for i, w in enumerate(windows):
            ref = ws.process(w)
            assert status[i] == ref["status"], (i, status[i], ref["status"])
            if ref["status"] == 0:
                assert msa[i] == ref["msa"], "window %d" % i
                assert [r.replace("-", "") for r in msa[i]] == w


def test_full_size_batch_properties():
    """1024 windows (BASELINE config 3): size-independent properties + parity on a sample of the same run."""
    windows = config3(1024)
    b = run_gpu(windows, "static_band")
    cons, cov, status = b.get_consensus()
    assert len(cons) == 1024 and all(s == 0 for s in status)
    assert all(900 <= len(c) <= 1000 for c in cons)
    assert all(len(c) == len(v) for c, v in zip(cons, cov))
    assert all(set(c) <= set("ACGT") for c in cons)
    # idempotence: same inputs resident in HBM, re-run -> identical outputs (buffers are reused dirty)
    b.relaunch()
    cons2, cov2, status2 = b.get_consensus()
    assert cons2 == cons and cov2 == cov and status2 == status
    # index-split invariance: a window's result does not depend on its position / batch composition
    sub = run_gpu(windows[512:520], "static_band")
    c3, v3, s3 = sub.get_consensus()
    assert c3 == cons[512:520] and v3 == cov[512:520]
    # oracle parity on a sample of the big batch
    with O.Workspace(oracle_cfg("static_band")) as ws:
        for i in (0, 511, 1023):
            ref = ws.process(windows[i])
            assert cons[i] == ref["consensus"] and cov[i] == list(ref["coverage"])


@pytest.mark.parametrize("scores", [(-3, -4, 5), (-30, -9, 7), (-1, -1, 1), (-8, 0, 8)])
def test_packed_forward_other_scores_bit_exact(scores):
    """The packed int16 forward pass / trace codes with score parameters other than the defaults (gap, mismatch, match)."""
    gap, mismatch, match = scores
    windows = config3(3, first=2100)
    b = run_gpu(windows, "static_band", gap_score=gap, mismatch_score=mismatch, match_score=match)
    cons, cov, status = b.get_consensus()
    cfg = oracle_cfg("static_band")
    cfg.gap_score, cfg.mismatch_score, cfg.match_score = gap, mismatch, match
    O.lib().poa_cfg_select_types(cfg)
    with O.Workspace(cfg) as ws:
        for i, w in enumerate(windows):
            ref = ws.process(w)
            assert status[i] == ref["status"]
            if ref["status"] == 0:
                assert cons[i] == ref["consensus"] and cov[i] == list(ref["coverage"])
        assert ws.overflow_events() == 0


def test_msa_static_band_256_bit_exact():
    """MSA output on the packed static-band path (merge with per-edge sequence coverage) vs the oracle."""
    from genomeworks_amd import synthetic
    windows = [[r.decode() for r in synthetic.generate_window(7300 + w, 400, 10, 20, 10, 10)] for w in range(3)]
    b = run_gpu(windows, "static_band", max_seq=512, max_seqs=16, output_type="msa")
    msa, status = b.get_msa()
    with O.Workspace(oracle_cfg("static_band", 512, 16, output_mask=2)) as ws:
        for i, w in enumerate(windows):
            ref = ws.process(w)
            assert status[i] == ref["status"] == 0
            assert msa[i] == ref["msa"]


@pytest.mark.parametrize("band_mode", ["static_band", "adaptive_band"])
def test_randomized_window_shapes_vs_oracle(band_mode):
    """Windows of varied length / depth / divergence (incl. N bases and heavy indels) through the band-256 path:
    consensus, coverage and status must equal the oracle's window by window."""
    import random
    from genomeworks_amd import synthetic
    rng = random.Random(20240917)
    windows = []
    for k in range(24):
        blen = rng.choice([300, 420, 640, 777, 900, 1000])
        reads = rng.choice([3, 8, 17, 32])
        mut, ins, dele = rng.choice([(5, 2, 2), (40, 20, 20), (90, 40, 40), (10, 60, 5), (10, 5, 60)])
        w = [bytearray(r) for r in synthetic.generate_window(9000 + k, blen, reads, mut, ins, dele)]
        if k % 5 == 0:  # sprinkle non-ACGT characters
            for r in w:
                for _ in range(3):
                    r[rng.randrange(len(r))] = ord("N")
        windows.append([bytes(r).decode() for r in w if len(r) < 1024])  # reads the batch would reject are left out
    b = run_gpu(windows, band_mode)
    cons, cov, status = b.get_consensus()
    with O.Workspace(oracle_cfg(band_mode)) as ws:
        for i, w in enumerate(windows):
            ref = ws.process(w)
            assert status[i] == ref["status"], (i, status[i], ref["status"])
            if ref["status"] == 0:
                assert cons[i] == ref["consensus"], "window %d consensus differs" % i
                assert cov[i] == list(ref["coverage"]), "window %d coverage differs" % i


BAND_TABLE = [(m, w) for m in ("static_band", "adaptive_band", "static_band_traceback", "adaptive_band_traceback") for w in (128, 256, 384, 512)]


@pytest.mark.parametrize("band_mode,band_width", BAND_TABLE)
def test_band_mode_table_equals_the_golden(band_mode, band_width):
    """EVERY cell of the band-mode x band-width table (multiples of 128, cudapoa/src/batch.cu:41; both traceback-buffer
    modes, cudapoa_nw_tb_banded.cuh:264-643) on ALL 1024 config-3 windows against the committed oracle goldens
    (tests/golden/make_band_mode_goldens.py): status, consensus and coverage of every window, and the cell total."""
    import golden_io as G
    s = G.band_mode_summary()
    gold = G.band_mode_goldens()
    mi, wi = s["modes"].index(band_mode), s["widths"].index(band_width)
    n = s["windows"]
    windows = config3(n, first=s["first_seed"])
    b = run_gpu(windows, band_mode, band_width=band_width, mem=24 << 30)  # adaptive modes: 2 x (band + 8) columns per row
    cons, cov, status = b.get_consensus()
    fp = G.band_mode_fingerprints(cons, cov, status)
    bad = [w for w in range(n) if fp[w] != gold["fingerprint"][mi, wi, w] or int(status[w]) != int(gold["status"][mi, wi, w])]
    assert not bad, "windows that differ from the oracle golden: %s" % bad[:20]
    assert b.total_cells() == s["cells"]["%s/%d" % (band_mode, band_width)]["cells"]
    assert G.band_gen.cell_digest(fp) == s["cells"]["%s/%d" % (band_mode, band_width)]["fingerprint_sha256"]


@pytest.mark.parametrize("band_mode,band_width", [(m, w) for m, w in BAND_TABLE if w != 256 and not (w == 128 and "traceback" not in m)])
def test_randomized_window_shapes_at_every_band_width_vs_oracle(band_mode, band_width):
    """The cells of the table that test_randomized_window_shapes_vs_oracle (band 256) and the band-128 test do not reach:
    windows of varied length / depth / divergence with heavy indels (alignment paths near and beyond the band edges,
    adaptive widening and reruns, traceback-buffer distance limits) against the live oracle, window by window."""
    import random
    from genomeworks_amd import synthetic
    rng = random.Random(4000 + band_width)
    windows = []
    for k in range(28):
        blen = rng.choice([200, 420, 640, 777, 900, 1000])
        reads = rng.choice([3, 8, 17, 32])
        mut, ins, dele = rng.choice([(5, 2, 2), (40, 20, 20), (90, 40, 40), (10, 60, 5), (10, 5, 60), (20, 150, 10), (20, 10, 150)])
        w = [r.decode() for r in synthetic.generate_window(9400 + k, blen, reads, mut, ins, dele)]
        w = [r for r in w if 2 < len(r) < 1024]
        if len(w) >= 2:
            windows.append(w)
    b = run_gpu(windows, band_mode, band_width=band_width)
    cons, cov, status = b.get_consensus()
    cells_ref = 0
    with O.Workspace(oracle_cfg(band_mode, band_width=band_width)) as ws:
        for i, w in enumerate(windows):
            ref = ws.process(w)
            cells_ref += ref["cells"]
            assert status[i] == ref["status"], (i, status[i], ref["status"])
            if ref["status"] == 0:
                assert cons[i] == ref["consensus"], "window %d consensus differs" % i
                assert cov[i] == list(ref["coverage"]), "window %d coverage differs" % i
    assert b.total_cells() == cells_ref


def test_c_abi_accessors_report_a_bad_index_instead_of_throwing():
    """gw_poa_consensus_str / _coverage / _output_status with an index outside the last get_consensus(): null / -1 and an
    error string, no exception across the C ABI (ADVICE r3 / VERDICT r3 item 8)."""
    import ctypes as C
    b = run_gpu([["ACGTACGTAC", "ACGTTCGTAC"]], "static_band", max_seq=256, max_seqs=4)
    b.get_consensus()
    ln = C.c_int32(0)
    assert not b._L.gw_poa_consensus_str(b._h, 7, C.byref(ln))
    assert not b._L.gw_poa_consensus_coverage(b._h, -1, C.byref(ln))
    assert b._L.gw_poa_output_status(b._h, 99) == -1


def test_long_read_adaptive_msa_32bit_path():
    """BASELINE configs[3] at reduced scale: long reads (7 kbp), adaptive band, MSA output -> 32-bit scores and ids,
    HBM row table, multi-pass band; bit-exact vs the oracle."""
    from genomeworks_amd import synthetic
    windows = [[r.decode() for r in synthetic.generate_window(8800 + w, 7000, 6, 350, 120, 120)] for w in range(2)]
    b = run_gpu(windows, "adaptive_band", max_seq=8192, max_seqs=8, output_type="msa", nodes=4 * 8192)
    msa, status = b.get_msa()
    cfg = oracle_cfg("adaptive_band", 8192, 8, output_mask=2, nodes=4 * 8192)
    assert cfg.score32 == 1
    with O.Workspace(cfg) as ws:
        for i, w in enumerate(windows):
            ref = ws.process(w)
            assert status[i] == ref["status"] == 0
            assert msa[i] == ref["msa"]
            assert [r.replace("-", "") for r in msa[i]] == w


def test_long_read_forward_and_traceback_variants_agree(monkeypatch):
    """A/B inside the long-read kernel (graphs beyond the LDS tables, adaptive band): the pipelined multi-wave forward
    pass with trace codes and the table-lookup traceback (default) against the table-lookup traceback switched off
    (GWHIP_DEBUG bit 6: recomputation from the score matrix), against the single-wave forward pass (bit 18) and against the
    full topological re-sorts (bit 17: cached Kahn order, bit 21: serial; default: incremental order with block replay and
    its hot state in LDS; bit 16: the same with its state in HBM), on
    divergent long reads whose bands widen to 512 .. 1536 columns: identical MSA, status and cell counts, and equal to
    the oracle."""
    from genomeworks_amd import synthetic
    windows = [[r.decode() for r in synthetic.generate_window(9100 + w, 5200 + 700 * w, 7, 260, 330, 330)] for w in range(3)]
    windows.append([r.decode() for r in synthetic.generate_window(9200, 6000, 10, 900, 40, 40)])   # four predecessors and more
    windows.append([r.decode() for r in synthetic.generate_window(9201, 3000, 5, 20, 900, 20)])    # reads much longer than the backbone
    out = {}
    for name, flag in (("default", None), ("recomputed_traceback", str(1 << 6)), ("single_wave", str(1 << 18)),
                       ("cached_full_resort", str(1 << 17)), ("serial_full_resort", str(1 << 21)),
                       ("incremental_in_hbm", str(1 << 16))):
        if flag is None:
            monkeypatch.delenv("GWHIP_DEBUG", raising=False)
        else:
            monkeypatch.setenv("GWHIP_DEBUG", flag)
        b = run_gpu(windows, "adaptive_band", max_seq=8192, max_seqs=12, output_type="msa", nodes=4 * 8192)
        out[name] = (b.get_msa(), b.total_cells())
    assert out["default"] == out["recomputed_traceback"]
    assert out["default"] == out["single_wave"]
    assert out["default"] == out["cached_full_resort"]   # incremental Kahn order (default) vs the cached full re-sort
    assert out["default"] == out["incremental_in_hbm"]   # ... LDS counters and window vs node words in HBM
    # MSA rows scattered by one lane per node (default) against one lane per sequence walking its path (and the serial
    # racon order): GWHIP_MSA_SERIAL=1
    monkeypatch.delenv("GWHIP_DEBUG", raising=False)
    monkeypatch.setenv("GWHIP_MSA_SERIAL", "1")
    b = run_gpu(windows, "adaptive_band", max_seq=8192, max_seqs=12, output_type="msa", nodes=4 * 8192)
    assert (b.get_msa(), b.total_cells()) == out["default"]
    short = [[r.decode() for r in synthetic.generate_window(9300 + w, 400, 14, 20, 12, 12)] for w in range(6)]
    short.append(["ACGTACGTAC"])                              # a window of one read
    short.append(["ACGTACGTAC", "A", "ACGTACGTACGGGT"])       # a read of one base
    walked = run_gpu(short, "static_band", max_seq=512, max_seqs=16, output_type="msa").get_msa()
    monkeypatch.delenv("GWHIP_MSA_SERIAL", raising=False)
    assert run_gpu(short, "static_band", max_seq=512, max_seqs=16, output_type="msa").get_msa() == walked
    assert out["default"] == out["serial_full_resort"]   # ... vs the reference's schedule on one lane
    (msa, status), _ = out["default"]
    cfg = oracle_cfg("adaptive_band", 8192, 12, output_mask=2, nodes=4 * 8192)
    with O.Workspace(cfg) as ws:
        for i, w in enumerate(windows):
            ref = ws.process(w)
            assert status[i] == ref["status"]
            if ref["status"] == 0:
                assert msa[i] == ref["msa"]


def test_group_with_every_read_rejected_keeps_its_output_slot():
    """Reference quirk (cudapoa_batch.cuh:122-150): add_poa_group opens the POA before it tries the reads, so a group
    whose reads are all rejected returns empty_poa_group but stays in the batch with zero reads and owns an output
    entry; callers that skip it (cudapoa/src/main.cpp:312-318 does) must still count that slot. The neighbours are
    unaffected and bit-exact."""
    from genomeworks_amd import cudapoa, synthetic
    good = [[r.decode() for r in synthetic.generate_window(4100 + w, 300, 6, 15, 6, 6)] for w in range(2)]
    too_long = ["ACGT" * 200, "TTGCA" * 150]  # both longer than max_sequence_size = 512
    b = cudapoa.CudaPoaBatch(8, 512, 1 << 30, output_type="consensus", band_mode="static_band", max_nodes_per_graph=1536)
    assert b.add_poa_group(good[0])[0] == 0
    st, seq_st = b.add_poa_group(too_long)
    assert st == cudapoa.empty_poa_group
    assert seq_st == [cudapoa.exceeded_maximum_sequence_size] * 2
    assert b.add_poa_group(good[1])[0] == 0
    assert b.total_poas == 3
    b.generate_poa()
    cons, cov, status = b.get_consensus()
    assert len(cons) == len(status) == 3
    cfg = oracle_cfg("static_band", 512, 8, nodes=1536)
    with O.Workspace(cfg) as ws:
        for slot, w in ((0, good[0]), (2, good[1])):
            ref = ws.process(w)
            assert status[slot] == 0 and cons[slot] == ref["consensus"]
            assert list(cov[slot]) == list(ref["coverage"])


@pytest.mark.parametrize("band_mode", ["static_band", "adaptive_band"])
def test_band_128_through_the_packed_pass_bit_exact_vs_oracle(band_mode):
    """alignment_band_width 128 (the reference's minimum, cudapoa/src/batch.cu:41) takes the packed forward pass and the
    move-byte traceback with the band in lanes 0..31: consensus, coverage, status and cell counts equal the oracle's on
    config-3 windows and on windows with heavy indels (paths near the band edges, adaptive widening to 256)."""
    import random
    from genomeworks_amd import synthetic
    rng = random.Random(128)
    windows = config3(24)
    for k in range(24):
        blen = rng.choice([200, 500, 900, 1000])
        mut, ins, dele = rng.choice([(5, 2, 2), (40, 20, 20), (90, 40, 40), (10, 60, 5), (10, 5, 60)])
        w = [r.decode() for r in synthetic.generate_window(8800 + k, blen, rng.choice([5, 16, 32]), mut, ins, dele)]
        windows.append([r for r in w if 130 < len(r) < 1024])
    windows = [w for w in windows if len(w) >= 2]
    b = run_gpu(windows, band_mode, band_width=128)
    cons, cov, status = b.get_consensus()
    cells_ref = 0
    with O.Workspace(oracle_cfg(band_mode, band_width=128)) as ws:
        for i, w in enumerate(windows):
            ref = ws.process(w)
            cells_ref += ref["cells"]
            assert status[i] == ref["status"], (i, status[i], ref["status"])
            if ref["status"] == 0:
                assert cons[i] == ref["consensus"], "window %d consensus differs" % i
                assert cov[i] == list(ref["coverage"]), "window %d coverage differs" % i
        assert ws.overflow_events() == 0
    assert b.total_cells() == cells_ref


def test_kernel_shortcuts_equal_the_plain_schedule(monkeypatch):
    """A/B inside the kernel (GWHIP_DEBUG selectors of the debug instantiation; the production instantiation is the first
    arm): the incremental Kahn order vs the full re-sort after every read (bit 21, the reference's schedule); rows with
    4..6 predecessors in the LDS-ring kind vs the general routine (bit 30); the row kinds of the forward pass demoted into
    each other -- register rows through the ring (bit 10), moved-band rows through the ring (bit 15), ring rows through the
    general routine (bit 9), register rows through the general routine (bit 11) -- and the consensus kernel's first
    heaviest-bundle pass node by node instead of 64 positions at a time (bit 4) must give identical consensus, coverage,
    status and cell counts on config-3 windows and on windows of varied shape."""
    import random
    from genomeworks_amd import synthetic
    rng = random.Random(5)
    windows = config3(192)
    for k in range(64):
        blen = rng.choice([40, 130, 300, 640, 900, 1000])
        reads = rng.choice([2, 3, 8, 17, 32])
        mut, ins, dele = rng.choice([(0, 0, 0), (5, 2, 2), (40, 20, 20), (90, 40, 40), (10, 60, 5), (10, 5, 60)])
        w = [r.decode() for r in synthetic.generate_window(7000 + k, blen, reads, mut, ins, dele)]
        if k % 4 == 0:
            w = [("GATTACA"[: rng.randrange(8)] + r)[rng.randrange(5):] for r in w]
        windows.append([r for r in w if 0 < len(r) < 1024])
    out = {}
    arms = (("production", None), ("full_resort", str(1 << 21)), ("many_predecessors_general", str(1 << 30)),
            ("plain", str((1 << 21) | (1 << 30))), ("registers_through_ring", str(1 << 10)), ("moved_band_through_ring", str(1 << 15)),
            ("ring_through_general", str(1 << 9)), ("registers_through_general", str(1 << 11)),
            ("everything_general", str((1 << 9) | (1 << 11) | (1 << 30))), ("consensus_node_by_node", str(1 << 4)),
            # round 5: the production pass keeps the score rows nobody reads out of HBM and reruns a read whose walk needs one
            # after all (kNwNeedScoreRows); bit 25 stores every row as rounds 1-4 did
            ("every_row_stores_scores", str(1 << 25)), ("every_row_stores_scores_registers_through_ring", str((1 << 25) | (1 << 10))),
            # ... and bit 23 sends EVERY read through that rerun (any recomputed step below row 0 ends the first walk)
            ("score_row_rerun_on_every_read", str(1 << 23)))
    for name, flag in arms:
        if flag is None:
            monkeypatch.delenv("GWHIP_DEBUG", raising=False)
        else:
            monkeypatch.setenv("GWHIP_DEBUG", flag)
        for mode in ("static_band", "adaptive_band"):
            b = run_gpu(windows, mode)
            out[name, mode] = (b.get_consensus(), b.total_cells())
    for mode in ("static_band", "adaptive_band"):
        for name, _ in arms[1:]:
            assert out["production", mode] == out[name, mode], (name, mode)


@pytest.mark.parametrize("band_width", [384, 512])
def test_two_pass_kernel_shortcuts_equal_the_plain_schedule(monkeypatch, band_width):
    """The same A/B for the two-pass packed pass of bands 384 / 512 (poa_forward_moves_wide.h; debug instantiation VARIANT 4):
    register rows through the 4-row ring (bit 10), moved-band rows through the ring (bit 15), ring rows through the general
    routine (bit 9), register rows through the general routine (bit 11), rows with 4..6 predecessors through the general
    routine (bit 30), the packed pass switched off altogether (bit 8: the generic multi-pass loop and its traceback, the
    path these widths took in round 3) -- identical consensus, coverage, status and cell counts, and equal to the oracle's
    on the production arm."""
    import random
    from genomeworks_amd import synthetic
    rng = random.Random(band_width)
    windows = config3(96)
    for k in range(48):
        blen = rng.choice([520, 640, 900, 1000])
        reads = rng.choice([3, 8, 17, 32])
        mut, ins, dele = rng.choice([(0, 0, 0), (5, 2, 2), (40, 20, 20), (90, 40, 40), (10, 60, 5), (10, 5, 60)])
        w = [r.decode() for r in synthetic.generate_window(7300 + k, blen, reads, mut, ins, dele)]
        if k % 4 == 0:
            w = [("GATTACA"[: rng.randrange(8)] + r)[rng.randrange(5):] for r in w]
        windows.append([r for r in w if 0 < len(r) < 1024])
    out = {}
    arms = (("production", None), ("registers_through_ring", str(1 << 10)), ("moved_band_through_ring", str(1 << 15)),
            ("ring_through_general", str(1 << 9)), ("registers_through_general", str(1 << 11)),
            ("many_predecessors_general", str(1 << 30)), ("everything_general", str((1 << 9) | (1 << 11) | (1 << 30))),
            ("generic_passes", str(1 << 8)))
    for name, flag in arms:
        if flag is None:
            monkeypatch.delenv("GWHIP_DEBUG", raising=False)
        else:
            monkeypatch.setenv("GWHIP_DEBUG", flag)
        for mode in ("static_band", "adaptive_band"):
            b = run_gpu(windows, mode, band_width=band_width, mem=16 << 30)
            out[name, mode] = (b.get_consensus(), b.total_cells())
    for mode in ("static_band", "adaptive_band"):
        for name, _ in arms[1:]:
            assert out["production", mode] == out[name, mode], (name, mode)
        (cons, cov, status), cells = out["production", mode]
        # (the arms above compare every window with every other arm; the oracle, the slow part, sees every third window --
        # the band-table and randomized-shape tests put every cell of the table against it on their own)
        with O.Workspace(oracle_cfg(mode, band_width=band_width)) as ws:
            for i, w in enumerate(windows):
                if i % 3:
                    continue
                ref = ws.process(w)
                assert status[i] == ref["status"], (mode, i)
                if ref["status"] == 0:
                    assert cons[i] == ref["consensus"] and cov[i] == list(ref["coverage"]), (mode, i)


@pytest.mark.parametrize("band_width", [128, 256, 384, 512])
def test_traceback_buffer_packed_pass_equals_the_memory_faithful_routine(monkeypatch, band_width):
    """The packed pass of the traceback-buffer modes (poa_forward_moves_tb.h: 16-bit pairs, the trace matrix as two byte
    planes, sheared-tile walk) against the memory-faithful routine of poa_tb_device.h (GWHIP_DEBUG bit 8) and against its own
    row kinds demoted into each other (bits 9, 10, 11, 15, 30), on config-3 windows and on windows of varied shape with heavy
    indels -- identical consensus, coverage, status and cell counts, and equal to the oracle's on the production arm."""
    import random
    from genomeworks_amd import synthetic
    rng = random.Random(1000 + band_width)
    windows = config3(96)
    for k in range(56):
        blen = rng.choice([40, 130, 300, 640, 900, 1000] if band_width <= 256 else [520, 640, 900, 1000])
        reads = rng.choice([2, 3, 8, 17, 32])
        mut, ins, dele = rng.choice([(0, 0, 0), (5, 2, 2), (40, 20, 20), (90, 40, 40), (10, 60, 5), (10, 5, 60)])
        w = [r.decode() for r in synthetic.generate_window(7600 + k, blen, reads, mut, ins, dele)]
        if k % 4 == 0:
            w = [("GATTACA"[: rng.randrange(8)] + r)[rng.randrange(5):] for r in w]
        windows.append([r for r in w if 0 < len(r) < 1024])
    out = {}
    arms = (("production", None), ("memory_faithful_routine", str(1 << 8)), ("registers_through_ring", str(1 << 10)),
            ("moved_band_through_ring", str(1 << 15)), ("ring_through_general", str(1 << 9)),
            ("registers_through_general", str(1 << 11)), ("many_predecessors_general", str(1 << 30)),
            ("everything_general", str((1 << 9) | (1 << 11) | (1 << 30))))
    modes = ("static_band_traceback", "adaptive_band_traceback")
    for name, flag in arms:
        if flag is None:
            monkeypatch.delenv("GWHIP_DEBUG", raising=False)
        else:
            monkeypatch.setenv("GWHIP_DEBUG", flag)
        for mode in modes:
            b = run_gpu(windows, mode, band_width=band_width, mem=16 << 30)
            out[name, mode] = (b.get_consensus(), b.total_cells())
    for mode in modes:
        for name, _ in arms[1:]:
            assert out["production", mode] == out[name, mode], (name, mode)
        (cons, cov, status), cells = out["production", mode]
        # (the arms above compare every window with every other arm; the oracle, the slow part, sees every third window --
        # the band-table and randomized-shape tests put every cell of the table against it on their own)
        with O.Workspace(oracle_cfg(mode, band_width=band_width)) as ws:
            for i, w in enumerate(windows):
                if i % 3:
                    continue
                ref = ws.process(w)
                assert status[i] == ref["status"], (mode, i)
                if ref["status"] == 0:
                    assert cons[i] == ref["consensus"] and cov[i] == list(ref["coverage"]), (mode, i)


def test_traceback_buffer_modes_with_a_short_predecessor_window_vs_oracle():
    """max_banded_pred_distance of 8 and 20 (int8 traces: the memory-faithful routine; predecessors that far up are
    skipped) and of 130 (int16 trace region: the packed pass with a window shorter than its own 126-row limit) -- status,
    consensus and coverage equal the oracle's."""
    from genomeworks_amd import cudapoa, synthetic
    windows = config3(12) + [[r.decode() for r in synthetic.generate_window(7900 + k, 700, 12, 40, 30, 30)] for k in range(12)]
    for mode in ("static_band_traceback", "adaptive_band_traceback"):
        for H in (8, 20, 130):
            b = cudapoa.CudaPoaBatch(32, 1024, 8 << 30, output_type="consensus", band_mode=mode, alignment_band_width=256,
                                     max_nodes_per_graph=3072, max_banded_pred_distance=H)
            for w in windows:
                assert b.add_poa_group(w)[0] == 0
            b.generate_poa()
            cons, cov, status = b.get_consensus()
            cfg = oracle_cfg(mode)
            cfg.max_banded_pred_distance = H
            O.lib().poa_cfg_select_types(cfg)
            with O.Workspace(cfg) as ws:
                for i, w in enumerate(windows):
                    ref = ws.process(w)
                    assert status[i] == ref["status"], (mode, H, i, status[i], ref["status"])
                    if ref["status"] == 0:
                        assert cons[i] == ref["consensus"] and cov[i] == list(ref["coverage"]), (mode, H, i)


def test_consensus_kernel_with_small_lds_tables_and_oversized_graphs():
    """More than 512 windows: the consensus kernel sizes its LDS tables for 2176 nodes (four blocks per CU) and a
    window whose graph is larger takes the HBM routine inside the same launch. Results must equal those of the same
    windows in a small batch (full-size tables) and the oracle's."""
    from genomeworks_amd import synthetic
    small = [[r.decode() for r in synthetic.generate_window(4000 + k, 90, 3, 4, 2, 2)] for k in range(520)]
    big = [[r.decode() for r in synthetic.generate_window(4600 + k, 980, 14, 400, 100, 100)] for k in range(6)]  # ~2450 nodes
    big = [[r for r in w if len(r) < 1024] for w in big]
    windows = small[:300] + big + small[300:]
    b = run_gpu(windows, "static_band", max_seqs=14)
    cons, cov, status = b.get_consensus()
    sub = run_gpu(big + small[:4], "static_band", max_seqs=14)
    c2, v2, s2 = sub.get_consensus()
    assert cons[300:306] == c2[:6] and cov[300:306] == v2[:6] and status[300:306] == s2[:6]
    assert cons[:4] == c2[6:10]
    n_big = 0
    with O.Workspace(oracle_cfg("static_band", max_seqs=14)) as ws:
        for i in range(300, 306):
            ref = ws.process(windows[i])
            assert status[i] == ref["status"]
            if ref["status"] == 0:
                assert cons[i] == ref["consensus"] and cov[i] == list(ref["coverage"])
                n_big += ref["node_count"] > 2176
    assert n_big >= 3, n_big  # the HBM routine really ran


def test_high_degree_graphs_vs_oracle():
    """Windows built to leave the common case of the LDS fast paths: many reads that each carry a DIFFERENT base or
    insertion at the same backbone positions (nodes with more than six in- and out-edges: the side tables of the
    forward pass and of the topological sort fall back to the HBM lists), long parallel branches (wide Kahn queues,
    beyond the 4-bit queue-length field), reads that start and end at different offsets (several sources and sinks)."""
    import random
    rng = random.Random(99)
    windows = []
    for k in range(10):
        backbone = "".join(rng.choice("ACGT") for _ in range(rng.choice([200, 400, 700])))
        reads = [backbone]
        hot = sorted(rng.sample(range(20, len(backbone) - 20), 6))
        for r in range(rng.choice([12, 24, 31])):
            s = list(backbone)
            for h in hot:
                kind = rng.randrange(4)
                if kind == 0:
                    s[h] = rng.choice("ACGT")                                             # many alternative bases
                elif kind == 1:
                    s[h] = s[h] + "".join(rng.choice("ACGT") for _ in range(rng.randrange(1, 12)))  # distinct insertions
                elif kind == 2:
                    for d in range(rng.randrange(1, 9)):                                  # deletions of different lengths
                        s[h + d] = ""
            t = "".join(s)
            a, b = rng.randrange(0, 8), rng.randrange(0, 8)
            reads.append(t[a:len(t) - b] if k % 2 else t)
        windows.append(reads)
    for mode in ("static_band", "adaptive_band"):
        b = run_gpu(windows, mode)
        cons, cov, status = b.get_consensus()
        with O.Workspace(oracle_cfg(mode)) as ws:
            for i, w in enumerate(windows):
                ref = ws.process(w)
                assert status[i] == ref["status"], (mode, i, status[i], ref["status"])
                if ref["status"] == 0:
                    assert cons[i] == ref["consensus"] and cov[i] == list(ref["coverage"]), (mode, i)
            assert ws.overflow_events() == 0
    # MSA on the same graphs (racon order in the output kernel) through the same build kernel
    b = run_gpu(windows, "static_band", output_type="msa")
    msa, status = b.get_msa()
    with O.Workspace(oracle_cfg("static_band", output_mask=2)) as ws:
        for i, w in enumerate(windows):
            ref = ws.process(w)
            assert status[i] == ref["status"]
            if ref["status"] == 0:
                assert msa[i] == ref["msa"], i


@pytest.mark.parametrize("band_mode", ["static_band", "full_band", "adaptive_band"])
def test_spoa_accurate_topological_order_bit_exact(band_mode, monkeypatch):
    """The reference's -Dspoa_accurate=ON build (racon-style topological sort after every read, cudapoa_topsort.cuh:45-128)
    as a run-time flag of the same kernels: consensus, coverage and MSA equal the oracle run with spoa_accurate = 1."""
    from genomeworks_amd import synthetic
    monkeypatch.setenv("GW_SPOA_ACCURATE", "1")
    windows = [[r.decode() for r in synthetic.generate_window(8100 + w, 400, 12, 24, 12, 12)] for w in range(4)]
    b = run_gpu(windows, band_mode, max_seq=512, max_seqs=16)
    cons, cov, status = b.get_consensus()
    m = run_gpu(windows, band_mode, max_seq=512, max_seqs=16, output_type="msa")
    msa, mstatus = m.get_msa()
    monkeypatch.delenv("GW_SPOA_ACCURATE")
    plain = run_gpu(windows, band_mode, max_seq=512, max_seqs=16)
    pcons, _, _ = plain.get_consensus()
    for mask, check in ((1, "consensus"), (2, "msa")):
        cfg = oracle_cfg(band_mode, 512, 16, output_mask=mask)
        cfg.spoa_accurate = 1
        with O.Workspace(cfg) as ws:
            for i, w in enumerate(windows):
                ref = ws.process(w)
                assert ref["status"] == 0
                if check == "consensus":
                    assert status[i] == 0 and cons[i] == ref["consensus"] and cov[i] == list(ref["coverage"])
                else:
                    assert mstatus[i] == 0 and msa[i] == ref["msa"]
    assert len(pcons) == len(cons)  # the default order still runs in the same process (the flag is read per batch)


def test_full_band_benchmark_shape_equals_the_golden():
    """The 17th cell of the band-mode table: the reference benchmarks' own BatchConfig(1024, 200) = FULL band
    (cudapoa/benchmarks/single_batch.hpp:52, multi_batch.hpp:49; kernel cudapoa_nw.cuh:149-454) on ALL 1024 metric windows
    against the committed oracle golden (tests/golden/make_full_band_goldens.py): status, consensus and coverage of every
    window, the batch's cell count, and a relaunch on the dirty buffers."""
    import golden_io as G
    from genomeworks_amd import cudapoa
    s, g = G.full_band_summary(), G.full_band_goldens()
    windows = config3(s["windows"], s["first_seed"])
    b = cudapoa.CudaPoaBatch(200, 1024, 16 << 30, output_type="consensus", band_mode="full_band", max_nodes_per_graph=3072,
                             matrix_sequence_dimension=1024)
    for w in windows:
        st, _ = b.add_poa_group(w)
        assert st == 0
    for launch in range(2):
        b.generate_poa() if launch == 0 else b.relaunch()
        cons, cov, status = b.get_consensus()
        fp = G.band_mode_fingerprints(cons, cov, status)
        bad = np.nonzero(fp != g["fingerprint"])[0]
        assert len(bad) == 0, (launch, bad[:10])
        assert [int(x) for x in status] == [int(x) for x in g["status"]]
        assert b.total_cells() == s["cells"]
        assert G.band_gen.cell_digest(fp) == s["fingerprint_sha256"]


def test_full_band_packed_pass_equals_the_generic_routine_and_the_oracle(monkeypatch):
    """The packed full-band pass (poa_forward_moves_full.h; 1 .. 4 register passes of 256 columns per row, predecessors from
    registers / the 4-row LDS ring / the HBM matrix, move bytes + sheared-tile walk; debug instantiation VARIANT 6) against
    the generic nw_full (GWHIP_DEBUG bit 8) and against its own row kinds demoted into each other (bits 9, 10, 11, 30), on
    config-3 windows and on windows whose reads end in every pass (lengths around 256 / 512 / 768 / 1023, a read of exactly
    1024 bases that the pass hands to the generic routine, single-base and very short reads, heavy indels: branchy graphs
    with 4+ predecessors) -- identical consensus, coverage, status and cell counts, and equal to the oracle's."""
    import random
    from genomeworks_amd import synthetic
    rng = random.Random(20250926)
    windows = config3(48)
    for k in range(64):
        blen = rng.choice([3, 40, 130, 250, 256, 257, 300, 500, 512, 513, 640, 766, 769, 900, 1000, 1019])
        reads = rng.choice([2, 3, 8, 17, 32])
        mut, ins, dele = rng.choice([(0, 0, 0), (5, 2, 2), (40, 20, 20), (90, 40, 40), (10, 60, 5), (10, 5, 60)])
        mut, ins, dele = (min(x, max(1, blen // 4)) for x in (mut, ins, dele))
        w = [r.decode() for r in synthetic.generate_window(9100 + k, blen, reads, mut, ins, dele)]
        if k % 4 == 0:
            w = [("GATTACA"[: rng.randrange(8)] + r)[rng.randrange(5):] for r in w]
        if k % 9 == 0:   # reads much shorter than the backbone: fewer passes than the widest read of the window
            w = w[:1] + [r[: max(1, len(r) // rng.choice([2, 3, 5]))] for r in w[1:]]
        if k % 16 == 5:  # a read of exactly max_sequence_size (1024): the packed pass declines it
            w.append((w[-1] * (1024 // max(1, len(w[-1])) + 1))[:1024])
        if k % 16 == 6:  # 1023: the longest read the pass takes (four full passes but one column)
            w.append((w[-1] * (1023 // max(1, len(w[-1])) + 1))[:1023])
        windows.append([r for r in w if 0 < len(r) <= 1024][:32])
    out = {}
    arms = (("production", None), ("registers_through_ring", str(1 << 10)), ("ring_through_general", str(1 << 9)),
            ("registers_through_general", str(1 << 11)), ("many_predecessors_general", str(1 << 30)),
            ("everything_general", str((1 << 9) | (1 << 11) | (1 << 30))), ("generic_nw_full", str(1 << 8)),
            # the production pass writes a score row to HBM only when somebody will read it back (general rows' predecessors,
            # rows with undecided cells and their predecessors, sinks): bit 25 stores every row
            ("every_row_stores_scores", str(1 << 25)), ("every_row_stores_scores_ring_through_general", str((1 << 25) | (1 << 9))))
    for name, flag in arms:
        if flag is None:
            monkeypatch.delenv("GWHIP_DEBUG", raising=False)
        else:
            monkeypatch.setenv("GWHIP_DEBUG", flag)
        b = run_gpu(windows, "full_band", mem=16 << 30)
        out[name] = (b.get_consensus(), b.total_cells())
    monkeypatch.delenv("GWHIP_DEBUG", raising=False)
    for name, _ in arms[1:]:
        (c0, v0, s0), n0 = out["production"]
        (c1, v1, s1), n1 = out[name]
        bad = [i for i in range(len(windows)) if (c0[i], v0[i], s0[i]) != (c1[i], v1[i], s1[i])]
        assert not bad and n0 == n1, (name, bad[:8], n0, n1)
    (cons, cov, status), cells = out["production"]
    cells_ref = 0
    with O.Workspace(oracle_cfg("full_band")) as ws:
        for i, w in enumerate(windows):
            ref = ws.process(w)
            cells_ref += ref["cells"]
            assert status[i] == ref["status"], i
            if ref["status"] == 0:
                assert cons[i] == ref["consensus"] and cov[i] == list(ref["coverage"]), i
        assert ws.overflow_events() == 0
    assert cells == cells_ref


def test_hip_path_equals_the_reference_itself_on_the_simt_goldens():
    """tests/golden/reference_simt_windows.json.gz holds what the REFERENCE's own cudapoa library answered (its CUDA sources
    compiled from /root/reference and run on the CPU by the SIMT emulator of oracle/simt; tests/golden/make_reference_simt_goldens.py):
    every band mode, consensus and MSA, other scores, per-base weights, graphs that outgrow max_nodes_per_graph, reads that
    add_poa_group refuses. The HIP path through the Python API, with the batch sized by the same BatchConfig constructor,
    gives the same add_poa_group statuses, window statuses, consensus, coverage and MSA rows."""
    import gzip
    import json
    import os
    from genomeworks_amd import cudapoa
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_simt_windows.json.gz"), "rb") as f:
        rows = json.loads(f.read().decode())["windows"]
    names = {v: k for k, v in BAND.items()}
    assert len(rows) >= 77
    bad = []
    for i, r in enumerate(rows):
        c, ref, bc = r["case"], r["reference"], r["reference"]["batch_config"]
        msa = bool(c["output_mask"] & 2)
        # the constructor the generator used on the reference: BatchConfig(max_seq_sz, max_seq_per_poa, band_width, banding)
        b = cudapoa.CudaPoaBatch.from_batch_config(c["max_seq"], c["max_seqs"], c["band_width"], names[c["band_mode"]], 1 << 30,
                                                   output_type="msa" if msa else "consensus", gap_score=c["gap"], mismatch_score=c["mismatch"],
                                                   match_score=c["match"], max_banded_pred_distance=c.get("max_pred", 0))
        got = b.batch_size
        assert [got.max_sequence_size, got.max_consensus_size, got.max_nodes_per_graph, got.matrix_sequence_dimension, got.alignment_band_width,
                got.max_sequences_per_poa, got.band_mode, got.max_banded_pred_distance] == [bc[k] for k in (
                    "max_sequence_size", "max_consensus_size", "max_nodes_per_graph", "matrix_sequence_dimension", "alignment_band_width",
                    "max_sequences_per_poa", "band_mode", "max_banded_pred_distance")], i
        st, seq_st = b.add_poa_group(c["reads"], c["weights"])
        if (st, list(seq_st)) != (ref["add_status"], ref["read_status"]):
            bad.append((i, "add_poa_group", st, list(seq_st)))
            continue
        b.generate_poa()
        if msa:
            rows_msa, status = b.get_msa()
            if status[0] != ref["status"] or (status[0] == 0 and rows_msa[0] != ref["msa"]):
                bad.append((i, "msa", status[0]))
        else:
            cons, cov, status = b.get_consensus()
            if status[0] != ref["status"] or (status[0] == 0 and (cons[0], list(cov[0])) != (ref["consensus"], ref["coverage"])):
                bad.append((i, "consensus", status[0]))
    assert not bad, "windows where the HIP path differs from the reference: %s" % bad[:10]
