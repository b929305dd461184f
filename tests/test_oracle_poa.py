"""Pins the CPU oracle to the reference's own inline known-answer vectors."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_poa as O

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "cudapoa_vectors.json")) as f:
    V = json.load(f)


def csv(a):
    return ",".join(str(int(x)) for x in a)


@pytest.mark.parametrize("case", V["nw"], ids=[c["name"] for c in V["nw"]])
def test_nw_full_known_answers(case):
    # Test_CudapoaNW.cu:83-189,295-301 (default BatchConfig => full band, msd 1024)
    cfg = O.make_cfg()
    g = O.graph_buffers(case["nodes"], case["outgoing"], cfg.max_nodes_per_graph, case["sorted"])
    n, ag, ar = O.run_nw(cfg, g, case["read"], "full")
    assert (csv(ag), csv(ar)) == (case["graph_ans"], case["read_ans"])


@pytest.mark.parametrize("mode", ["static", "adaptive", "static_tb", "adaptive_tb"])
@pytest.mark.parametrize("case", V["nw"], ids=[c["name"] for c in V["nw"]])
def test_nw_banded_small_cases_match_full(case, mode):
    bm = {"static": 1, "adaptive": 2, "static_tb": 3, "adaptive_tb": 4}[mode]
    cfg = O.make_cfg(1024, 2, 128, bm)
    g = O.graph_buffers(case["nodes"], case["outgoing"], cfg.max_nodes_per_graph, case["sorted"])
    n, ag, ar = O.run_nw(cfg, g, case["read"], mode)
    assert (csv(ag), csv(ar)) == (case["graph_ans"], case["read_ans"])


@pytest.mark.parametrize("mode", ["static", "adaptive", "static_tb", "adaptive_tb"])
def test_nw_banded_equals_full_493x530(mode):
    # Test_CudapoaNW.cu:446-508: BatchConfig(1024, 2, 128, mode); every banded variant == full band
    nb = V["nw_banded"]
    nodes, read = nb["nodes"], nb["read"]
    outgoing = [[i + 1] for i in range(len(nodes) - 1)] + [[]]
    bm = {"static": 1, "adaptive": 2, "static_tb": 3, "adaptive_tb": 4}[mode]
    cfg_f = O.make_cfg()
    g = O.graph_buffers(nodes, outgoing, cfg_f.max_nodes_per_graph, list(range(len(nodes))))
    nf, agf, arf = O.run_nw(cfg_f, g, read, "full")
    cfg_b = O.make_cfg(1024, 2, 128, bm)
    nb_, agb, arb = O.run_nw(cfg_b, g, read, mode)
    assert nf > 0 and nf == nb_
    assert csv(agf) == csv(agb) and csv(arf) == csv(arb)
    assert nf == 550  # alignment length observed by the survey probe (SURVEY.md Appendix A validation note)


@pytest.mark.parametrize("case", V["topsort"], ids=[c["answer"] for c in V["topsort"]])
def test_topsort_known_answers(case):
    # Test_CudapoaTopSort.cu:48-58
    n = len(case["outgoing"])
    g = O.graph_buffers(None, case["outgoing"], 64)
    sp = np.zeros(64, np.int32)
    pos = np.zeros(64, np.int32)
    O.lib().poa_run_topsort(O.p(sp), O.p(pos), C.c_int32(n), O.p(g["incoming_count"]), O.p(g["outgoing"]), O.p(g["outgoing_count"]))
    assert "-".join(str(int(x)) for x in sp[:n]) == case["answer"]
    assert all(pos[sp[i]] == i for i in range(n))


@pytest.mark.parametrize("idx", range(len(V["add_alignment"])))
def test_add_alignment_known_answers(idx):
    # Test_CudapoaAddAlignment.cu:127-229, harness :233-340
    case = V["add_alignment"][idx]
    mx = 3072
    g = O.graph_buffers(case["nodes"], case["outgoing"], mx)
    na = np.zeros(mx * O.E, np.int32)
    nac = np.zeros(mx, np.uint16)
    w = np.zeros(mx * O.E, np.uint16)
    cov = np.zeros(mx, np.uint16)
    cov[:len(case["coverage"])] = case["coverage"]
    rd = np.frombuffer(case["read"].encode(), np.uint8).copy()
    bw = np.array(case["weights"], np.int8)
    ag = np.array(case["alignment_graph"], np.int32)
    ar = np.array(case["alignment_read"], np.int32)
    nc = C.c_int32(len(case["nodes"]))
    st = O.lib().poa_run_add_alignment(O.p(g["nodes"]), C.byref(nc), O.p(na), O.p(nac), O.p(g["incoming"]),
                                       O.p(g["incoming_count"]), O.p(g["outgoing"]), O.p(g["outgoing_count"]), O.p(w),
                                       C.c_int32(len(ag)), O.p(ag), O.p(rd), O.p(ar), O.p(cov), O.p(bw), C.c_int32(mx))
    assert st == 0
    res = [[int(g["outgoing"][i * O.E + j]) for j in range(g["outgoing_count"][i])] for i in range(nc.value)]
    assert res == case["answer"]


@pytest.mark.parametrize("idx", range(len(V["consensus"])))
def test_consensus_known_answers(idx):
    # Test_CudapoaGenerateConsensus.cu:95-160 with the harness's weight placement [to*50 + from] (:62-73)
    case = V["consensus"][idx]
    mx = 3072
    g = O.graph_buffers(case["nodes"], case["outgoing"], mx, case["sorted"])
    na = np.zeros(mx * O.E, np.int32)
    nac = np.zeros(mx, np.uint16)
    for i, al in enumerate(case["node_alignments"]):
        for j, a in enumerate(al):
            na[i * O.E + j] = a
            nac[i] += 1
    w = np.zeros(mx * O.E, np.uint16)
    for i, outs in enumerate(case["outgoing"]):
        for j, to in enumerate(outs):
            w[to * O.E + i] = case["outgoing_w"][i][j]
    cov = np.zeros(mx, np.uint16)
    cov[:len(case["coverage"])] = case["coverage"]
    cons = np.zeros(2048, np.uint8)
    cvg = np.zeros(2048, np.uint16)
    O.lib().poa_run_consensus(O.p(g["nodes"]), C.c_int32(len(case["nodes"])), O.p(g["graph"]), O.p(g["pos"]),
                              O.p(g["incoming"]), O.p(g["incoming_count"]), O.p(g["outgoing"]), O.p(g["outgoing_count"]),
                              O.p(w), O.p(cons), O.p(cvg), O.p(cov), O.p(na), O.p(nac), C.c_int32(2048))
    n = int(np.argmax(cons == 0))
    assert bytes(cons[:n]).decode() == case["answer"]


def test_band_start_fp32_semantics():
    # cudapoa_nw_banded.cuh:67-78 -- IEEE fp32 product, truncation, clamp to max_column, round down to x4
    L = O.lib()
    g = np.float32(985.0) / np.float32(1351.0)
    for row in (0, 1, 7, 100, 777, 1350):
        d = int(np.float32(row) * g)
        s = max(0, d - 128)
        if 985 < s + 256:
            s = max(0, 985 - 256 + 4)
        s -= s % 4
        assert L.poa_band_start_for_row(row, C.c_float(float(g)), 256, 128, 985) == s


@pytest.mark.parametrize("band_mode", [0, 1, 2, 3, 4])
def test_batch_three_identical_reads(band_mode):
    # Test_CudapoaBatch.cu:155-205: 3 x ('A' x 1023) -> consensus == read, every band mode
    read = "A" * 1023
    cfg = O.make_cfg(1024, 10, 256, band_mode)
    with O.Workspace(cfg) as ws:
        r = ws.process([read, read, read])
        assert r["status"] == 0 and r["consensus"] == read
        assert list(r["coverage"]) == [3] * 1023
        assert ws.overflow_events() == 0


def test_python_binding_graph_shape():
    # test_cudapoa_bindings.py:102-123: ACTGACTG / ACTTACTG / ACTCACTG -> 10 nodes, 11 edges
    cfg = O.make_cfg(1024, 10, 256, 0)
    with O.Workspace(cfg) as ws:
        r = ws.process(["ACTGACTG", "ACTTACTG", "ACTCACTG"])
        assert r["status"] == 0 and r["node_count"] == 10
        g = O.lib().poa_workspace_graph
        g.restype = C.c_void_p
        g.argtypes = [C.c_void_p]
        # count edges through the incoming_edge_count array (5th pointer of poa_graph)
        ptrs = C.cast(g(ws.h), C.POINTER(C.c_void_p))
        inc = np.ctypeslib.as_array(C.cast(ptrs[4], C.POINTER(C.c_uint16)), shape=(10,))
        assert int(inc.sum()) == 11


def test_python_binding_consensus_2pct_substitutions():
    # test_cudapoa_bindings.py:129-152: 100 reads at 2% substitution of a 500-bp reference -> consensus == reference
    import random
    random.seed(2)
    ref = "".join(random.choice("ACGT") for _ in range(500))
    reads = []
    for _ in range(100):
        r = list(ref)
        for i in range(len(r)):
            if random.random() < 0.02:
                r[i] = random.choice([c for c in "ACGT" if c != r[i]])
        reads.append("".join(r))
    cfg = O.make_cfg(1024, 100, 256, 0)
    with O.Workspace(cfg) as ws:
        r = ws.process(reads)
        assert r["status"] == 0 and r["consensus"] == ref


def test_msa_rows_degap_to_inputs():
    # Test_CudapoaGenerateMSA2.cu:119-128 (default build): every de-gapped MSA row equals its input
    import random
    random.seed(7)
    bb = "".join(random.choice("ACGT") for _ in range(50))
    reads = [bb]
    for _ in range(30):
        r = list(bb)
        for _m in range(5):
            r[random.randrange(len(r))] = random.choice("ACGT")
        for _m in range(3):
            r.insert(random.randrange(len(r)), random.choice("ACGT"))
        for _m in range(4):
            del r[random.randrange(len(r))]
        reads.append("".join(r))
    cfg = O.make_cfg(1024, 100, 256, 0, output_mask=2)
    with O.Workspace(cfg) as ws:
        r = ws.process(reads)
        assert r["status"] == 0
        assert len({len(x) for x in r["msa"]}) == 1
        assert [x.replace("-", "") for x in r["msa"]] == reads


def test_msa_failure_status():
    # Test_CudapoaGenerateMSA2.cu:131-164: max_consensus_size too small -> exceeded_maximum_sequence_size
    cfg = O.make_cfg(1024, 100, 256, 0, output_mask=2)
    cfg.max_consensus_size = 40
    with O.Workspace(cfg) as ws:
        r = ws.process(["ACGT" * 12, "ACGT" * 12])
        assert r["status"] == 2


@pytest.mark.parametrize("lane_order", [0, 1])
def test_incremental_topsort_model_equals_kahn(lane_order):
    """The kernel re-sorts incrementally (topsort_kahn_incr_lds); its scalar model must give the order of
    topologicalSortDeviceUtil (cudapoa_topsort.cuh:45-97) after every read, on short-read windows, on windows with
    heavy indels / few reads / N bases, and regardless of the order in which a block's lanes hit the counters."""
    import random
    from genomeworks_amd import synthetic
    rng = random.Random(77)
    windows = [[r.decode() for r in synthetic.generate_window(1000 + w)] for w in range(6)]
    for k in range(30):
        blen = rng.choice([40, 130, 300, 640, 900, 1000])
        reads = rng.choice([2, 3, 8, 17, 32])
        mut, ins, dele = rng.choice([(0, 0, 0), (5, 2, 2), (40, 20, 20), (90, 40, 40), (10, 60, 5), (10, 5, 60)])
        w = [r.decode() for r in synthetic.generate_window(7000 + k, blen, reads, mut, ins, dele)]
        if k % 4 == 0:  # reads that start / end differently: new source and sink nodes
            w = [("GATTACA"[: rng.randrange(8)] + r)[rng.randrange(5):] for r in w]
        windows.append([r for r in w if 0 < len(r) < 1024])
    with O.topsort_model(lane_order) as tm:
        for mode in (1, 2):
            with O.Workspace(O.make_cfg(1024, 32, 256, mode)) as ws:
                for w in windows:
                    ws.process(w)
        st = tm.stats()
    assert st["reads"] > 500 and st["mismatch"] == 0, st
    assert st["empty_blocks"] == 0                       # a block always replays at least the queue head
    assert st["block_nodes"] > 4 * st["real_steps"], st  # and most pops are replayed, not recomputed


def test_msa_rows_by_scatter_over_the_nodes_equal_the_walk():
    """The MSA kernel does not walk every sequence along the edges whose coverage list names it (the reference's rule,
    cudapoa_generate_msa.cuh:56-125) but scatters every node's base into the rows of the sequences on its out-edges
    (generate_msa_rows_wave). The oracle runs a scalar model of that next to the walk on every MSA window; over short and
    long, similar and divergent, single-read and single-base windows, all band modes, the two must never differ."""
    from genomeworks_amd import synthetic
    before = O.Workspace.msa_scatter_mismatches()
    cases = 0
    for mode, band in (("full_band", 256), ("static_band", 256), ("adaptive_band", 256), ("static_band_traceback", 256)):
        cfg = O.make_cfg(512, 16, band, {"full_band": 0, "static_band": 1, "adaptive_band": 2, "static_band_traceback": 3}[mode],
                         output_mask=2)
        with O.Workspace(cfg) as ws:
            for w in range(12):
                reads = [r.decode() for r in synthetic.generate_window(7000 + w, 60 + 35 * w, 3 + w, 4 + 3 * w, 2 + 2 * w, 2 + 2 * w)]
                ws.process(reads)
                cases += 1
            for reads in (["ACGTACGTAC"], ["ACGTACGTAC", "A", "ACGTACGTACGGGT"], ["AAAA", "CCCC", "GGGG", "TTTT"], ["ACGT" * 30, "ACGT" * 29 + "A"]):
                ws.process(reads)
                cases += 1
    assert cases == 64
    assert O.Workspace.msa_scatter_mismatches() == before


@pytest.mark.parametrize("lane_order", [0, 1])
def test_incremental_topsort_with_lds_state_model_equals_kahn(lane_order):
    """The long-read kernel keeps the incremental order's hot state in LDS (topsort_kahn_incr_cnt8: byte counters, a sliding
    window over 1024 positions of the previous order refilled 256 at a time, a 1024-entry queue ring, "in sync at p" as "the
    head is sigma[p]"). Its scalar model must give the order of topologicalSortDeviceUtil after every read -- on short-read
    windows (graphs of 1 400 .. 2 800 nodes: the window slides), on divergent multi-kbp reads (graphs of many thousand
    nodes), with new sources and sinks, in both lane orders of a block's decrements -- and never read outside its window."""
    import random
    from genomeworks_amd import synthetic
    rng = random.Random(78)
    windows = [[r.decode() for r in synthetic.generate_window(1000 + w)] for w in range(4)]
    for k in range(16):
        blen = rng.choice([40, 300, 900, 1000])
        reads = rng.choice([2, 8, 17, 32])
        mut, ins, dele = rng.choice([(0, 0, 0), (5, 2, 2), (40, 20, 20), (90, 40, 40), (10, 60, 5), (10, 5, 60)])
        w = [r.decode() for r in synthetic.generate_window(7100 + k, blen, reads, mut, ins, dele)]
        if k % 4 == 0:  # reads that start / end differently: new source and sink nodes
            w = [("GATTACA"[: rng.randrange(8)] + r)[rng.randrange(5):] for r in w]
        windows.append([r for r in w if 0 < len(r) < 1024])
    long_windows = [[r.decode() for r in synthetic.generate_window(9100 + w, 3000 + 900 * w, 6, 150, 190, 190)] for w in range(2)]
    with O.topsort_cnt8_model(lane_order) as tm:
        for mode in (1, 2):
            with O.Workspace(O.make_cfg(1024, 32, 256, mode)) as ws:
                for w in windows:
                    ws.process(w)
        with O.Workspace(O.make_cfg(8192, 8, 256, 2, storage_factor=4.0, graph_factor=4.0)) as ws:
            for w in long_windows:
                ws.process(w)
        st = tm.stats()
    assert st["reads"] > 400 and st["mismatch"] == 0, st
    assert st["coverage"] == 0 and st["gave_up"] == 0, st
    assert st["refills"] > st["reads"], st                 # the window did slide
    assert st["block_nodes"] > 3 * st["real_steps"], st    # most pops are replayed, not recomputed
    assert st["hbm_steps"] * 2 < st["real_steps"], st      # and most ordinary steps need no HBM round trip


def test_banded_int16_scores_cannot_wrap_inside_the_type_selection_bounds():
    """VERDICT r4 item 9 / ADVICE r3: is there an in-spec window whose banded int16 scores wrap (the kernels would end it with
    generic_error, INTEGRATION.md section 5, where the reference returns a wrapped result)? A search at the edge of the type
    selection -- the deepest graphs that still select int16 (max_nodes_per_graph 4352 at max_sequence_size 1024, scores
    8 / -6 / -8: cudapoa_limits.hpp:34-59), filled by mutually unrelated reads so that every alignment is as bad as alignments
    get -- finds none, and there is a bound behind it: a banded row whose band starts past column 0 takes min / 2 = -16384 as
    its left boundary AFRESH in every row (cudapoa_nw_banded.cuh:158-175), so no cell of the band lies below
    -16384 + min(mismatch, gap) + (band_width - 1) * gap (= -18 430 at band 256, -28 670 at the widest adaptive band of 1536
    columns), and the rows whose band starts at column 0 are the first band_width / 2 / gradient rows, whose cells lie above
    (rows + columns) * gap. A wrap needs scores outside the range the API documents (|gap| >= 11 at 1536 columns)."""
    import random
    cfg = O.make_cfg(1024, 32, 256, 1)
    cfg.max_nodes_per_graph = 4352
    cfg.matrix_sequence_dimension = 264
    cfg.max_banded_pred_distance = 512
    O.lib().poa_cfg_select_types(cfg)
    assert cfg.score32 == 0  # still the int16 kernels
    rng = random.Random(5)
    with O.Workspace(cfg) as ws:
        for n, length in ((12, 1000), (8, 1024), (14, 900)):
            reads = ["".join(rng.choice("ACGT") for _ in range(length)) for _ in range(n)]
            ref = ws.process(reads)
            assert ref["status"] == 0 and ref["node_count"] > 2500
        assert ws.overflow_events() == 0
    acfg = O.make_cfg(1024, 32, 256, 2)   # adaptive band: widens for these gradients
    acfg.max_nodes_per_graph = 4352
    O.lib().poa_cfg_select_types(acfg)
    with O.Workspace(acfg) as ws:
        reads = ["".join(rng.choice("ACGT") for _ in range(1000)) for _ in range(10)]
        ws.process(reads)
        assert ws.overflow_events() == 0
