"""GPU unit tests of the individual POA device functions through the gwhip_poa_test_* hooks, fed with the
reference's own inline known-answer vectors (tests/golden/cudapoa_vectors.json)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_poa as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "cudapoa_vectors.json")) as f:
    V = json.load(f)
E = 50


class HookGraph(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("graph", C.c_void_p), ("node_id_to_pos", C.c_void_p), ("graph_count", C.c_int32),
                ("incoming_edge_count", C.c_void_p), ("incoming_edges", C.c_void_p), ("outgoing_edge_count", C.c_void_p),
                ("outgoing_edges", C.c_void_p)]


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def lib():
    from genomeworks_amd import _native
    L = _native.gwhip()
    L.gwhip_poa_test_nw_scratch_bytes.restype = C.c_size_t
    L.gwhip_poa_test_nw_scratch_bytes.argtypes = [C.POINTER(_native.PoaConfig)]
    L.gwhip_poa_test_nw.argtypes = [C.POINTER(_native.PoaConfig), C.POINTER(HookGraph), C.c_void_p, C.c_int32] + [C.c_void_p] * 5
    L.gwhip_poa_test_topsort.argtypes = [C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 5
    L.gwhip_poa_test_add_alignment.argtypes = [C.c_void_p] * 9 + [C.c_int32] + [C.c_void_p] * 5 + [C.c_int32, C.c_void_p, C.c_void_p]
    L.gwhip_poa_test_consensus.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 14 + [C.c_int32, C.c_void_p]
    return L


def device_cfg(ocfg):
    from genomeworks_amd import _native
    return _native.PoaConfig(ocfg.max_sequence_size, ocfg.max_consensus_size, ocfg.max_nodes_per_graph,
                             ocfg.matrix_sequence_dimension, ocfg.alignment_band_width, ocfg.max_sequences_per_poa,
                             ocfg.band_mode, ocfg.max_banded_pred_distance, ocfg.gap_score, ocfg.mismatch_score,
                             ocfg.match_score, 1, 0, 1, ocfg.trace16, 0)


def run_nw_gpu(ocfg, g, read):
    import torch
    L = lib()
    cfg = device_cfg(ocfg)
    mx = ocfg.max_nodes_per_graph
    t = {k: dev(g[k]) for k in ("nodes", "graph", "pos", "incoming_count", "incoming", "outgoing_count", "outgoing")}
    tg = HookGraph(t["nodes"].data_ptr(), t["graph"].data_ptr(), t["pos"].data_ptr(), g["count"],
                   t["incoming_count"].data_ptr(), t["incoming"].data_ptr(), t["outgoing_count"].data_ptr(),
                   t["outgoing"].data_ptr())
    rb = np.zeros(ocfg.max_sequence_size + 4096, np.uint8)
    r = np.frombuffer(read.encode(), np.uint8)
    rb[:len(r)] = r
    d_read = dev(rb)
    scratch = torch.zeros(L.gwhip_poa_test_nw_scratch_bytes(C.byref(cfg)) + 256, dtype=torch.uint8, device="cuda")
    ag = torch.zeros(2 * mx + 16, dtype=torch.int32, device="cuda")
    ar = torch.zeros(2 * mx + 16, dtype=torch.int32, device="cuda")
    n = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = L.gwhip_poa_test_nw(C.byref(cfg), C.byref(tg), d_read.data_ptr(), len(r), scratch.data_ptr(), ag.data_ptr(),
                             ar.data_ptr(), n.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    k = int(n.item())
    return k, ag[:max(k, 0)].cpu().numpy(), ar[:max(k, 0)].cpu().numpy()


def csv(a):
    return ",".join(str(int(x)) for x in a)


@pytest.mark.parametrize("mode", ["full", "static", "adaptive", "static_tb", "adaptive_tb"])
@pytest.mark.parametrize("case", V["nw"], ids=[c["name"] for c in V["nw"]])
def test_nw_known_answers(case, mode):
    # Test_CudapoaNW.cu:100-187 (full band) and the same graphs through every banded variant
    bm = {"full": 0, "static": 1, "adaptive": 2, "static_tb": 3, "adaptive_tb": 4}[mode]
    ocfg = O.make_cfg() if mode == "full" else O.make_cfg(1024, 2, 128, bm)
    g = O.graph_buffers(case["nodes"], case["outgoing"], ocfg.max_nodes_per_graph, case["sorted"])
    n, ag, ar = run_nw_gpu(ocfg, g, case["read"])
    assert (csv(ag), csv(ar)) == (case["graph_ans"], case["read_ans"])


@pytest.mark.parametrize("mode", ["static", "adaptive", "static_tb", "adaptive_tb"])
def test_nw_banded_equals_full_493x530(mode):
    # Test_CudapoaNW.cu:446-508
    nb = V["nw_banded"]
    nodes, read = nb["nodes"], nb["read"]
    outgoing = [[i + 1] for i in range(len(nodes) - 1)] + [[]]
    bm = {"static": 1, "adaptive": 2, "static_tb": 3, "adaptive_tb": 4}[mode]
    cf = O.make_cfg()
    g = O.graph_buffers(nodes, outgoing, cf.max_nodes_per_graph, list(range(len(nodes))))
    nf, agf, arf = run_nw_gpu(cf, g, read)
    cb = O.make_cfg(1024, 2, 128, bm)
    nbn, agb, arb = run_nw_gpu(cb, g, read)
    assert nf == nbn == 550
    assert csv(agf) == csv(agb) and csv(arf) == csv(arb)
    no, ago, aro = O.run_nw(cb, g, read, mode)
    assert csv(ago) == csv(agb) and csv(aro) == csv(arb)


@pytest.mark.parametrize("case", V["topsort"], ids=[c["answer"] for c in V["topsort"]])
def test_topsort_known_answers(case):
    # Test_CudapoaTopSort.cu:48-58
    import torch
    n = len(case["outgoing"])
    g = O.graph_buffers(None, case["outgoing"], 64)
    sp = torch.zeros(64, dtype=torch.int32, device="cuda")
    pos = torch.zeros(64, dtype=torch.int32, device="cuda")
    loc = torch.zeros(64, dtype=torch.int16, device="cuda")
    ic, oe, oc = dev(g["incoming_count"]), dev(g["outgoing"]), dev(g["outgoing_count"])
    assert lib().gwhip_poa_test_topsort(sp.data_ptr(), pos.data_ptr(), n, ic.data_ptr(), oe.data_ptr(), oc.data_ptr(),
                                        loc.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert "-".join(str(int(x)) for x in sp[:n].cpu()) == case["answer"]


@pytest.mark.parametrize("idx", range(len(V["add_alignment"])))
def test_add_alignment_known_answers(idx):
    # Test_CudapoaAddAlignment.cu:127-229
    import torch
    case = V["add_alignment"][idx]
    mx = 3072
    g = O.graph_buffers(case["nodes"], case["outgoing"], mx)
    t = {k: dev(g[k]) for k in ("nodes", "incoming_count", "incoming", "outgoing_count", "outgoing")}
    na = torch.zeros(mx * E, dtype=torch.int32, device="cuda")
    nac = torch.zeros(mx, dtype=torch.int16, device="cuda")
    w = torch.zeros(mx * E, dtype=torch.int16, device="cuda")
    cov = np.zeros(mx, np.uint16)
    cov[:len(case["coverage"])] = case["coverage"]
    d_cov = dev(cov)
    rd = dev(np.frombuffer(case["read"].encode(), np.uint8).copy())
    bw = dev(np.array(case["weights"], np.int8))
    ag = dev(np.array(case["alignment_graph"], np.int32))
    ar = dev(np.array(case["alignment_read"], np.int32))
    nc = dev(np.array([len(case["nodes"])], np.int32))
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = lib().gwhip_poa_test_add_alignment(t["nodes"].data_ptr(), nc.data_ptr(), na.data_ptr(), nac.data_ptr(),
                                            t["incoming"].data_ptr(), t["incoming_count"].data_ptr(),
                                            t["outgoing"].data_ptr(), t["outgoing_count"].data_ptr(), w.data_ptr(),
                                            len(case["alignment_graph"]), ag.data_ptr(), rd.data_ptr(), ar.data_ptr(),
                                            d_cov.data_ptr(), bw.data_ptr(), mx, st.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert int(st.item()) == 0
    n = int(nc.item())
    oc = t["outgoing_count"].cpu().numpy()
    oe = t["outgoing"].cpu().numpy()
    assert [[int(oe[i * E + j]) for j in range(oc[i])] for i in range(n)] == case["answer"]


@pytest.mark.parametrize("idx", range(len(V["consensus"])))
def test_consensus_known_answers(idx):
    # Test_CudapoaGenerateConsensus.cu:95-160 with the harness's weight placement [to*50 + from] (:62-73)
    import torch
    case = V["consensus"][idx]
    mx = 3072
    g = O.graph_buffers(case["nodes"], case["outgoing"], mx, case["sorted"])
    na = np.zeros(mx * E, np.int32)
    nac = np.zeros(mx, np.uint16)
    for i, al in enumerate(case["node_alignments"]):
        for j, a in enumerate(al):
            na[i * E + j] = a
            nac[i] += 1
    w = np.zeros(mx * E, np.uint16)
    for i, outs in enumerate(case["outgoing"]):
        for j, to in enumerate(outs):
            w[to * E + i] = case["outgoing_w"][i][j]
    cov = np.zeros(mx, np.uint16)
    cov[:len(case["coverage"])] = case["coverage"]
    t = {k: dev(g[k]) for k in ("nodes", "graph", "pos", "incoming_count", "incoming", "outgoing_count", "outgoing")}
    d_na, d_nac, d_w, d_cov = dev(na), dev(nac), dev(w), dev(cov)
    preds = torch.zeros(mx, dtype=torch.int32, device="cuda")
    scores = torch.zeros(mx + 2, dtype=torch.int32, device="cuda")
    cons = torch.zeros(2048, dtype=torch.uint8, device="cuda")
    cvg = torch.zeros(2048, dtype=torch.int16, device="cuda")
    rc = lib().gwhip_poa_test_consensus(t["nodes"].data_ptr(), len(case["nodes"]), t["graph"].data_ptr(), t["pos"].data_ptr(),
                                        t["incoming"].data_ptr(), t["incoming_count"].data_ptr(), t["outgoing"].data_ptr(),
                                        t["outgoing_count"].data_ptr(), d_w.data_ptr(), preds.data_ptr(), scores.data_ptr(),
                                        cons.data_ptr(), cvg.data_ptr(), d_cov.data_ptr(), d_na.data_ptr(), d_nac.data_ptr(),
                                        2048, None)
    assert rc == 0
    torch.cuda.synchronize()
    c = cons.cpu().numpy()
    n = int(np.argmax(c == 0))
    assert bytes(c[:n]).decode() == case["answer"]


def test_scores_that_leave_int16_end_as_an_error_not_as_a_result():
    """The kernels equal the reference only while no int16 store wraps (with a wrap the reference's result depends on its
    relaxation order). The precondition is enforced: the 493-node case through the static band with a match score that drives
    the scores past 32767 returns the kernel's "score wrapped" code (StatusType::generic_error at the batch level), while the
    same call with the default scores still gives the known answer."""
    nb = V["nw_banded"]
    nodes, read = nb["nodes"], nb["read"]
    outgoing = [[i + 1] for i in range(len(nodes) - 1)] + [[]]
    cb = O.make_cfg(1024, 2, 128, 1)
    g = O.graph_buffers(nodes, outgoing, cb.max_nodes_per_graph, list(range(len(nodes))))
    n_ok, _, _ = run_nw_gpu(cb, g, read)
    assert n_ok == 550
    hot = O.make_cfg(1024, 2, 128, 1, match=120)  # ~490 matches x 120 > 32767; the hook keeps the int16 instantiation
    hot.score32 = 0
    n_bad, _, _ = run_nw_gpu(hot, g, read)
    assert n_bad == -5  # kNwScoreWrapped (poa_layout.h)
