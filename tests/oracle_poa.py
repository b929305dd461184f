"""ctypes view of oracle/build/libpoa_oracle.so -- TEST INFRASTRUCTURE ONLY (never imported by genomeworks_amd)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

E = 50  # CUDAPOA_MAX_NODE_EDGES / ALIGNMENTS


class PoaCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "max_sequence_size", "max_consensus_size", "max_nodes_per_graph", "matrix_sequence_dimension",
        "alignment_band_width", "max_sequences_per_poa", "band_mode", "max_banded_pred_distance",
        "gap_score", "mismatch_score", "match_score", "score32", "trace16", "output_mask", "spoa_accurate")]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "build", "libpoa_oracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "build/libpoa_oracle.so"], check=True)
        L = C.CDLL(path)
        L.poa_band_start_for_row.restype = C.c_int32
        L.poa_band_start_for_row.argtypes = [C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_int32]
        L.poa_workspace_create.restype = C.c_void_p
        L.poa_workspace_create.argtypes = [C.POINTER(PoaCfg)]
        L.poa_workspace_destroy.argtypes = [C.c_void_p]
        L.poa_workspace_overflow_events.restype = C.c_int64
        L.poa_oracle_msa_scatter_mismatches.restype = C.c_int64
        L.poa_oracle_msa_scatter_mismatches.argtypes = []
        L.poa_workspace_overflow_events.argtypes = [C.c_void_p]
        L.poa_process_window.restype = C.c_int32
        L.poa_process_window.argtypes = [C.c_void_p] + [C.c_void_p] * 3 + [C.c_int32, C.c_size_t] + [C.c_void_p] * 4
        L.poa_cfg_init.argtypes = [C.POINTER(PoaCfg), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                   C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.poa_cfg_select_types.argtypes = [C.POINTER(PoaCfg)]
        L.poa_topsort_model_enable.argtypes = [C.c_int, C.c_int]
        L.poa_topsort_model_stats.argtypes = [C.c_void_p]
        L.poa_topsort_cnt8_model_enable.argtypes = [C.c_int, C.c_int]
        L.poa_topsort_cnt8_model_stats.argtypes = [C.c_void_p]
        L.poa_run_nw_full.restype = C.c_int32
        L.poa_run_nw_banded.restype = C.c_int32
        L.poa_run_add_alignment.restype = C.c_int32
        _LIB = L
    return _LIB


def make_cfg(max_seq=1024, max_seqs=100, band_width=256, band_mode=0, storage_factor=2.0, graph_factor=3.0,
             max_pred=0, gap=-8, mismatch=-6, match=8, output_mask=1, spoa_accurate=0):
    c = PoaCfg()
    lib().poa_cfg_init(C.byref(c), max_seq, max_seqs, band_width, band_mode, storage_factor, graph_factor, max_pred,
                       gap, mismatch, match, output_mask)
    c.spoa_accurate = spoa_accurate
    return c


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def graph_buffers(nodes, outgoing, max_nodes, sorted_graph=None):
    """BasicGraph::get_edges / SortedGraph (cudapoa/tests/basic_graph.hpp:74-90, sorted_graph.hpp:52-68)."""
    n = len(outgoing)
    g = dict(
        nodes=np.zeros(max_nodes, np.uint8), incoming=np.zeros(max_nodes * E, np.int32),
        incoming_count=np.zeros(max_nodes, np.uint16), outgoing=np.zeros(max_nodes * E, np.int32),
        outgoing_count=np.zeros(max_nodes, np.uint16), graph=np.zeros(max_nodes, np.int32),
        pos=np.zeros(max_nodes, np.int32), count=n)
    if nodes is not None:
        g["nodes"][:len(nodes)] = np.frombuffer(nodes.encode() if isinstance(nodes, str) else bytes(nodes), np.uint8)
    for i, outs in enumerate(outgoing):
        g["outgoing_count"][i] = len(outs)
        for j, o in enumerate(outs):
            k = g["incoming_count"][o]
            g["incoming_count"][o] = k + 1
            g["incoming"][o * E + k] = i
            g["outgoing"][i * E + j] = o
    if sorted_graph is not None:
        g["graph"][:n] = sorted_graph
        for pos, nid in enumerate(sorted_graph):
            g["pos"][nid] = pos
    return g


def run_nw(cfg, g, read, mode="full"):
    """mode: full | static | adaptive | static_tb | adaptive_tb. Returns (len, alignment_graph, alignment_read)."""
    L = lib()
    mx = cfg.max_nodes_per_graph
    rd = np.zeros(cfg.max_sequence_size + 2048, np.uint8)
    rb = read.encode() if isinstance(read, str) else bytes(read)
    rd[:len(rb)] = np.frombuffer(rb, np.uint8)
    ag = np.zeros(2 * mx + 16, np.int32)
    ar = np.zeros(2 * mx + 16, np.int32)
    common = [p(g["nodes"]), p(g["graph"]), p(g["pos"]), C.c_int32(g["count"]), p(g["incoming_count"]), p(g["incoming"]),
              p(g["outgoing_count"]), p(rd), C.c_int32(len(rb)), p(ag), p(ar)]
    if mode == "full":
        n = L.poa_run_nw_full(C.byref(cfg), *common)
    else:
        adaptive = 1 if mode.startswith("adaptive") else 0
        tb = 1 if mode.endswith("_tb") else 0
        n = L.poa_run_nw_banded(C.byref(cfg), C.c_int32(adaptive), C.c_int32(tb), *common)
    return n, ag[:max(n, 0)].copy(), ar[:max(n, 0)].copy()


def pack_window(reads, weights=None):
    """Host packing of one window: cudapoa_batch.cuh:516-537 (each read padded to a multiple of 4 bytes)."""
    lens = np.array([len(r) for r in reads], np.int32)
    tot = int(sum((l + 3) & ~3 for l in lens))
    seqs = np.zeros(tot + 2048, np.uint8)
    wts = np.zeros(tot + 2048, np.int8)
    off = 0
    for i, r in enumerate(reads):
        rb = r.encode() if isinstance(r, str) else bytes(r)
        seqs[off:off + len(rb)] = np.frombuffer(rb, np.uint8)
        wts[off:off + len(rb)] = 1 if weights is None or weights[i] is None else np.asarray(weights[i], np.int8)
        off += (len(rb) + 3) & ~3
    return seqs, wts, lens, tot


class Workspace:
    def __init__(self, cfg):
        self.cfg = cfg
        self.h = lib().poa_workspace_create(C.byref(cfg))

    def close(self):
        if self.h:
            lib().poa_workspace_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @staticmethod
    def msa_scatter_mismatches():
        """Windows (process-wide) whose MSA rows by scatter over the nodes -- the kernel's formulation -- differed from the
        reference's per-sequence walk."""
        return lib().poa_oracle_msa_scatter_mismatches()

    def overflow_events(self):
        return lib().poa_workspace_overflow_events(self.h)

    def process(self, reads, weights=None):
        """Returns dict(status, consensus(str, un-reversed like the host API), coverage, msa, node_count, cells)."""
        cfg = self.cfg
        # the output buffers below are sized by the config: the batch API rejects what does not fit before the kernels run
        assert len(reads) <= cfg.max_sequences_per_poa and all(len(r) <= cfg.max_sequence_size for r in reads)
        seqs, wts, lens, tot = pack_window(reads, weights)
        cons = np.zeros(cfg.max_consensus_size, np.uint8)
        cov = np.zeros(cfg.max_consensus_size, np.uint16)
        msa = np.zeros(cfg.max_sequences_per_poa * cfg.max_consensus_size, np.uint8) if cfg.output_mask & 2 else None
        cells = C.c_int64(0)
        st = lib().poa_process_window(self.h, p(seqs), p(wts), p(lens), len(reads), tot + 2048, p(cons), p(cov),
                                      p(msa) if msa is not None else None, C.byref(cells))
        out = dict(status=st, node_count=int(lens[0]), cells=cells.value, raw_consensus=cons, raw_coverage=cov)
        if st == 0 and not (cfg.output_mask & 2):
            n = int(np.argmax(cons == 0))
            # host un-reversal: cudapoa_batch.cuh:246-252
            out["consensus"] = bytes(cons[:n][::-1]).decode()
            out["coverage"] = cov[:n][::-1].copy()
        if st == 0 and msa is not None:
            rows = []
            for s in range(len(reads)):
                row = msa[s * cfg.max_consensus_size:(s + 1) * cfg.max_consensus_size]
                rows.append(bytes(row[:int(np.argmax(row == 0))]).decode())
            out["msa"] = rows
        return out


TOPSORT_MODEL_STATS = ("reads", "nodes", "real_steps", "blocks", "block_nodes", "sync_checks", "sync_hits", "mismatch",
                       "empty_blocks", "wide_nodes", "wide_steps")


class topsort_model:
    """Context manager: run the scalar model of the kernel's incremental Kahn order (oracle/topsort_incr_model.inc)
    next to the plain topologicalSortDeviceUtil restatement inside poa_process_window and count disagreements."""

    def __init__(self, lane_order=0):
        self.lane_order = lane_order

    def __enter__(self):
        lib().poa_topsort_model_enable(1, self.lane_order)
        return self

    def stats(self):
        st = (C.c_int64 * len(TOPSORT_MODEL_STATS))()
        lib().poa_topsort_model_stats(st)
        return dict(zip(TOPSORT_MODEL_STATS, list(st)))

    def __exit__(self, *a):
        lib().poa_topsort_model_enable(0, 0)


TOPSORT_CNT8_MODEL_STATS = ("reads", "nodes", "real_steps", "blocks", "block_nodes", "mismatch", "gave_up", "coverage", "refills",
                            "hbm_steps")


class topsort_cnt8_model:
    """Context manager: run the scalar model of the long-read kernel's incremental Kahn order with its state in LDS
    (oracle/topsort_incr_cnt8_model.inc: byte counters, sliding window over the previous order, queue ring) next to the plain
    topologicalSortDeviceUtil restatement inside poa_process_window and count disagreements."""

    def __init__(self, lane_order=0):
        self.lane_order = lane_order

    def __enter__(self):
        lib().poa_topsort_cnt8_model_enable(1, self.lane_order)
        return self

    def stats(self):
        st = (C.c_int64 * len(TOPSORT_CNT8_MODEL_STATS))()
        lib().poa_topsort_cnt8_model_stats(st)
        return dict(zip(TOPSORT_CNT8_MODEL_STATS, list(st)))

    def __exit__(self, *a):
        lib().poa_topsort_cnt8_model_enable(0, 0)
