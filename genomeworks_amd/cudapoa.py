"""Python mirror of pygenomeworks' genomeworks.cudapoa (cudapoa.pyx:69-334) over the object-level C API.

Same class, method and argument names; status codes are the integers of cudapoa::StatusType. There is no CPU
path: constructing a batch without the native libraries or without a GPU raises."""
import ctypes as C

import numpy as np

from . import _native
from .cuda import CudaStream

# cudapoa::StatusType (cudapoa.hpp:34-49)
success = 0
exceeded_maximum_poas = 1
exceeded_maximum_sequence_size = 2
exceeded_maximum_sequences_per_poa = 3
node_count_exceeded_maximum_graph_size = 4
edge_count_exceeded_maximum_graph_size = 5
exceeded_adaptive_banded_matrix_size = 6
exceeded_maximum_predecessor_distance = 7
loop_count_exceeded_upper_bound = 8
output_type_unavailable = 9
zero_weighted_poa_sequence = 10
empty_poa_group = 11
generic_error = 12

_STATUS_NAMES = {
    0: "success", 1: "exceeded_maximum_poas", 2: "exceeded_maximum_sequence_size",
    3: "exceeded_maximum_sequences_per_poa", 4: "node_count_exceeded_maximum_graph_size",
    5: "edge_count_exceeded_maximum_graph_size", 6: "exceeded_adaptive_banded_matrix_size",
    7: "exceeded_maximum_predecessor_distance", 8: "loop_count_exceeded_upper_bound",
    9: "output_type_unavailable", 10: "zero_weighted_poa_sequence", 11: "empty_poa_group", 12: "generic_error"}

_BAND_MODES = {"full_band": 0, "static_band": 1, "adaptive_band": 2, "static_band_traceback": 3,
               "adaptive_band_traceback": 4}


def status_to_str(status):
    """Convert status to their string representations (cudapoa.pyx:33-66)."""
    if status not in _STATUS_NAMES:
        raise RuntimeError("Unknown error status : " + str(status))
    return _STATUS_NAMES[status]


def _bind(L):
    if getattr(L, "_gw_poa_bound", False):
        return L
    vp, i32 = C.c_void_p, C.c_int32
    L.gw_poa_batch_config_full.argtypes = [C.POINTER(_native.PoaBatchConfig)] + [i32] * 8
    L.gw_poa_batch_config_default.argtypes = [C.POINTER(_native.PoaBatchConfig), i32, i32, i32, i32, C.c_float,
                                              C.c_float, i32]
    L.gw_poa_create_batch.restype = vp
    L.gw_poa_create_batch.argtypes = [i32, vp, C.c_int64, C.c_int8, C.POINTER(_native.PoaBatchConfig), C.c_int16,
                                      C.c_int16, C.c_int16]
    L.gw_poa_destroy_batch.argtypes = [vp]
    L.gw_poa_add_poa_group.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(vp), C.POINTER(i32), C.POINTER(i32)]
    for name in ("gw_poa_get_total_poas", "gw_poa_generate_poa", "gw_poa_batch_id", "gw_poa_reset", "gw_poa_max_poas",
                 "gw_poa_relaunch"):
        getattr(L, name).argtypes = [vp]
    L.gw_poa_get_consensus.argtypes = [vp, C.POINTER(i32)]
    L.gw_poa_get_consensus_in_place.argtypes = [vp, C.POINTER(i32)]
    L.gw_poa_consensus_str.restype = C.POINTER(C.c_char)
    L.gw_poa_consensus_str.argtypes = [vp, i32, C.POINTER(i32)]
    L.gw_poa_consensus_coverage.restype = C.POINTER(C.c_uint16)
    L.gw_poa_consensus_coverage.argtypes = [vp, i32, C.POINTER(i32)]
    L.gw_poa_output_status.argtypes = [vp, i32]
    L.gw_poa_get_msa.argtypes = [vp, C.POINTER(i32)]
    L.gw_poa_msa_rows.argtypes = [vp, i32]
    L.gw_poa_msa_row.restype = C.POINTER(C.c_char)
    L.gw_poa_msa_row.argtypes = [vp, i32, i32, C.POINTER(i32)]
    L.gw_poa_get_graphs.argtypes = [vp, C.POINTER(i32)]
    L.gw_poa_graph_num_nodes.argtypes = [vp, i32]
    L.gw_poa_graph_num_edges.argtypes = [vp, i32]
    L.gw_poa_graph_copy.argtypes = [vp, i32, vp, vp, vp, vp]
    L.gw_poa_total_cells.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.gw_poa_relaunch_timed.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.gw_poa_profile_phases.argtypes = [vp, C.POINTER(C.c_double)]
    L.gw_poa_profile_phases_per_window.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int32]
    L.gw_poa_profile_phases_per_window.restype = C.c_int32
    L._gw_poa_bound = True
    return L


class CudaPoaBatch:
    """Python API for GPU-accelerated partial order alignment (pygenomeworks CudaPoaBatch)."""

    def __init__(self, max_sequences_per_poa, max_sequence_size, max_gpu_mem, output_type="consensus",
                 band_mode="adaptive_band", device_id=0, stream=None, gap_score=-8, mismatch_score=-6, match_score=8,
                 alignment_band_width=256, max_consensus_size=None, max_nodes_per_graph=None,
                 matrix_sequence_dimension=None, max_banded_pred_distance=None, *args, **kwargs):
        # unknown keyword arguments are swallowed, as in cudapoa.pyx:87-88 (its tests rely on it)
        self._L = _bind(_native.host())
        if stream is not None and not isinstance(stream, CudaStream):
            raise RuntimeError("Type for stream option must be CudaStream")
        self.stream = stream
        if output_type == "consensus":
            output_mask = 1
        elif output_type == "msa":
            output_mask = 2
        else:
            raise RuntimeError("Unknown output_type provided. Must be consensus/msa.")
        if band_mode not in _BAND_MODES:
            raise RuntimeError("Unknown band_mode provided. Must be full_band/static_band/adaptive_band.")
        mx_consensus = 2 * max_sequence_size if max_consensus_size is None else max_consensus_size
        # defaults of cudapoa.pyx:146-160 (4x graph length for the banded modes)
        if band_mode == "full_band":
            nodes = 3 * max_sequence_size if max_nodes_per_graph is None else max_nodes_per_graph
            msd = max_sequence_size if matrix_sequence_dimension is None else matrix_sequence_dimension
        elif band_mode in ("static_band", "static_band_traceback"):
            nodes = 4 * max_sequence_size if max_nodes_per_graph is None else max_nodes_per_graph
            msd = (alignment_band_width + 8) if matrix_sequence_dimension is None else matrix_sequence_dimension
        else:
            nodes = 4 * max_sequence_size if max_nodes_per_graph is None else max_nodes_per_graph
            msd = 2 * (alignment_band_width + 8) if matrix_sequence_dimension is None else matrix_sequence_dimension
        # cudapoa.pyx passes an uninitialised mx_pred_dist; we use BatchConfig's own default (2 x band width)
        pred = 2 * ((alignment_band_width + 127) // 128 * 128) if max_banded_pred_distance is None else max_banded_pred_distance
        cfg = _native.PoaBatchConfig()
        if self._L.gw_poa_batch_config_full(C.byref(cfg), max_sequence_size, mx_consensus, nodes, alignment_band_width,
                                            max_sequences_per_poa, msd, _BAND_MODES[band_mode], pred) != 0:
            raise ValueError(self._L.gw_last_error().decode())
        self.batch_size = cfg
        self._h = self._L.gw_poa_create_batch(device_id, stream.stream if stream is not None else None,
                                              int(max_gpu_mem), output_mask, C.byref(cfg), gap_score, mismatch_score,
                                              match_score)
        if not self._h:
            raise RuntimeError(self._L.gw_last_error().decode())

    @classmethod
    def from_batch_config(cls, max_sequence_size, max_sequences_per_poa, alignment_band_width, band_mode, max_gpu_mem,
                          output_type="consensus", device_id=0, stream=None, gap_score=-8, mismatch_score=-6, match_score=8,
                          adaptive_storage_factor=2.0, graph_length_factor=3.0, max_banded_pred_distance=0):
        """A batch sized by the first BatchConfig constructor of batch.hpp -- BatchConfig(max_seq_sz, max_seq_per_poa,
        band_width, banding, adaptive_storage_factor, graph_length_factor, max_pred_dist), which derives the consensus, graph
        and matrix dimensions and, unlike the all-explicit constructor that __init__ (and cudapoa.pyx) uses, accepts a band
        wider than the longest sequence. What C++ callers of the reference write."""
        if output_type not in ("consensus", "msa"):
            raise RuntimeError("Unknown output_type provided. Must be consensus/msa.")
        if band_mode not in _BAND_MODES:
            raise RuntimeError("Unknown band_mode provided. Must be full_band/static_band/adaptive_band.")
        if stream is not None and not isinstance(stream, CudaStream):
            raise RuntimeError("Type for stream option must be CudaStream")
        self = cls.__new__(cls)
        self._L = _bind(_native.host())
        self.stream = stream
        cfg = _native.PoaBatchConfig()
        if self._L.gw_poa_batch_config_default(C.byref(cfg), max_sequence_size, max_sequences_per_poa, alignment_band_width,
                                               _BAND_MODES[band_mode], adaptive_storage_factor, graph_length_factor,
                                               max_banded_pred_distance) != 0:
            raise ValueError(self._L.gw_last_error().decode())
        self.batch_size = cfg
        self._h = self._L.gw_poa_create_batch(device_id, stream.stream if stream is not None else None, int(max_gpu_mem),
                                              1 if output_type == "consensus" else 2, C.byref(cfg), gap_score, mismatch_score, match_score)
        if not self._h:
            raise RuntimeError(self._L.gw_last_error().decode())
        return self

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.gw_poa_destroy_batch(self._h)
                self._h = None
        except Exception:
            pass

    def add_poa_group(self, poa, weights=None):
        """Add one POA group (list of sequences). Returns (status, per-sequence statuses)."""
        if not isinstance(poa, list):
            poa = [poa]
        if len(poa) < 1:
            raise RuntimeError("At least one sequence must be present in POA group")
        n = len(poa)
        raw = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in poa]
        seqs = (C.c_char_p * n)(*raw)
        lens = (C.c_int32 * n)(*[len(b) for b in raw])
        st = (C.c_int32 * n)()
        wptr, keep = None, []
        if weights is not None:
            arr = (C.c_void_p * n)()
            for i, w in enumerate(weights):
                if w is None:
                    arr[i] = None
                else:
                    a = np.ascontiguousarray(w, dtype=np.int8)
                    keep.append(a)
                    arr[i] = a.ctypes.data
            wptr = arr
        status = self._L.gw_poa_add_poa_group(self._h, n, seqs, wptr, lens, st)
        if status < 0:
            raise RuntimeError(self._L.gw_last_error().decode())
        return (status, list(st) if status not in (exceeded_maximum_poas,) else [])

    @property
    def total_poas(self):
        return self._L.gw_poa_get_total_poas(self._h)

    @property
    def batch_id(self):
        return self._L.gw_poa_batch_id(self._h)

    @property
    def max_poas(self):
        return self._L.gw_poa_max_poas(self._h)

    def generate_poa(self):
        """Run asynchronous partial order alignment on all POA groups in batch."""
        if self._L.gw_poa_generate_poa(self._h) != 0:
            raise RuntimeError(self._L.gw_last_error().decode())

    def relaunch(self):
        """Benchmark helper: run the kernels again on the inputs already resident in HBM."""
        if self._L.gw_poa_relaunch(self._h) != 0:
            raise RuntimeError(self._L.gw_last_error().decode())

    def relaunch_timed(self):
        """Benchmark helper: relaunch and return (graph_build_ms, output_ms) from HIP events on the batch stream."""
        a, b = C.c_float(0), C.c_float(0)
        if self._L.gw_poa_relaunch_timed(self._h, C.byref(a), C.byref(b)) != 0:
            raise RuntimeError(self._L.gw_last_error().decode())
        return a.value, b.value

    def profile_phases(self):
        """Profiling aid: mean device ticks per window spent in each graph-build phase (one extra launch)."""
        out = (C.c_double * 6)()
        if self._L.gw_poa_profile_phases(self._h, out) != 0:
            raise RuntimeError(self._L.gw_last_error().decode())
        names = ("row_table", "nw_forward", "sink_traceback", "graph_merge", "topsort", "other")
        return dict(zip(names, [float(x) for x in out]))

    def profile_phases_per_window(self):
        """Profiling aid: the six phase counters of every window of the batch (one extra launch), a list of dicts."""
        n = self.total_poas
        out = (C.c_uint64 * (6 * max(n, 1)))()
        got = self._L.gw_poa_profile_phases_per_window(self._h, out, n)
        if got < 0:
            raise RuntimeError(self._L.gw_last_error().decode())
        names = ("row_table", "nw_forward", "sink_traceback", "graph_merge", "topsort", "other")
        return [dict(zip(names, [int(out[6 * w + k]) for k in range(6)])) for w in range(got)]

    def get_consensus_native(self, in_place=False):
        """Batch::get_consensus inside the library -- fresh result vectors through the public call, as the reference's
        benchmark does (single_batch.hpp:86-93) -- without marshalling the strings to Python. in_place=True: the library's
        extension that reuses the previous call's storage. Returns window count."""
        n = C.c_int32(0)
        err = (self._L.gw_poa_get_consensus_in_place if in_place else self._L.gw_poa_get_consensus)(self._h, C.byref(n))
        if err != 0:
            raise RuntimeError("get_consensus failed: %d" % err)
        return n.value

    def total_cells(self):
        v = C.c_uint64(0)
        if self._L.gw_poa_total_cells(self._h, C.byref(v)) != 0:
            raise RuntimeError(self._L.gw_last_error().decode())
        return v.value

    def get_consensus(self):
        """Returns (consensus strings, per-base coverages, per-group status)."""
        n = C.c_int32(0)
        err = self._L.gw_poa_get_consensus(self._h, C.byref(n))
        if err == output_type_unavailable:
            raise RuntimeError("Output type not requested during batch initialization")
        if err < 0:
            raise RuntimeError(self._L.gw_last_error().decode())
        cons, cov, status = [], [], []
        for i in range(n.value):
            ln = C.c_int32(0)
            p = self._L.gw_poa_consensus_str(self._h, i, C.byref(ln))
            cons.append(C.string_at(p, ln.value).decode("utf-8"))
            q = self._L.gw_poa_consensus_coverage(self._h, i, C.byref(ln))
            cov.append([q[k] for k in range(ln.value)])
            status.append(self._L.gw_poa_output_status(self._h, i))
        return (cons, cov, status)

    def get_msa_native(self):
        """D2H + host unpack inside the library (Batch::get_msa), without marshalling the rows to Python.
        Returns the group count; collect_msa(count) fetches the rows afterwards."""
        n = C.c_int32(0)
        err = self._L.gw_poa_get_msa(self._h, C.byref(n))
        if err == output_type_unavailable:
            raise RuntimeError("Output type not requested during batch initialization")
        if err < 0:
            raise RuntimeError(self._L.gw_last_error().decode())
        return n.value

    def get_msa(self):
        """Returns (msa[group][sequence], per-group status)."""
        return self.collect_msa(self.get_msa_native())

    def collect_msa_one(self, i):
        """(rows, status) of group i after get_msa_native()."""
        rows = []
        for r in range(self._L.gw_poa_msa_rows(self._h, i)):
            ln = C.c_int32(0)
            p = self._L.gw_poa_msa_row(self._h, i, r, C.byref(ln))
            rows.append(C.string_at(p, ln.value).decode("utf-8"))
        return rows, self._L.gw_poa_output_status(self._h, i)

    def collect_msa(self, count):
        msa, status = [], []
        for i in range(count):
            rows = []
            for r in range(self._L.gw_poa_msa_rows(self._h, i)):
                ln = C.c_int32(0)
                p = self._L.gw_poa_msa_row(self._h, i, r, C.byref(ln))
                rows.append(C.string_at(p, ln.value).decode("utf-8"))
            msa.append(rows)
            status.append(self._L.gw_poa_output_status(self._h, i))
        return (msa, status)

    def get_graphs(self):
        """Returns (networkx.DiGraph per group, per-group status)."""
        import networkx as nx
        n = C.c_int32(0)
        if self._L.gw_poa_get_graphs(self._h, C.byref(n)) != 0:
            raise RuntimeError(self._L.gw_last_error().decode())
        graphs, status = [], []
        si = 0
        for i in range(n.value):
            nn = self._L.gw_poa_graph_num_nodes(self._h, i)
            ne = self._L.gw_poa_graph_num_edges(self._h, i)
            labels = np.zeros(max(nn, 1), np.uint8)
            src = np.zeros(max(ne, 1), np.int32)
            dst = np.zeros(max(ne, 1), np.int32)
            w = np.zeros(max(ne, 1), np.int32)
            self._L.gw_poa_graph_copy(self._h, i, labels.ctypes.data, src.ctypes.data, dst.ctypes.data, w.ctypes.data)
            g = nx.DiGraph()
            for e in range(ne):
                g.add_edge(int(src[e]), int(dst[e]), weight=int(w[e]))
            nx.set_node_attributes(g, {k: {"label": chr(labels[k])} for k in g.nodes})
            graphs.append(g)
            status.append(self._L.gw_poa_output_status(self._h, si) if si < n.value else 0)
            si += 1
        return (graphs, status)

    def reset(self):
        """Reset the batch object. Involves deleting all windows previously assigned to batch object."""
        self._L.gw_poa_reset(self._h)


# ---- cudapoa/multi_device.hpp -----------------------------------------------------------------------------------
def process_windows_multi_device(windows, max_sequences_per_poa, max_sequence_size, devices=(0,), batches_per_device=1,
                                 memory_per_device=-1, output_type="consensus", band_mode="static_band",
                                 alignment_band_width=256, max_consensus_size=None, max_nodes_per_graph=None,
                                 matrix_sequence_dimension=None, max_banded_pred_distance=None, gap_score=-8,
                                 mismatch_score=-6, match_score=8):
    """cudapoa::process_windows_multi_device: every window through one worker (host thread + stream + Batch) per entry of
    `devices` x batches_per_device; a device id may repeat (logical shards of one device). Windows are pulled from a shared
    cursor and results are placed by global window index, so the output does not depend on the worker layout.
    Returns dict(consensus, coverage | msa, status, worker, launches, seconds)."""
    L = _bind(_native.host())
    vp, i32 = C.c_void_p, C.c_int32
    L.gw_poa_multi_device_run.restype = vp
    L.gw_poa_multi_device_run.argtypes = [i32, C.POINTER(i32), C.POINTER(C.c_char_p), C.POINTER(i32),
                                          C.POINTER(_native.PoaBatchConfig), C.POINTER(i32), i32, i32, C.c_int64, C.c_int8,
                                          C.c_int16, C.c_int16, C.c_int16]
    for name in ("gw_poa_multi_destroy", "gw_poa_multi_launches", "gw_poa_multi_seconds"):
        getattr(L, name).argtypes = [vp]
    L.gw_poa_multi_seconds.restype = C.c_double
    for name in ("gw_poa_multi_status", "gw_poa_multi_worker", "gw_poa_multi_msa_rows"):
        getattr(L, name).argtypes = [vp, i32]
    L.gw_poa_multi_consensus.restype = C.POINTER(C.c_char)
    L.gw_poa_multi_consensus.argtypes = [vp, i32, C.POINTER(i32)]
    L.gw_poa_multi_coverage.restype = C.POINTER(C.c_uint16)
    L.gw_poa_multi_coverage.argtypes = [vp, i32, C.POINTER(i32)]
    L.gw_poa_multi_msa_row.restype = C.POINTER(C.c_char)
    L.gw_poa_multi_msa_row.argtypes = [vp, i32, i32, C.POINTER(i32)]
    if band_mode not in _BAND_MODES:
        raise RuntimeError("Unknown band_mode provided.")
    mx_consensus = 2 * max_sequence_size if max_consensus_size is None else max_consensus_size
    nodes = 3 * max_sequence_size if max_nodes_per_graph is None else max_nodes_per_graph
    if matrix_sequence_dimension is None:
        msd = max_sequence_size if band_mode == "full_band" else (
            alignment_band_width + 8 if band_mode.startswith("static") else 2 * (alignment_band_width + 8))
    else:
        msd = matrix_sequence_dimension
    pred = 2 * ((alignment_band_width + 127) // 128 * 128) if max_banded_pred_distance is None else max_banded_pred_distance
    cfg = _native.PoaBatchConfig()
    if L.gw_poa_batch_config_full(C.byref(cfg), max_sequence_size, mx_consensus, nodes, alignment_band_width,
                                  max_sequences_per_poa, msd, _BAND_MODES[band_mode], pred) != 0:
        raise ValueError(L.gw_last_error().decode())
    raw = [[s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in w] for w in windows]
    flat = [b for w in raw for b in w]
    n = len(raw)
    per_window = (i32 * max(n, 1))(*[len(w) for w in raw])
    seqs = (C.c_char_p * max(len(flat), 1))(*flat)
    lens = (i32 * max(len(flat), 1))(*[len(b) for b in flat])
    devs = (i32 * len(devices))(*devices)
    mask = 2 if output_type == "msa" else 1
    h = L.gw_poa_multi_device_run(n, per_window, seqs, lens, C.byref(cfg), devs, len(devices), batches_per_device,
                                  int(memory_per_device), mask, gap_score, mismatch_score, match_score)
    if not h:
        raise RuntimeError(L.gw_last_error().decode())
    try:
        out = dict(status=[L.gw_poa_multi_status(h, w) for w in range(n)], worker=[L.gw_poa_multi_worker(h, w) for w in range(n)],
                   launches=L.gw_poa_multi_launches(h), seconds=L.gw_poa_multi_seconds(h),
                   seconds_after_creation=_multi_seconds_after_creation(L, h))
        ln = i32(0)
        if mask == 1:
            out["consensus"], out["coverage"] = [], []
            for w in range(n):
                p = L.gw_poa_multi_consensus(h, w, C.byref(ln))
                out["consensus"].append(C.string_at(p, ln.value).decode())
                q = L.gw_poa_multi_coverage(h, w, C.byref(ln))
                out["coverage"].append([q[k] for k in range(ln.value)])
        else:
            out["msa"] = []
            for w in range(n):
                rows = []
                for r in range(L.gw_poa_multi_msa_rows(h, w)):
                    p = L.gw_poa_multi_msa_row(h, w, r, C.byref(ln))
                    rows.append(C.string_at(p, ln.value).decode())
                out["msa"].append(rows)
        return out
    finally:
        L.gw_poa_multi_destroy(h)


class SizeClassPlan:
    """cudapoa::plan_size_classes: windows binned geometrically by their longest read, one BatchConfig per class, so
    that all classes are resident and run at once (process_windows_size_classes). Host-only.
    .configs (BatchConfig dicts), .groups (window indices per class), .bytes_per_window, .total_bytes."""

    def __init__(self, windows, msa_flag=False, band_width=256, band_mode="adaptive_band", adaptive_storage_factor=2.0,
                 graph_length_factor=3.0, max_pred_distance=0, mismatch_score=-6, gap_score=-8, match_score=8):
        L = _bind(_native.host())
        vp, i32 = C.c_void_p, C.c_int32
        L.gw_poa_plan_size_classes.restype = vp
        L.gw_poa_plan_size_classes.argtypes = [i32, C.POINTER(i32), C.POINTER(i32), i32, i32, i32, C.c_float, C.c_float, i32, i32, i32, i32]
        L.gw_poa_size_plan_destroy.argtypes = [vp]
        L.gw_poa_size_plan_classes.argtypes = [vp]
        L.gw_poa_size_plan_total_bytes.restype = C.c_int64
        L.gw_poa_size_plan_total_bytes.argtypes = [vp]
        L.gw_poa_size_plan_class.argtypes = [vp, i32, C.POINTER(_native.PoaBatchConfig), C.POINTER(C.c_int64), C.POINTER(i32)]
        L.gw_poa_size_plan_windows.argtypes = [vp, i32, C.POINTER(i32)]
        self._L = L
        n = len(windows)
        longest = (i32 * max(n, 1))(*[max((len(s) for s in g), default=0) for g in windows])
        reads = (i32 * max(n, 1))(*[len(g) for g in windows])
        self._h = L.gw_poa_plan_size_classes(n, longest, reads, int(msa_flag), band_width, _BAND_MODES[band_mode],
                                             adaptive_storage_factor, graph_length_factor, max_pred_distance, mismatch_score,
                                             gap_score, match_score)
        if not self._h:
            raise RuntimeError(L.gw_last_error().decode())
        fields = [f[0] for f in _native.PoaBatchConfig._fields_]
        self.configs, self.groups, self.bytes_per_window = [], [], []
        for k in range(L.gw_poa_size_plan_classes(self._h)):
            cfg, b, m = _native.PoaBatchConfig(), C.c_int64(0), i32(0)
            L.gw_poa_size_plan_class(self._h, k, C.byref(cfg), C.byref(b), C.byref(m))
            ids = (i32 * max(m.value, 1))()
            L.gw_poa_size_plan_windows(self._h, k, ids)
            self.configs.append({f: getattr(cfg, f) for f in fields})
            self.groups.append([ids[i] for i in range(m.value)])
            self.bytes_per_window.append(b.value)
        self.total_bytes = L.gw_poa_size_plan_total_bytes(self._h)

    def admission_gates(self, compute_units=256):
        """cudapoa::size_class_admission_gates: per class, the class whose end it waits for on the device (-1: none)."""
        out = (C.c_int32 * max(len(self.groups), 1))()
        self._L.gw_poa_size_plan_admission_gates.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        if self._L.gw_poa_size_plan_admission_gates(self._h, compute_units, out) != 0:
            raise RuntimeError(self._L.gw_last_error().decode())
        return [out[k] for k in range(len(self.groups))]

    def keep(self, window_ids):
        """Restrict the plan to these windows (a rank's share of a multi-GPU job); the configs stay those of the whole set."""
        n = max(max((max(g) for g in self.groups if g), default=-1), max(window_ids, default=-1)) + 1
        flags = (C.c_uint8 * max(n, 1))()
        for w in window_ids:
            flags[w] = 1
        self._L.gw_poa_size_plan_keep.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int32]
        if self._L.gw_poa_size_plan_keep(self._h, flags, n) != 0:
            raise RuntimeError(self._L.gw_last_error().decode())
        wanted = set(window_ids)
        self.groups = [[w for w in g if w in wanted] for g in self.groups]
        self.total_bytes = self._L.gw_poa_size_plan_total_bytes(self._h)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.gw_poa_size_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass


def _multi_seconds_after_creation(L, h):
    L.gw_poa_multi_seconds_after_creation.argtypes = [C.c_void_p]
    L.gw_poa_multi_seconds_after_creation.restype = C.c_double
    return L.gw_poa_multi_seconds_after_creation(h)


def process_windows_size_classes(windows, plan, device=0, memory_budget=-1, output_type="msa", gap_score=-8, mismatch_score=-6,
                                 match_score=8, collect=True, digest=None):
    """cudapoa::process_windows_size_classes: every class of `plan` (SizeClassPlan) in its own batch on its own host
    thread and stream, all at once. Returns dict(status, worker (= class), launches, seconds (workers' wall time),
    compute_seconds (from all first fills to the last end), consensus + coverage | msa). digest(rows) replaces the MSA
    rows of a successful window as they are fetched."""
    L = _bind(_native.host())
    vp, i32 = C.c_void_p, C.c_int32
    L.gw_poa_size_classes_run.restype = vp
    L.gw_poa_size_classes_run.argtypes = [i32, C.POINTER(i32), C.POINTER(C.c_char_p), C.POINTER(i32), vp, i32, C.c_int64, C.c_int8,
                                          C.c_int16, C.c_int16, C.c_int16, C.POINTER(C.c_double)]
    for name in ("gw_poa_multi_destroy", "gw_poa_multi_launches", "gw_poa_multi_seconds"):
        getattr(L, name).argtypes = [vp]
    L.gw_poa_multi_seconds.restype = C.c_double
    for name in ("gw_poa_multi_status", "gw_poa_multi_worker", "gw_poa_multi_msa_rows"):
        getattr(L, name).argtypes = [vp, i32]
    L.gw_poa_multi_consensus.restype = C.POINTER(C.c_char)
    L.gw_poa_multi_consensus.argtypes = [vp, i32, C.POINTER(i32)]
    L.gw_poa_multi_coverage.restype = C.POINTER(C.c_uint16)
    L.gw_poa_multi_coverage.argtypes = [vp, i32, C.POINTER(i32)]
    L.gw_poa_multi_msa_row.restype = C.POINTER(C.c_char)
    L.gw_poa_multi_msa_row.argtypes = [vp, i32, i32, C.POINTER(i32)]
    if memory_budget < 0:
        from .cuda import cuda_get_mem_info
        memory_budget = int(0.9 * cuda_get_mem_info(device)[0])
    raw = [[s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in w] for w in windows]
    flat = [b for w in raw for b in w]
    n = len(raw)
    per_window = (i32 * max(n, 1))(*[len(w) for w in raw])
    seqs = (C.c_char_p * max(len(flat), 1))(*flat)
    lens = (i32 * max(len(flat), 1))(*[len(b) for b in flat])
    mask = 2 if output_type == "msa" else 1
    compute = C.c_double(0)
    h = L.gw_poa_size_classes_run(n, per_window, seqs, lens, plan._h, device, int(memory_budget), mask, gap_score, mismatch_score,
                                  match_score, C.byref(compute))
    if not h:
        raise RuntimeError(L.gw_last_error().decode())
    try:
        out = dict(status=[L.gw_poa_multi_status(h, w) for w in range(n)], worker=[L.gw_poa_multi_worker(h, w) for w in range(n)],
                   launches=L.gw_poa_multi_launches(h), seconds=L.gw_poa_multi_seconds(h), compute_seconds=compute.value,
                   seconds_after_creation=_multi_seconds_after_creation(L, h))
        ln = i32(0)
        if collect and mask == 1:
            out["consensus"], out["coverage"] = [], []
            for w in range(n):
                p = L.gw_poa_multi_consensus(h, w, C.byref(ln))
                out["consensus"].append(C.string_at(p, ln.value).decode())
                q = L.gw_poa_multi_coverage(h, w, C.byref(ln))
                out["coverage"].append([q[k] for k in range(ln.value)])
        elif collect:
            out["msa"] = []
            for w in range(n):
                rows = []
                for r in range(L.gw_poa_multi_msa_rows(h, w)):
                    p = L.gw_poa_multi_msa_row(h, w, r, C.byref(ln))
                    rows.append(C.string_at(p, ln.value).decode("utf-8"))
                out["msa"].append(digest(rows) if (digest and out["status"][w] == success) else rows)
        return out
    finally:
        L.gw_poa_multi_destroy(h)


# ---- cudapoa/utils.hpp: batch-shape planning and window-file readers -------------------------------------------
def _bind_utils(L):
    if getattr(L, "_gw_poa_utils_bound", False):
        return L
    i32, f32 = C.c_int32, C.c_float
    pi32, pcfg = C.POINTER(i32), C.POINTER(_native.PoaBatchConfig)
    L.gw_poa_bin_groups.argtypes = [i32, pi32, pi32, pi32, i32, i32, f32, f32, i32, pi32, i32, pi32, pcfg, pi32, pi32]
    L.gw_poa_get_multi_batch_sizes.argtypes = [i32, pi32, pi32, i32, i32, i32, f32, f32, i32, f32, i32, i32, i32, pi32,
                                               pcfg, pi32, pi32]
    L.gw_poa_estimate_max_poas.argtypes = [pcfg, i32, f32, i32, i32, i32]
    L.gw_poa_window_device_bytes.restype = C.c_int64
    L.gw_poa_window_device_bytes.argtypes = [pcfg, i32, i32, i32, i32]
    L.gw_windows_parse.restype = C.c_void_p
    L.gw_windows_parse.argtypes = [C.POINTER(C.c_char_p), i32, i32, i32]
    L.gw_windows_destroy.argtypes = [C.c_void_p]
    L.gw_windows_count.argtypes = [C.c_void_p]
    L.gw_windows_num_sequences.argtypes = [C.c_void_p, i32]
    L.gw_windows_sequence.restype = C.POINTER(C.c_char)
    L.gw_windows_sequence.argtypes = [C.c_void_p, i32, i32, pi32]
    L._gw_poa_utils_bound = True
    return L


def _plan_out(n):
    return (C.c_int32(0), (_native.PoaBatchConfig * max(n, 1))(), (C.c_int32 * max(n, 1))(), (C.c_int32 * max(n, 1))())


def _plan_unpack(nb, cfgs, per_batch, ids):
    fields = [f[0] for f in _native.PoaBatchConfig._fields_]
    out_cfgs = [{f: getattr(cfgs[b], f) for f in fields} for b in range(nb.value)]
    groups, pos = [], 0
    for b in range(nb.value):
        groups.append([ids[pos + k] for k in range(per_batch[b])])
        pos += per_batch[b]
    return out_cfgs, groups


def _i32_array(values):
    return (C.c_int32 * max(len(values), 1))(*values)


def bin_poa_groups(capacity, longest, reads, band_width=256, band_mode="adaptive_band", adaptive_storage_factor=2.0,
                   graph_length_factor=3.0, max_pred_distance=0, bins_capacity=None):
    """The binning rule of get_multi_batch_sizes given per-group capacities (no device query).
    Returns (list of BatchConfig dicts, list of group-index lists)."""
    L = _bind_utils(_native.host())
    n = len(capacity)
    nb, cfgs, per_batch, ids = _plan_out(n)
    bins = _i32_array(bins_capacity) if bins_capacity is not None else None
    rc = L.gw_poa_bin_groups(n, _i32_array(capacity), _i32_array(longest), _i32_array(reads), band_width,
                             _BAND_MODES[band_mode], adaptive_storage_factor, graph_length_factor, max_pred_distance,
                             bins, len(bins_capacity) if bins_capacity is not None else 0, C.byref(nb), cfgs, per_batch, ids)
    if rc != 0:
        raise RuntimeError(L.gw_last_error().decode())
    return _plan_unpack(nb, cfgs, per_batch, ids)


def plan_multi_batch_sizes(poa_groups, memory_budget_bytes, msa_flag=False, band_width=256, band_mode="adaptive_band",
                           adaptive_storage_factor=2.0, graph_length_factor=3.0, max_pred_distance=0, mismatch_score=-6,
                           gap_score=-8, match_score=8):
    """get_multi_batch_sizes for a stated device-memory budget instead of the free memory of the current device: the
    same per-group capacities (budget // bytes per window of the group's own shape) and the same binning rule, but
    reproducible on any box and without a GPU (the goldens of the long-read config are planned this way).
    Returns (list of BatchConfig dicts, list of group-index lists)."""
    L = _bind_utils(_bind(_native.host()))
    longest = [max((len(s) for s in g), default=0) for g in poa_groups]
    reads = [len(g) for g in poa_groups]
    capacity = []
    for ln, nr in zip(longest, reads):
        cfg = _native.PoaBatchConfig()
        if L.gw_poa_batch_config_default(C.byref(cfg), ln, nr, band_width, _BAND_MODES[band_mode], adaptive_storage_factor,
                                         graph_length_factor, max_pred_distance) != 0:
            raise ValueError(L.gw_last_error().decode())
        per_window = L.gw_poa_window_device_bytes(C.byref(cfg), int(msa_flag), mismatch_score, gap_score, match_score)
        if per_window <= 0:
            raise RuntimeError(L.gw_last_error().decode())
        capacity.append(int(min(memory_budget_bytes // per_window, 2 ** 31 - 1)))
    return bin_poa_groups(capacity, longest, reads, band_width, band_mode, adaptive_storage_factor, graph_length_factor,
                          max_pred_distance)


def get_multi_batch_sizes(poa_groups, msa_flag=False, band_width=256, band_mode="adaptive_band",
                          adaptive_storage_factor=2.0, graph_length_factor=3.0, max_pred_distance=0,
                          gpu_memory_usage_quota=0.9, mismatch_score=-6, gap_score=-8, match_score=8):
    """cudapoa::get_multi_batch_sizes (utils.hpp:36-69): poa_groups is a list of windows (lists of sequences).
    Returns (list of BatchConfig dicts, list of group-index lists). Needs a GPU (free-memory query)."""
    L = _bind_utils(_native.host())
    n = len(poa_groups)
    longest = [max((len(s) for s in g), default=0) for g in poa_groups]
    reads = [len(g) for g in poa_groups]
    nb, cfgs, per_batch, ids = _plan_out(n)
    rc = L.gw_poa_get_multi_batch_sizes(n, _i32_array(longest), _i32_array(reads), int(msa_flag), band_width,
                                        _BAND_MODES[band_mode], adaptive_storage_factor, graph_length_factor,
                                        max_pred_distance, gpu_memory_usage_quota, mismatch_score, gap_score, match_score,
                                        C.byref(nb), cfgs, per_batch, ids)
    if rc != 0:
        raise RuntimeError(L.gw_last_error().decode())
    return _plan_unpack(nb, cfgs, per_batch, ids)


def _parse(paths, fasta, total_windows):
    L = _bind_utils(_native.host())
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    h = L.gw_windows_parse(arr, len(paths), int(fasta), int(total_windows))
    if not h:
        raise RuntimeError(L.gw_last_error().decode())
    try:
        out = []
        for w in range(L.gw_windows_count(h)):
            seqs = []
            for s in range(L.gw_windows_num_sequences(h, w)):
                n = C.c_int32(0)
                p = L.gw_windows_sequence(h, w, s, C.byref(n))
                seqs.append(C.string_at(p, n.value).decode())
            out.append(seqs)
        return out
    finally:
        L.gw_windows_destroy(h)


def parse_cudapoa_file(filename, total_windows=-1):
    """cudapoa::parse_cudapoa_file (utils.hpp:112-137): list of windows, each a list of sequences."""
    return _parse([filename], False, total_windows)


def parse_fasta_files(input_paths, total_windows=-1):
    """cudapoa::parse_fasta_files (utils.hpp:147-162): one window per FASTA file."""
    return _parse(list(input_paths), True, total_windows)
