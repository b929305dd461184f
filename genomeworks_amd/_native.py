"""ctypes bindings of the two native libraries. Fails loudly when they are missing: there is no CPU fallback."""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.join(PKG, "lib")


class NativeLibraryMissing(ImportError):
    pass


class PoaConfig(C.Structure):
    """gwhip_poa_config (include/gwhip.h)"""
    _fields_ = [(n, C.c_int32) for n in (
        "max_sequence_size", "max_consensus_size", "max_nodes_per_graph", "matrix_sequence_dimension",
        "alignment_band_width", "max_sequences_per_poa", "band_mode", "max_banded_pred_distance",
        "gap_score", "mismatch_score", "match_score", "output_mask", "score32", "size32", "trace16", "spoa_accurate")]


class WindowDetails(C.Structure):
    """gwhip_window_details == WindowDetails (cudapoa_structs.cuh:70-87)"""
    _fields_ = [("num_seqs", C.c_uint16), ("seq_len_buffer_offset", C.c_int32), ("seq_starts", C.c_int32),
                ("scores_offset", C.c_uint64), ("scores_width", C.c_int32)]


class PoaArgs(C.Structure):
    _fields_ = [("cfg", PoaConfig), ("total_windows", C.c_int32), ("sequences", C.c_void_p),
                ("base_weights", C.c_void_p), ("sequence_lengths", C.c_void_p), ("window_details", C.c_void_p),
                ("consensus", C.c_void_p), ("coverage", C.c_void_p), ("msa", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("cells", C.c_void_p), ("event_after_graph_build", C.c_void_p), ("phase_cycles", C.c_void_p), ("work_counters", C.c_void_p), ("shared_device", C.c_int32)]


class MyersArgs(C.Structure):
    _fields_ = [("n_alignments", C.c_int32), ("sequences", C.c_void_p), ("sequence_starts", C.c_void_p),
                ("max_bandwidths", C.c_void_p), ("results", C.c_void_p), ("result_counts", C.c_void_p),
                ("result_starts", C.c_void_p), ("result_metadata", C.c_void_p), ("results_capacity", C.c_int64),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("total_sequence_length", C.c_int64),
                ("scheduling_index", C.c_void_p), ("band_cells", C.c_void_p), ("run_counts_out", C.c_void_p),
                ("max_query_length", C.c_int32), ("max_bandwidth_hint", C.c_int32),
                # chunked batches (include/gwhip.h); all zero = one call for the whole batch
                ("index_base", C.c_int32), ("first_sequence_offset", C.c_int64), ("result_starts_base", C.c_void_p),
                # pipelined chunks: second stream, phases of the call, pinned host mirrors of the results; all zero = none
                ("side_stream", C.c_void_p), ("phases", C.c_int32), ("results_host", C.c_void_p), ("result_counts_host", C.c_void_p),
                ("results_host_capacity", C.c_int64), ("result_starts_host", C.c_void_p), ("result_metadata_host", C.c_void_p)]


class PoaBatchConfig(C.Structure):
    """gw_poa_batch_config (include/gw_capi.h) == cudapoa::BatchConfig fields"""
    _fields_ = [(n, C.c_int32) for n in (
        "max_sequence_size", "max_consensus_size", "max_nodes_per_graph", "matrix_sequence_dimension",
        "alignment_band_width", "max_sequences_per_poa", "band_mode", "max_banded_pred_distance")]


_gwhip = None
_host = None


def _load(name):
    path = os.path.join(LIBDIR, name)
    if not os.path.exists(path):
        raise NativeLibraryMissing(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). genomeworks_amd has no CPU fallback.")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


def gwhip():
    """libgwhip.so: HIP kernels behind the thin C-ABI."""
    global _gwhip
    if _gwhip is None:
        L = _load("libgwhip.so")
        L.gwhip_poa_workspace_bytes.restype = C.c_size_t
        L.gwhip_poa_workspace_bytes.argtypes = [C.POINTER(PoaConfig), C.c_int32, C.c_uint64]
        L.gwhip_poa_bytes_per_window.argtypes = [C.POINTER(PoaConfig), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.gwhip_poa_generate.restype = C.c_int
        L.gwhip_poa_generate.argtypes = [C.POINTER(PoaArgs), C.c_void_p]
        L.gwhip_poa_export_graphs.restype = C.c_int
        L.gwhip_poa_export_graphs.argtypes = [C.POINTER(PoaArgs)] + [C.c_void_p] * 7
        L.gwhip_last_error_string.restype = C.c_int
        L.gwhip_last_error_string.argtypes = [C.c_char_p, C.c_size_t]
        L.gwhip_build_arch.restype = C.c_char_p
        L.gwhip_abi_version.restype = C.c_int
        _gwhip = L
    return _gwhip


def host():
    """libgenomeworks_amd.so: host C++ classes behind the object-level C API."""
    global _host
    if _host is None:
        gwhip()
        L = _load("libgenomeworks_amd.so")
        L.gw_last_error.restype = C.c_char_p
        L.gw_generate_window.restype = C.c_int64
        L.gw_generate_window.argtypes = [C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_void_p, C.c_int64, C.c_void_p]
        L.gw_generate_pairs.restype = C.c_int64
        L.gw_generate_pairs.argtypes = [C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        _host = L
    return _host


def gwhip_error():
    buf = C.create_string_buffer(512)
    gwhip().gwhip_last_error_string(buf, 512)
    return buf.value.decode(errors="replace")


GWHIP_SYMBOLS = [
    "gwhip_poa_workspace_bytes", "gwhip_poa_bytes_per_window", "gwhip_poa_generate", "gwhip_poa_export_graphs",
    "gwhip_last_error_string", "gwhip_build_arch", "gwhip_abi_version",
]
