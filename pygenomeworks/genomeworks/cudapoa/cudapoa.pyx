# distutils: language = c++
"""cudapoa bindings: CudaPoaBatch over cudapoa::Batch (API of pygenomeworks' genomeworks.cudapoa)."""
import networkx as nx

from cython.operator cimport dereference as deref
from libc.stdint cimport int8_t, int32_t, int64_t, uint16_t
from libcpp.memory cimport unique_ptr
from libcpp.pair cimport pair
from libcpp.string cimport string
from libcpp.vector cimport vector

from genomeworks.cuda.cuda cimport CudaStream
from genomeworks.cuda.cuda_runtime_api cimport _Stream
from genomeworks.cudapoa.graph cimport DirectedGraph
cimport genomeworks.cudapoa.cudapoa as cudapoa

_STATUS_NAMES = {
    cudapoa.success: "success",
    cudapoa.exceeded_maximum_poas: "exceeded_maximum_poas",
    cudapoa.exceeded_maximum_sequence_size: "exceeded_maximum_sequence_size",
    cudapoa.exceeded_maximum_sequences_per_poa: "exceeded_maximum_sequences_per_poa",
    cudapoa.node_count_exceeded_maximum_graph_size: "node_count_exceeded_maximum_graph_size",
    cudapoa.edge_count_exceeded_maximum_graph_size: "edge_count_exceeded_maximum_graph_size",
    cudapoa.exceeded_adaptive_banded_matrix_size: "exceeded_adaptive_banded_matrix_size",
    cudapoa.exceeded_maximum_predecessor_distance: "exceeded_maximum_predecessor_distance",
    cudapoa.loop_count_exceeded_upper_bound: "loop_count_exceeded_upper_bound",
    cudapoa.output_type_unavailable: "output_type_unavailable",
    cudapoa.zero_weighted_poa_sequence: "zero_weighted_poa_sequence",
    cudapoa.empty_poa_group: "empty_poa_group",
    cudapoa.generic_error: "generic_error",
}

_BAND_MODES = {
    "full_band": cudapoa.full_band,
    "static_band": cudapoa.static_band,
    "adaptive_band": cudapoa.adaptive_band,
    "static_band_traceback": cudapoa.static_band_traceback,
    "adaptive_band_traceback": cudapoa.adaptive_band_traceback,
}


def status_to_str(status):
    """Name of a cudapoa StatusType value."""
    try:
        return _STATUS_NAMES[status]
    except KeyError:
        raise RuntimeError("Unknown error status : " + str(status))


cdef class CudaPoaBatch:
    """A batch of POA groups (windows) processed together on one GPU."""
    cdef unique_ptr[cudapoa.Batch] batch
    cdef unique_ptr[cudapoa.BatchConfig] batch_size
    cdef object stream_ref  # keeps the CudaStream alive as long as the batch uses it

    def __cinit__(self, max_sequences_per_poa, max_sequence_size, max_gpu_mem, output_type="consensus",
                  band_mode="adaptive_band", device_id=0, stream=None, gap_score=-8, mismatch_score=-6, match_score=8,
                  alignment_band_width=256, max_consensus_size=None, max_nodes_per_graph=None,
                  matrix_sequence_dimension=None, max_banded_pred_distance=None, *args, **kwargs):
        """Args (as pygenomeworks): max_sequences_per_poa, max_sequence_size, max_gpu_mem (bytes), output_type
        ("consensus" | "msa"), band_mode ("full_band" | "static_band" | "adaptive_band" | "*_traceback"), device_id,
        stream (CudaStream or None), gap / mismatch / match scores, alignment_band_width, max_consensus_size,
        max_nodes_per_graph, matrix_sequence_dimension. Unknown keyword arguments are ignored."""
        cdef _Stream raw_stream = NULL
        cdef size_t handle
        if stream is not None:
            if not isinstance(stream, CudaStream):
                raise RuntimeError("Type for stream option must be CudaStream")
            handle = stream.stream
            raw_stream = <_Stream>handle
        self.stream_ref = stream

        cdef int8_t output_mask
        if output_type == "consensus":
            output_mask = cudapoa.consensus
        elif output_type == "msa":
            output_mask = cudapoa.msa
        else:
            raise RuntimeError("Unknown output_type provided. Must be consensus/msa.")
        if band_mode not in _BAND_MODES:
            raise RuntimeError("Unknown band_mode provided. Must be full_band/static_band/adaptive_band.")
        cdef cudapoa.BandMode mode = _BAND_MODES[band_mode]

        # defaults of the Python layer: consensus 2x, graphs 3x (full band) / 4x (banded) the read length
        cdef int32_t consensus_size = 2 * max_sequence_size if max_consensus_size is None else max_consensus_size
        cdef int32_t nodes
        cdef int32_t matrix_dim
        if band_mode == "full_band":
            nodes = 3 * max_sequence_size if max_nodes_per_graph is None else max_nodes_per_graph
            matrix_dim = max_sequence_size if matrix_sequence_dimension is None else matrix_sequence_dimension
        elif band_mode.startswith("static_band"):
            nodes = 4 * max_sequence_size if max_nodes_per_graph is None else max_nodes_per_graph
            matrix_dim = alignment_band_width + 8 if matrix_sequence_dimension is None else matrix_sequence_dimension
        else:
            nodes = 4 * max_sequence_size if max_nodes_per_graph is None else max_nodes_per_graph
            matrix_dim = 2 * (alignment_band_width + 8) if matrix_sequence_dimension is None else matrix_sequence_dimension
        # the reference passes an uninitialised value here (cudapoa.pyx:144,165); BatchConfig's own default is used
        cdef int32_t pred_distance = (2 * ((alignment_band_width + 127) // 128 * 128) if max_banded_pred_distance is None
                                      else max_banded_pred_distance)
        self.batch_size.reset(new cudapoa.BatchConfig(<int32_t>max_sequence_size, consensus_size, nodes,
                                                      <int32_t>alignment_band_width, <int32_t>max_sequences_per_poa,
                                                      matrix_dim, mode, pred_distance))
        cdef int64_t mem = <int64_t>max_gpu_mem
        self.batch = cudapoa.create_batch(device_id, raw_stream, mem, output_mask, deref(self.batch_size), gap_score,
                                          mismatch_score, match_score)

    def __init__(self, *args, **kwargs):
        # present so that Python subclasses can define their own __init__
        pass

    def add_poa_group(self, poa, weights=None):
        """Queue one POA group (a list of sequences; optionally one list of base weights per sequence).
        Returns (group status, list of per-sequence statuses)."""
        if not isinstance(poa, list):
            poa = [poa]
        if len(poa) < 1:
            raise RuntimeError("At least one sequence must be present in POA group")
        cdef cudapoa.Group group
        cdef cudapoa.Entry entry
        cdef vector[cudapoa.StatusType] seq_status
        cdef vector[vector[int8_t]] weight_store
        cdef char* raw
        encoded = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in poa]  # owns the bytes until the call returns
        if weights is not None:
            weight_store.resize(len(poa))
            for i, w in enumerate(weights):
                if w is not None:
                    for v in w:
                        weight_store[i].push_back(<int8_t>v)
        for i, b in enumerate(encoded):
            raw = b
            entry.seq = raw
            entry.length = len(b)
            entry.weights = NULL
            if weights is not None and weights[i] is not None:
                entry.weights = weight_store[i].data()
            group.push_back(entry)
        status = deref(self.batch).add_poa_group(seq_status, group)
        return (status, [s for s in seq_status])

    @property
    def total_poas(self):
        """Number of POA groups in the batch."""
        return deref(self.batch).get_total_poas()

    @property
    def batch_id(self):
        """Identifier of the underlying Batch object."""
        return deref(self.batch).batch_id()

    def generate_poa(self):
        """Launch partial order alignment for every group of the batch (asynchronous on the batch's stream)."""
        deref(self.batch).generate_poa()

    def get_msa(self):
        """-> (msa[group][sequence], per-group statuses)."""
        cdef vector[vector[string]] msa
        cdef vector[cudapoa.StatusType] status
        err = deref(self.batch).get_msa(msa, status)
        if err == cudapoa.output_type_unavailable:
            raise RuntimeError("Output type not requested during batch initialization")
        return ([[row.decode("utf-8") for row in group] for group in msa], [s for s in status])

    def get_consensus(self):
        """-> (consensus strings, per-base coverage lists, per-group statuses)."""
        cdef vector[string] consensus
        cdef vector[vector[uint16_t]] coverage
        cdef vector[cudapoa.StatusType] status
        err = deref(self.batch).get_consensus(consensus, coverage, status)
        if err == cudapoa.output_type_unavailable:
            raise RuntimeError("Output type not requested during batch initialization")
        return ([c.decode("utf-8") for c in consensus], [list(c) for c in coverage], [s for s in status])

    def get_graphs(self):
        """-> (one networkx.DiGraph per group: node attribute 'label', edge attribute 'weight'; per-group statuses)."""
        cdef vector[DirectedGraph] graphs
        cdef vector[cudapoa.StatusType] status
        cdef vector[pair[DirectedGraph.edge_t, DirectedGraph.edge_weight_t]] edges
        deref(self.batch).get_graphs(graphs, status)
        out = []
        for g in range(graphs.size()):
            edges = graphs[g].get_edges()
            digraph = nx.DiGraph()
            for e in range(edges.size()):
                digraph.add_edge(edges[e].first.first, edges[e].first.second, weight=edges[e].second)
            nx.set_node_attributes(digraph, {n: {"label": graphs[g].get_node_label(n).decode("utf-8")} for n in digraph.nodes})
            out.append(digraph)
        return (out, [s for s in status])

    def reset(self):
        """Drop every group from the batch."""
        deref(self.batch).reset()

    def __dealloc__(self):
        self.batch.reset()
        self.batch_size.reset()
