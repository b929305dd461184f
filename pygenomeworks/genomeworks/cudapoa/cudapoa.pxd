# cudapoa::Batch and friends (include/claraparabricks/genomeworks/cudapoa/{cudapoa,batch}.hpp).
from libc.stdint cimport int8_t, int16_t, int32_t, int64_t, uint16_t
from libcpp.memory cimport unique_ptr
from libcpp.string cimport string
from libcpp.vector cimport vector

from genomeworks.cuda.cuda_runtime_api cimport _Stream
from genomeworks.cudapoa.graph cimport DirectedGraph


cdef extern from "claraparabricks/genomeworks/cudapoa/cudapoa.hpp" namespace "claraparabricks::genomeworks::cudapoa":
    cdef enum StatusType:
        success = 0
        exceeded_maximum_poas
        exceeded_maximum_sequence_size
        exceeded_maximum_sequences_per_poa
        node_count_exceeded_maximum_graph_size
        edge_count_exceeded_maximum_graph_size
        exceeded_adaptive_banded_matrix_size
        exceeded_maximum_predecessor_distance
        loop_count_exceeded_upper_bound
        output_type_unavailable
        zero_weighted_poa_sequence
        empty_poa_group
        generic_error

    cdef enum BandMode:
        full_band = 0
        static_band
        adaptive_band
        static_band_traceback
        adaptive_band_traceback

    cdef enum OutputType:
        consensus = 0x1
        msa = 0x2

    cdef StatusType Init()


cdef extern from "claraparabricks/genomeworks/cudapoa/batch.hpp" namespace "claraparabricks::genomeworks::cudapoa":
    cdef struct Entry:
        const char* seq
        const int8_t* weights
        int32_t length

    ctypedef vector[Entry] Group

    cdef cppclass BatchConfig:
        int32_t max_sequence_size
        int32_t max_consensus_size
        int32_t max_nodes_per_graph
        int32_t matrix_sequence_dimension
        int32_t alignment_band_width
        int32_t max_sequences_per_poa
        BandMode band_mode
        int32_t max_banded_pred_distance
        # (max_seq_sz, max_seq_per_poa, band_width, banding, adaptive_storage_factor, graph_length_factor, max_pred_dist)
        BatchConfig(int32_t, int32_t, int32_t, BandMode, float, float, int32_t) except +
        # (max_seq_sz, max_consensus_sz, max_nodes_per_poa, band_width, max_seq_per_poa, matrix_seq_dim, banding, max_pred_distance)
        BatchConfig(int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, BandMode, int32_t) except +

    cdef cppclass Batch:
        StatusType add_poa_group(vector[StatusType]&, const Group&) except +
        void generate_poa() except +
        StatusType get_msa(vector[vector[string]]&, vector[StatusType]&) except +
        StatusType get_consensus(vector[string]&, vector[vector[uint16_t]]&, vector[StatusType]&) except +
        void get_graphs(vector[DirectedGraph]&, vector[StatusType]&) except +
        int get_total_poas() except +
        int batch_id() except +
        void reset() except +

    cdef unique_ptr[Batch] create_batch(int32_t, _Stream, int64_t, int8_t, const BatchConfig&, int16_t, int16_t, int16_t) except +
