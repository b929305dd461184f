# cudaaligner::Aligner and Alignment (include/claraparabricks/genomeworks/cudaaligner/{cudaaligner,alignment,aligner}.hpp).
from libc.stdint cimport int32_t, int64_t
from libcpp.memory cimport shared_ptr, unique_ptr
from libcpp.string cimport string
from libcpp.vector cimport vector

from genomeworks.cuda.cuda_runtime_api cimport _Stream


cdef extern from "claraparabricks/genomeworks/cudaaligner/cudaaligner.hpp" \
        namespace "claraparabricks::genomeworks::cudaaligner":
    cdef enum StatusType:
        success = 0
        uninitialized
        exceeded_max_alignments
        exceeded_max_length
        exceeded_max_alignment_difference
        generic_error

    cdef enum AlignmentType:
        global_alignment = 0
        unset

    cdef enum AlignmentState:
        match = 0
        mismatch
        insertion   # absent in the query, present in the target
        deletion    # present in the query, absent in the target

    cdef StatusType Init()


cdef extern from "claraparabricks/genomeworks/cudaaligner/alignment.hpp" \
        namespace "claraparabricks::genomeworks::cudaaligner":
    cdef enum CigarFormat:
        basic "claraparabricks::genomeworks::cudaaligner::CigarFormat::basic"
        extended "claraparabricks::genomeworks::cudaaligner::CigarFormat::extended"

    ctypedef struct FormattedAlignment:
        string query
        string pairing
        string target

    cdef cppclass Alignment:
        const string& get_query_sequence() except +
        const string& get_target_sequence() except +
        string convert_to_cigar() except +
        string convert_to_cigar(CigarFormat) except +
        AlignmentType get_alignment_type() except +
        StatusType get_status() except +
        bint is_optimal() except +
        int32_t get_edit_distance() except +
        const vector[AlignmentState]& get_alignment() except +
        FormattedAlignment format_alignment() except +


cdef extern from "claraparabricks/genomeworks/cudaaligner/aligner.hpp" \
        namespace "claraparabricks::genomeworks::cudaaligner":
    cdef cppclass Aligner:
        StatusType align_all() except +
        StatusType sync_alignments() except +
        StatusType add_alignment(const char*, int32_t, const char*, int32_t) except +
        const vector[shared_ptr[Alignment]]& get_alignments() except +
        void reset() except +

    unique_ptr[Aligner] create_aligner(int32_t, int32_t, int32_t, AlignmentType, _Stream, int32_t, int64_t) except +
