# distutils: language = c++
"""cudaaligner bindings: CudaAlignerBatch over cudaaligner::Aligner (API of pygenomeworks' genomeworks.cudaaligner)."""
from cython.operator cimport dereference as deref
from libc.stdint cimport int64_t
from libcpp.memory cimport shared_ptr, unique_ptr
from libcpp.vector cimport vector

from genomeworks.cuda.cuda cimport CudaStream
from genomeworks.cuda.cuda_runtime_api cimport _Stream
cimport genomeworks.cudaaligner.cudaaligner as cudaaligner

_STATUS_NAMES = {
    cudaaligner.success: "success",
    cudaaligner.uninitialized: "uninitialized",
    cudaaligner.exceeded_max_alignments: "exceeded_max_alignments",
    cudaaligner.exceeded_max_length: "exceeded_max_length",
    cudaaligner.exceeded_max_alignment_difference: "exceeded_max_alignment_difference",
    cudaaligner.generic_error: "generic_error",
}
_STATE_NAMES = {cudaaligner.match: "m", cudaaligner.mismatch: "mm", cudaaligner.insertion: "i", cudaaligner.deletion: "d"}


def status_to_str(status):
    """Name of a cudaaligner StatusType value."""
    try:
        return _STATUS_NAMES[status]
    except KeyError:
        raise RuntimeError("Unknown error status : " + str(status))


class CudaAlignment:
    """One finished alignment: query, target, cigar, alignment_type ("global"), status, alignment (list of
    'm' / 'mm' / 'i' / 'd' per position) and format_alignment ([query line, pairing line, target line])."""

    def __init__(self, query, target, cigar, alignment_type, status, alignment, format_alignment):
        self.query = query
        self.target = target
        self.cigar = cigar
        self.alignment_type = self._alignment_type_str(alignment_type)
        self.status = status
        self.alignment = [self._alignment_state_enum_str(s) for s in alignment]
        self.format_alignment = format_alignment

    @staticmethod
    def _alignment_type_str(t):
        if t == cudaaligner.global_alignment:
            return "global"
        raise RuntimeError("Unknown alignment type encountered: " + str(t))

    @staticmethod
    def _alignment_state_enum_str(s):
        try:
            return _STATE_NAMES[s]
        except KeyError:
            raise RuntimeError("Unknown alignment state encountered: " + str(s))

    def __str__(self):
        return "{}\n{}\n{}\n".format(self.format_alignment[0], self.format_alignment[1], self.format_alignment[2])


cdef class CudaAlignerBatch:
    """A batch of (query, target) pairs aligned globally on one GPU."""
    cdef unique_ptr[cudaaligner.Aligner] aligner
    cdef public object stream

    def __cinit__(self, max_query_length, max_target_length, max_alignments, alignment_type="global", stream=None,
                  device_id=0, max_device_memory_allocator_caching_size=-1, *args, **kwargs):
        """Args (as pygenomeworks): max_query_length, max_target_length, max_alignments, alignment_type (only
        "global"), stream (CudaStream or None), device_id, max_device_memory_allocator_caching_size (bytes; -1 = all
        available device memory). Unknown keyword arguments are ignored."""
        cdef _Stream raw_stream = NULL
        cdef size_t handle
        if stream is not None:
            if not isinstance(stream, CudaStream):
                raise RuntimeError("Type for stream option must be CudaStream")
            handle = stream.stream
            raw_stream = <_Stream>handle
        self.stream = stream
        if alignment_type != "global":
            raise RuntimeError("Unknown alignment_type provided. Must be global.")
        cdef int64_t mem = <int64_t>max_device_memory_allocator_caching_size
        self.aligner = cudaaligner.create_aligner(max_query_length, max_target_length, max_alignments,
                                                  cudaaligner.global_alignment, raw_stream, device_id, mem)

    def __init__(self, *args, **kwargs):
        # present so that Python subclasses can define their own __init__
        pass

    def add_alignment(self, query, target):
        """Queue one pair (strings over ACGT). Returns the StatusType of the call."""
        q = query.encode("utf-8")
        t = target.encode("utf-8")
        return deref(self.aligner).add_alignment(q, len(q), t, len(t))

    def align_all(self):
        """Launch the alignment of every queued pair (asynchronous on the aligner's stream)."""
        deref(self.aligner).align_all()

    def get_alignments(self):
        """Wait for the device and return one CudaAlignment per pair, in the order the pairs were added."""
        deref(self.aligner).sync_alignments()
        cdef vector[shared_ptr[cudaaligner.Alignment]] res = deref(self.aligner).get_alignments()
        cdef cudaaligner.FormattedAlignment formatted
        cdef vector[cudaaligner.AlignmentState] states
        out = []
        for i in range(res.size()):
            formatted = deref(res[i]).format_alignment()
            states = deref(res[i]).get_alignment()
            out.append(CudaAlignment(
                deref(res[i]).get_query_sequence().decode("utf-8"),
                deref(res[i]).get_target_sequence().decode("utf-8"),
                deref(res[i]).convert_to_cigar().decode("utf-8"),
                deref(res[i]).get_alignment_type(),
                deref(res[i]).get_status(),
                [states[k] for k in range(states.size())],
                [formatted.query.decode("utf-8"), formatted.pairing.decode("utf-8"), formatted.target.decode("utf-8")]))
        return out

    def reset(self):
        """Drop every pair and result; the batch can be filled again."""
        deref(self.aligner).reset()

    def __dealloc__(self):
        self.aligner.reset()
