cdef class CudaStream:
    # the raw hipStream_t kept as an integer so that Python code can pass it around
    cdef size_t stream
