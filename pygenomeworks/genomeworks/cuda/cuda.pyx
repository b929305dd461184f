# distutils: language = c++
"""Runtime bindings (module name and API of pygenomeworks' genomeworks.cuda, on the HIP runtime)."""

cimport genomeworks.cuda.cuda_runtime_api as rt


class CudaRuntimeError(Exception):
    """A device runtime call failed; the message is '<error name> : <error string>'."""

    def __init__(self, error):
        cdef rt._Error e = error
        cdef bytes name = rt.hipGetErrorName(e)
        cdef bytes text = rt.hipGetErrorString(e)
        super().__init__("{} : {}".format(name.decode(), text.decode()))


cdef _check(rt._Error e):
    if e != 0:
        raise CudaRuntimeError(e)


cdef class CudaStream:
    """One device stream; kernels and copies queued on it run asynchronously in order."""

    def __cinit__(self):
        cdef rt._Stream s
        _check(rt.hipStreamCreate(&s))
        self.stream = <size_t>s

    def __init__(self):
        # present so that Python subclasses can define their own __init__
        pass

    def __dealloc__(self):
        cdef rt._Stream s = <rt._Stream>self.stream
        rt.hipStreamSynchronize(s)
        rt.hipStreamDestroy(s)

    def sync(self):
        """Block until everything queued on the stream has finished."""
        _check(rt.hipStreamSynchronize(<rt._Stream>self.stream))

    @property
    def stream(self):
        """The raw stream handle as an integer (size_t)."""
        return self.stream


def cuda_get_device_count():
    """Number of GPUs visible to the process."""
    cdef int n = 0
    _check(rt.hipGetDeviceCount(&n))
    return n


def cuda_set_device(device_id):
    """Make `device_id` the current device of the calling thread."""
    _check(rt.hipSetDevice(device_id))


def cuda_get_device():
    """The current device of the calling thread."""
    cdef int d = 0
    _check(rt.hipGetDevice(&d))
    return d


def cuda_get_mem_info(device_id):
    """(free bytes, total bytes) of `device_id`; the current device is restored afterwards."""
    cdef size_t free_bytes = 0
    cdef size_t total_bytes = 0
    before = cuda_get_device()
    if before != device_id:
        cuda_set_device(device_id)
    try:
        _check(rt.hipMemGetInfo(&free_bytes, &total_bytes))
    finally:
        if before != device_id:
            cuda_set_device(before)
    return (free_bytes, total_bytes)
