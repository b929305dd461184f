# The device runtime the bindings talk to. The module keeps the name the other .pxd / .pyx files cimport
# (genomeworks.cuda.cuda_runtime_api, as in the reference's pygenomeworks/genomeworks/cuda/cuda_runtime_api.pxd),
# but declares the HIP runtime: on MI355X a stream is a hipStream_t and there is no CUDA header anywhere.

cdef extern from *:
    ctypedef void* _Stream "hipStream_t"
    ctypedef int _Error "hipError_t"

cdef extern from "hip/hip_runtime_api.h":
    # streams
    cdef _Error hipStreamCreate(_Stream* s)
    cdef _Error hipStreamDestroy(_Stream s)
    cdef _Error hipStreamSynchronize(_Stream s)
    # errors
    cdef _Error hipGetLastError()
    cdef const char* hipGetErrorString(_Error e)
    cdef const char* hipGetErrorName(_Error e)
    # devices
    cdef _Error hipGetDeviceCount(int* count)
    cdef _Error hipSetDevice(int device)
    cdef _Error hipGetDevice(int* device)
    cdef _Error hipMemGetInfo(size_t* free, size_t* total)
