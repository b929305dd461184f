"""Runtime helpers with the surface of pygenomeworks' genomeworks.cuda (cuda.pyx): streams, device selection,
memory info -- implemented on the HIP runtime through the host library."""
import ctypes as C

from . import _native


class CudaRuntimeError(Exception):
    """Raised when a HIP runtime call fails (name kept from pygenomeworks)."""

    def __init__(self, error):
        super().__init__("HIP runtime error code %d" % error)


def _check(err):
    if err != 0:
        raise CudaRuntimeError(err)


def cuda_get_device_count():
    n = C.c_int(0)
    _check(_native.host().gw_device_count(C.byref(n)))
    return n.value


def cuda_set_device(device_id):
    _check(_native.host().gw_set_device(int(device_id)))


def cuda_get_device():
    n = C.c_int(0)
    _check(_native.host().gw_get_device(C.byref(n)))
    return n.value


def cuda_get_mem_info(device_id):
    prev = cuda_get_device()
    cuda_set_device(device_id)
    free, total = C.c_size_t(0), C.c_size_t(0)
    try:
        _check(_native.host().gw_mem_info(C.byref(free), C.byref(total)))
    finally:
        cuda_set_device(prev)
    return (free.value, total.value)


class CudaStream:
    """Owning stream wrapper: `.stream` is the raw handle as an integer, `.sync()` blocks until it drains."""

    def __init__(self):
        h = C.c_void_p(0)
        _check(_native.host().gw_stream_create(C.byref(h)))
        self._h = h

    @property
    def stream(self):
        return self._h.value or 0

    def sync(self):
        _check(_native.host().gw_stream_sync(self._h))

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                _native.host().gw_stream_destroy(self._h)
        except Exception:
            pass
