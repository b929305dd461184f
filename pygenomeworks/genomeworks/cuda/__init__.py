"""Device runtime helpers: CudaStream, cuda_get_device_count / set_device / get_device / get_mem_info."""
from genomeworks.cuda.cuda import *  # noqa: F401,F403
from genomeworks.cuda.cuda import CudaRuntimeError, CudaStream  # noqa: F401
