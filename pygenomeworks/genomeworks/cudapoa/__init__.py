"""Partial order alignment on the GPU: CudaPoaBatch."""
from genomeworks.cudapoa.cudapoa import *  # noqa: F401,F403
from genomeworks.cudapoa.cudapoa import CudaPoaBatch, status_to_str  # noqa: F401
