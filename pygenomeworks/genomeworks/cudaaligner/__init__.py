"""Pairwise global alignment on the GPU: CudaAlignerBatch."""
from genomeworks.cudaaligner.cudaaligner import *  # noqa: F401,F403
from genomeworks.cudaaligner.cudaaligner import CudaAlignerBatch, CudaAlignment, status_to_str  # noqa: F401
