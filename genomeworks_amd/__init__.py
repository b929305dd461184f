"""genomeworks_amd -- MI355X-native partial-order and pairwise alignment engine behind the GenomeWorks
cudapoa::Batch / cudaaligner::Aligner interfaces. Hand-written gfx950 HIP kernels (libgwhip.so) under host C++
(libgenomeworks_amd.so); this package is the Python mirror of pygenomeworks' bindings."""

__version__ = "0.1.0"
