"""genomeworks -- Python bindings (Cython) of the MI355X-native cudapoa / cudaaligner libraries.

Same package layout, class, method and argument names as pygenomeworks of the reference
(pygenomeworks/genomeworks/{cuda,cudapoa,cudaaligner}); the extensions are compiled against this repository's
include/claraparabricks/genomeworks headers and linked with libgenomeworks_amd.so (host C++) + the HIP runtime."""
