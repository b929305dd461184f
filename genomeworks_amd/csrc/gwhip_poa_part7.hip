// The full-band kernels with int16 scores (packed pass: poa_forward_moves_full.h, VARIANT 5 / 6) and their launcher: see the
// note at the top of gwhip_poa.hip.
#define GWHIP_POA_PART 7
#include "gwhip_poa.hip"
