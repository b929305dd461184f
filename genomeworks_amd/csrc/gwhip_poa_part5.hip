// The two-pass packed pass of bands 384 / 512 (poa_window_kernel VARIANT 3, 4) and its launcher: see the note at the top of gwhip_poa.hip.
#define GWHIP_POA_PART 5
#include "gwhip_poa.hip"
