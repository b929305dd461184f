// The packed pass of band 128 (poa_window_kernel VARIANT 2) and its launcher: see the note at the top of gwhip_poa.hip.
#define GWHIP_POA_PART 4
#include "gwhip_poa.hip"
