// One (score type, id type) pair of the graph-build kernel and its launcher: see the note at the top of gwhip_poa.hip.
#define GWHIP_POA_PART 1
#include "gwhip_poa.hip"
