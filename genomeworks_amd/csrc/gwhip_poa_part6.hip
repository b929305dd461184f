// The traceback-buffer kernels with int16 scores, ids and traces (packed pass: poa_forward_moves_tb.h) and their launcher: see
// the note at the top of gwhip_poa.hip.
#define GWHIP_POA_PART 6
#include "gwhip_poa.hip"
