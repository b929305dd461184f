/*
 * aligner_oracle.h -- CPU restatement of the reference's banded Myers global aligner
 * (cudaaligner/src/myers_gpu.cu:196-255,444-1021 and the host clamp aligner_global_myers_banded.cpp:174-178).
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline; never by the product.
 */
#ifndef ALIGNER_ORACLE_H
#define ALIGNER_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* AlignmentState, cudaaligner.hpp:52-58 */
enum { ALN_MATCH = 0, ALN_MISMATCH = 1, ALN_INSERTION = 2, ALN_DELETION = 3 };

int32_t aligner_oracle_host_max_bandwidth(int32_t max_bandwidth, int32_t query_length);

/* One pair through the kernel's per-alignment logic. ops/counts: run-length encoded, in the kernel's (reversed)
   order, capacity >= query_size + target_size. Returns 0 when a result exists, 1 when the pair gets no result
   (Alignment stays `uninitialized`). band_cells (optional) accumulates 32*n_words_band*target_size per attempt. */
int32_t aligner_oracle_myers_banded(const char* query, int32_t query_size, const char* target, int32_t target_size,
                                    int32_t max_bandwidth, int8_t* ops, int32_t* counts, int32_t* n_runs,
                                    int32_t* is_optimal, int64_t* band_cells);
#ifdef __cplusplus
}
#endif
#endif
