/*
 * global_oracle.h -- CPU restatement of the reference's two non-default global aligners:
 *   AlignerGlobalUkkonen (cudaaligner/src/ukkonen_gpu.cu, aligner_global_ukkonen.cpp; band parameter p = 100) and
 *   AlignerGlobalMyers   (cudaaligner/src/myers_gpu.cu:240-315 backtrace over the full edit-distance matrix).
 * TEST INFRASTRUCTURE ONLY (see global_oracle.c for the pinning status).
 */
#ifndef GLOBAL_ORACLE_H
#define GLOBAL_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* AlignmentState, cudaaligner.hpp:52-58 */
enum { GO_MATCH = 0, GO_MISMATCH = 1, GO_INSERTION = 2, GO_DELETION = 3 };

/* One pair each. `path` (capacity >= query_size + target_size) receives the states back to front, exactly as the
   kernels write them (the host reverses, aligner_global.cpp:180). Return 0, or -1 when out of memory. */
int32_t ukkonen_oracle_align(const char* query, int32_t query_size, const char* target, int32_t target_size, int32_t p,
                             int8_t* path, int32_t* path_length);
int32_t myers_full_oracle_align(const char* query, int32_t query_size, const char* target, int32_t target_size,
                                int8_t* path, int32_t* path_length);
#ifdef __cplusplus
}
#endif
#endif
