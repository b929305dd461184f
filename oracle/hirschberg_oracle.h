/*
 * hirschberg_oracle.h -- CPU restatement of the reference's default aligner (Hirschberg + Myers,
 * cudaaligner/src/hirschberg_myers_gpu.cu). TEST INFRASTRUCTURE ONLY (see hirschberg_oracle.c for the pinning status).
 */
#ifndef HIRSCHBERG_ORACLE_H
#define HIRSCHBERG_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* AlignmentState, cudaaligner.hpp:52-58 */
enum { HO_MATCH = 0, HO_MISMATCH = 1, HO_INSERTION = 2, HO_DELETION = 3 };

/* One pair. `path` (capacity >= query_size + target_size) receives the alignment states back to front, exactly as
   the kernel writes them (the host reverses, aligner_global.cpp:180); max_query_length is the aligner's constructor
   argument (it bounds the matrix of the full-Myers leaves). Returns 0, or 1 when the range stack overflowed
   (length 0, alignment stays uninitialized unless both sequences are empty). */
int32_t hirschberg_oracle_align(const char* query, int32_t query_size, const char* target, int32_t target_size,
                                int32_t max_query_length, int8_t* path, int32_t* path_length);
/* The same with the bit-vector kernels' view of the pair in query / target and the caller's characters in raw_query / raw_target
   (same lengths): the single-character leaf compares characters, everything else goes through the pattern tables. */
int32_t hirschberg_oracle_align_raw(const char* query, int32_t query_size, const char* target, int32_t target_size,
                                    const char* raw_query, const char* raw_target, int32_t max_query_length, int8_t* path,
                                    int32_t* path_length);
#ifdef __cplusplus
}
#endif
#endif
