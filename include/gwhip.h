/*
 * gwhip.h -- the thin C-ABI between host C++ (Batch / Aligner implementations, bindings, tests) and the
 * hand-written gfx950 HIP kernels (libgwhip.so). This is the ONLY way hipcc-compiled device code is entered.
 *
 * Every entry point is extern "C", takes plain pointers and sizes (device pointers unless noted), returns a
 * hipError_t as int (0 = success), never throws, launches asynchronously on the given stream and is
 * re-entrant from multiple host threads. The caller owns all memory.
 *
 * Reference seams replaced (paths relative to /root/reference):
 *   gwhip_poa_generate          <- generatePOA(...)              cudapoa/src/cudapoa_kernels.cuh:544-1076
 *                                  (generatePOAKernel :76-542 + generateConsensusKernel / generateMSAKernel)
 *   gwhip_poa_export_graphs     <- the device arrays CudapoaBatch::get_graphs copies back
 *                                  cudapoa/src/cudapoa_batch.cuh:315-393
 *   gwhip_poa_test_*            <- runNW / runNWbanded / runNWbandedTB / runTopSort / addAlignment /
 *                                  generateConsensusTestHost  (cudapoa_nw.cuh:499, cudapoa_nw_banded.cuh:608,
 *                                  cudapoa_nw_tb_banded.cuh:727, cudapoa_topsort.cuh:220,
 *                                  cudapoa_add_alignment.cuh:335, cudapoa_generate_consensus.cuh:395)
 *   gwhip_myers_banded          <- myers_banded_gpu(...)         cudaaligner/src/myers_gpu.cuh:55-74
 *   gwhip_myers_banded_workspace<- AlignerGlobalMyersBanded workspace sizing
 *                                  cudaaligner/src/aligner_global_myers_banded.cpp:318-360
 */
#ifndef GWHIP_H
#define GWHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gwhip_stream_t; /* hipStream_t */

#define GWHIP_MAX_NODE_EDGES 50      /* cudapoa_structs.cuh:24 */
#define GWHIP_MAX_NODE_ALIGNMENTS 50 /* cudapoa_structs.cuh:27 */

/* cudapoa.hpp:68-75 */
enum gwhip_band_mode
{
    GWHIP_FULL_BAND = 0,
    GWHIP_STATIC_BAND,
    GWHIP_ADAPTIVE_BAND,
    GWHIP_STATIC_BAND_TRACEBACK,
    GWHIP_ADAPTIVE_BAND_TRACEBACK
};

/* WindowDetails, cudapoa_structs.cuh:70-87 (same fields; 32 bytes) */
typedef struct gwhip_window_details
{
    uint16_t num_seqs;
    int32_t seq_len_buffer_offset;
    int32_t seq_starts;
    uint64_t scores_offset;
    int32_t scores_width;
} gwhip_window_details;

/* BatchConfig (batch.hpp:60-86) + scores + output mask + the type selection of cudapoa_limits.hpp:34-59 */
typedef struct gwhip_poa_config
{
    int32_t max_sequence_size;
    int32_t max_consensus_size;
    int32_t max_nodes_per_graph;
    int32_t matrix_sequence_dimension;
    int32_t alignment_band_width;
    int32_t max_sequences_per_poa;
    int32_t band_mode;
    int32_t max_banded_pred_distance;
    int32_t gap_score, mismatch_score, match_score;
    int32_t output_mask;   /* 1 consensus, 2 msa (cudapoa.hpp:81-85) */
    int32_t score32;       /* ScoreT int32 (else int16) */
    int32_t size32;        /* SizeT  int32 (else int16) */
    int32_t trace16;       /* TraceT int16 (else int8) */
    int32_t spoa_accurate; /* racon topsort inside the graph build (reference build flag SPOA_ACCURATE) */
} gwhip_poa_config;

typedef struct gwhip_poa_args
{
    gwhip_poa_config cfg;
    int32_t total_windows;
    /* inputs, device resident (the four arrays CudapoaBatch::generate_poa uploads, cudapoa_batch.cuh:171-178).
       `sequences` / `base_weights` must be followed by >= 2048 readable zero bytes. */
    const uint8_t* sequences;
    const int8_t* base_weights;
    int32_t* sequence_lengths; /* [sum num_seqs]; element [seq_len_buffer_offset] of each window is overwritten
                                  with the final node count (cudapoa_kernels.cuh:506) */
    const gwhip_window_details* window_details;
    /* outputs, device: consensus[total_windows*max_consensus_size] (reversed, NUL terminated; [0]==0xFF =>
       [1] is the StatusType), coverage likewise (uint16), msa[total_windows*max_seqs*max_consensus_size] */
    uint8_t* consensus;
    uint16_t* coverage;
    uint8_t* msa;
    /* opaque scratch of gwhip_poa_workspace_bytes() bytes, 256-byte aligned; graphs live here afterwards */
    void* workspace;
    size_t workspace_bytes;
    /* optional: per-window DP cell counter (sum over reads of rows x band cells), uint64[total_windows] */
    uint64_t* cells;
    /* optional hipEvent_t recorded on `stream` between the graph-build kernel and the consensus/MSA kernel,
       so a caller can time the dominant kernel with events on the launch stream */
    void* event_after_graph_build;
    /* optional profiling aid: per-window cycle totals (s_memtime ticks) of the graph-build phases,
       uint64[total_windows][6] = {row table, NW forward, sink+traceback, graph merge, topsort, other} */
    uint64_t* phase_cycles;
    /* optional: two zeroed uint32 words, device. With them a batch of more windows than the device holds at one wavefront per
       SIMD runs as a PERSISTENT grid -- one block per SIMD, each taking the next window from the counter when it has finished
       one -- instead of one block per window (replacement blocks land on SIMDs that are still busy: 2048 full-band windows
       took 2.47 x the time of 1024). The last block to finish zeroes both words again. NULL: one block per window. */
    uint32_t* work_counters;
    /* non-zero: other batches' kernels may run on the device at the same time (several Batch objects on host threads). The
       consensus kernel then keeps to the LDS footprint of the graph-build kernel's blocks (tables for 2176 nodes, larger graphs
       through the HBM routine) even for small batches: a 55 KB block needs two neighbouring slots of a CU to fall free at once
       and was seen waiting 46 ms for them behind another batch's window kernel (profiles/r06_multibatch_timeline.txt). */
    int32_t shared_device;
} gwhip_poa_args;

/* Bytes of scratch needed for `windows` windows under cfg (host function, no GPU needed).
   full band uses sum_scores_width = sum over windows of the per-window score row width (cudapoa_batch.cuh:502-507);
   pass 0 to size for the worst case. */
size_t gwhip_poa_workspace_bytes(const gwhip_poa_config* cfg, int32_t windows, uint64_t sum_scores_width);

/* Device bytes per window excluding the score/trace matrix, and bytes of the matrix: the two terms of
   max_poas = avail / (per_poa + matrix) (allocate_block.hpp:75-89), for OUR layout. Host function. */
void gwhip_poa_bytes_per_window(const gwhip_poa_config* cfg, int64_t* per_poa, int64_t* per_matrix);

/* How many windows of this configuration the current device runs side by side at full speed: one wavefront per SIMD
   (4 x compute units) for the configurations whose graph-build kernel is one wavefront per window with its tables in LDS
   (16-bit ids, <= 3072 graph rows, reads that fit the LDS copy), 0 for the others (long reads: blocks of eight wavefronts
   that are admitted by residency, host/multi_device.cpp). A launch lasts a whole number of such rounds, so a host that
   spreads windows over several batches fills them in multiples of it (2048 full-band windows as 1400 + 648: 145 ms; as
   1024 + 1024: 107 ms; profiles/r06_multibatch_timeline.txt). */
int32_t gwhip_poa_resident_windows(const gwhip_poa_config* cfg);

int gwhip_poa_generate(const gwhip_poa_args* args, gwhip_stream_t stream);

/* Export graphs in the reference's device layout for get_graphs: nodes u8[W*max_nodes],
   incoming_edges int32[W*max_nodes*50], incoming_edge_weights u16[same], incoming_edge_count u16[W*max_nodes],
   outgoing_edges int32[W*max_nodes*50] (may be NULL), outgoing_edge_count u16 (may be NULL). */
int gwhip_poa_export_graphs(const gwhip_poa_args* args, uint8_t* nodes, int32_t* incoming_edges,
                            uint16_t* incoming_edge_weights, uint16_t* incoming_edge_count, int32_t* outgoing_edges,
                            uint16_t* outgoing_edge_count, gwhip_stream_t stream);
/* Same for windows [first_window, first_window + n_windows) of the batch only; the output arrays hold n_windows
   graphs (window first_window + k at slot k), so a caller can export a large batch through bounded temporaries. */
int gwhip_poa_export_graphs_range(const gwhip_poa_args* args, int32_t first_window, int32_t n_windows, uint8_t* nodes,
                                  int32_t* incoming_edges, uint16_t* incoming_edge_weights, uint16_t* incoming_edge_count,
                                  int32_t* outgoing_edges, uint16_t* outgoing_edge_count, gwhip_stream_t stream);

/* ---- unit hooks (device pointers, reference array layout with SizeT=int32, 50 slots per node) ---- */
typedef struct gwhip_poa_test_graph
{
    const uint8_t* nodes;
    const int32_t* graph;          /* sorted_poa */
    const int32_t* node_id_to_pos;
    int32_t graph_count;
    const uint16_t* incoming_edge_count;
    const int32_t* incoming_edges;
    const uint16_t* outgoing_edge_count;
    const int32_t* outgoing_edges; /* only topsort needs it */
} gwhip_poa_test_graph;

/* mode: gwhip_band_mode. Outputs alignment_graph/alignment_read (int32, >= graph_count+read_length+2 entries)
   and *aligned_nodes (device int32). scratch = gwhip_poa_test_nw_scratch_bytes(). */
size_t gwhip_poa_test_nw_scratch_bytes(const gwhip_poa_config* cfg);
int gwhip_poa_test_nw(const gwhip_poa_config* cfg, const gwhip_poa_test_graph* g, const uint8_t* read,
                      int32_t read_length, void* scratch, int32_t* alignment_graph, int32_t* alignment_read,
                      int32_t* aligned_nodes, gwhip_stream_t stream);
int gwhip_poa_test_topsort(int32_t* sorted_poa, int32_t* node_id_to_pos, int32_t node_count,
                           const uint16_t* incoming_edge_count, const int32_t* outgoing_edges,
                           const uint16_t* outgoing_edge_count, uint16_t* local_incoming_edge_count,
                           gwhip_stream_t stream);
int gwhip_poa_test_add_alignment(uint8_t* nodes, int32_t* node_count, int32_t* node_alignments,
                                 uint16_t* node_alignment_count, int32_t* incoming_edges,
                                 uint16_t* incoming_edge_count, int32_t* outgoing_edges,
                                 uint16_t* outgoing_edge_count, uint16_t* incoming_edge_w, int32_t alignment_length,
                                 const int32_t* alignment_graph, const uint8_t* read, const int32_t* alignment_read,
                                 uint16_t* node_coverage_counts, const int8_t* base_weights,
                                 int32_t max_nodes_per_graph, int32_t* status, gwhip_stream_t stream);
int gwhip_poa_test_consensus(const uint8_t* nodes, int32_t node_count, const int32_t* graph,
                             const int32_t* node_id_to_pos, const int32_t* incoming_edges,
                             const uint16_t* incoming_edge_count, const int32_t* outgoing_edges,
                             const uint16_t* outgoing_edge_count, const uint16_t* incoming_edge_w,
                             int32_t* predecessors, int32_t* scores, uint8_t* consensus, uint16_t* coverage,
                             const uint16_t* node_coverage_counts, const int32_t* node_alignments,
                             const uint16_t* node_alignment_count, int32_t max_consensus_size,
                             gwhip_stream_t stream);

/* ---- cudaaligner: banded Myers ---- */
typedef struct gwhip_myers_args
{
    int32_t n_alignments;
    const char* sequences;          /* concatenated: q0 t0 q1 t1 ... (aligner_global_myers_banded.cpp:226-232) */
    const int64_t* sequence_starts; /* [2n+1] */
    const int32_t* max_bandwidths;  /* [n] */
    /* outputs (myers_gpu.cuh:55-74 contract): run-length encoded, reversed per alignment */
    int8_t* results;        /* ops, packed */
    int32_t* result_counts; /* run lengths, packed */
    int32_t* result_starts; /* [n+1] offsets into results (start of alignment i; [n] = total) */
    uint32_t* result_metadata; /* [n] bit31 = optimal, bits 0-26 = alignment index; entry i belongs to
                                  result_starts[i] (we keep index order, which the contract leaves unspecified) */
    int64_t results_capacity;  /* entries available in results / result_counts (>= total_sequence_length suffices) */
    void* workspace;
    size_t workspace_bytes;
    int64_t total_sequence_length;   /* sequence_starts[2n] (host-known), sizes the per-pair result slots */
    const int32_t* scheduling_index; /* optional device int32[n]: processing order, longest pairs first
                                        (aligner_global_myers_banded.cpp:306-309); NULL = input order */
    uint64_t* band_cells;            /* optional device uint64[n]: 32*n_words_band*target_len summed over attempts */
    int32_t* run_counts_out;         /* reserved */
    int32_t max_query_length;        /* optional hints (0 = unknown): longest query and largest max_bandwidth of the batch; */
    int32_t max_bandwidth_hint;      /* when both are known and small enough, the column state and the query patterns
                                        of every pair are kept in LDS instead of being re-read from the HBM workspace */
    /* A large batch processed in CHUNKS of consecutive pairs, so that the upload of chunk k + 1 overlaps the kernels of chunk k
       (host/cudaaligner.cpp): every chunk is one call with the arrays advanced to its first pair -- sequence_starts + 2 lo
       (absolute offsets into `sequences`, which stays the batch's base), max_bandwidths + lo, result_starts + lo,
       result_metadata + lo, band_cells + lo, a scheduling_index of chunk-local indices -- and
         total_sequence_length  = sequence_starts[2 hi] - sequence_starts[2 lo] (sizes the workspace's per-pair result slots),
         first_sequence_offset  = sequence_starts[2 lo] (host-known),
         index_base             = lo (added to the pair index kept in result_metadata),
         result_starts_base     = device pointer to the number of runs before the chunk: result_starts + lo of the batch's
                                  array, whose entry the previous chunk's call has written (NULL for the first chunk).
       `results` / `result_counts` stay the batch's packed arrays: the chunks append in input order. All zero / NULL = one call
       for the whole batch. */
    int32_t index_base;
    int64_t first_sequence_offset;
    const int32_t* result_starts_base;
    /* Optional second stream for what surrounds the alignment kernel: the workspace sizing ahead of it (reads sequence_starts,
       max_bandwidths and scheduling_index only) and the run-offset scan, compaction and band_cells copy behind it. With chunked
       calls the kernels of consecutive chunks then run back to back on `stream` while chunk k's compaction and chunk k + 1's
       sizing run beside them. Ordering is by events inside the call; the caller passes the SAME side stream to every chunk
       (result_starts_base is written there), has the inputs named above ordered before the side stream's work, and joins the
       side stream into `stream` (or drains both) before it reads results. NULL = everything on `stream`. */
    gwhip_stream_t side_stream;
    /* 0 = the whole call. A caller that pipelines chunks can split it: GWHIP_MYERS_SIZING runs only the workspace sizing (on
       side_stream, or stream when NULL); GWHIP_MYERS_ALIGN runs the alignment kernel on `stream` and what follows it, for the
       same args, once the caller has ordered the SIZING call's work before `stream` (an event of its own: the call then adds no
       hand-over from the side stream, so chunk k's kernel does not wait for sizing calls queued later). */
    int32_t phases;
    /* Optional mirrors of the packed runs in pinned (device-accessible) host memory: once the call's runs are in `results` /
       `result_counts`, entries below results_host_capacity are also copied there by a kernel of the same call (the device
       knows the range; the host would have to wait for the offsets first). When the batch's total stays below the capacity the
       host needs no copy of its own after the streams have drained. NULL / 0 = none. */
    int8_t* results_host;
    int32_t* result_counts_host;
    int64_t results_host_capacity;
    /* The same for the call's n_alignments + 1 entries of result_starts and its n_alignments entries of result_metadata (host
       pointers that correspond to the device pointers above; independent of the capacity). NULL = none. */
    int32_t* result_starts_host;
    uint32_t* result_metadata_host;
} gwhip_myers_args;
#define GWHIP_MYERS_SIZING 1
#define GWHIP_MYERS_ALIGN 2

size_t gwhip_myers_banded_workspace_bytes(int32_t n_alignments, const int64_t* sequence_starts_host,
                                          const int32_t* max_bandwidths_host);
/* The workspace is laid out per wave of 64 pairs in PROCESSING order (scheduling_index, or input order when NULL):
   size it with the same order that gwhip_myers_banded will be given (pairs whose wave does not fit report no result). */
size_t gwhip_myers_banded_workspace_bytes_ordered(int32_t n_alignments, const int64_t* sequence_starts_host,
                                                  const int32_t* max_bandwidths_host, const int32_t* scheduling_index_host);
/* The same sum in pieces, for hosts that size a large batch on several threads: `..._workspace_words` is the part of the
   slots [first_slot, first_slot + n_slots) of the processing order (first_slot a multiple of 64, n_slots too unless the
   piece is the last); `..._workspace_bytes_of_words` turns the pieces' total into the byte count the call above returns
   (total_sequence_length = sequence_starts_host[2 n] - sequence_starts_host[0]). */
int64_t gwhip_myers_banded_workspace_words(int32_t first_slot, int32_t n_slots, const int64_t* sequence_starts_host,
                                           const int32_t* max_bandwidths_host, const int32_t* scheduling_index_host);
size_t gwhip_myers_banded_workspace_bytes_of_words(int32_t n_alignments, int64_t total_sequence_length, int64_t words);
int gwhip_myers_banded(const gwhip_myers_args* args, gwhip_stream_t stream);
/* myers_banded_gpu_get_blocks_per_sm() (myers_gpu.cuh:53; the reference sizes its persistent launch with it,
   aligner_global_myers_banded.cpp:137-140): resident blocks of the one-lane-per-pair kernel per compute unit on `device`.
   Our launches are one block per wave of pairs, not persistent, so the host classes do not need it; exported for callers that
   size their batches by it. Returns a hipError_t as int. */
int gwhip_myers_occupancy(int device, int* blocks_per_cu);

/* Packed upload of the banded aligner's sequences (round 5): two bases per byte, base i in bits 4 (i & 1) .. 4 (i & 1) + 3 of
   byte i >> 1, codes 0..4 = 'A', 'C', 'T', 'G', 'N'. The host encodes a QUERY base as A / C / T / G -> 0..3 and anything else
   -> 4 (the kernels compare query characters with 'A', 'C', 'T', 'G' only, myers_gpu.cu:196-208) and a TARGET base c as
   (c >> 1) & 3 (all the kernels use of a target character, myers_gpu.cu:210-241): the decoded batch behaves exactly like the
   original one. Writes sequences[first .. last) from packed; asynchronous on `stream`. Returns a hipError_t as int. */
int gwhip_unpack_bases(const uint8_t* packed, char* sequences, int64_t first, int64_t last, gwhip_stream_t stream);

/* ---- cudaaligner: default aligner, Hirschberg + Myers (hirschberg_myers_gpu.cuh:45, hirschberg_myers_gpu.cu:684-701) ---- */
typedef struct gwhip_hirschberg_args
{
    int32_t n_alignments;
    const char* sequences;          /* concatenated: q0 t0 q1 t1 ... */
    const int64_t* sequence_starts; /* [2n+1] */
    int32_t max_query_length;       /* the aligner's constructor argument: bounds the matrix of the full-Myers leaves
                                       (aligner_global_hirschberg_myers.cpp:37-44), i.e. it is part of the result */
    int8_t* results;                /* alignment i: AlignmentState bytes, BACK TO FRONT (the host reverses,
                                       aligner_global.cpp:180), in the slot [sequence_starts[2i], sequence_starts[2i+2]) */
    int32_t* result_lengths;        /* [n]; 0 = no result (range stack overflow) or two empty sequences */
    void* workspace;
    size_t workspace_bytes;
} gwhip_hirschberg_args;

size_t gwhip_hirschberg_myers_workspace_bytes(int32_t n_alignments, const int64_t* sequence_starts_host, int32_t max_query_length);
int gwhip_hirschberg_myers(const gwhip_hirschberg_args* args, gwhip_stream_t stream);

/* ---- cudaaligner: AlignerGlobalUkkonen (ukkonen_gpu.cuh:43-50, ukkonen_gpu.cu:313-327) ---- */
typedef struct gwhip_ukkonen_args
{
    int32_t n_alignments;
    const char* sequences;          /* concatenated: q0 t0 q1 t1 ... */
    const int64_t* sequence_starts; /* [2n+1] */
    int32_t ukkonen_p;              /* band parameter; AlignerGlobalUkkonen fixes 100 (aligner_global_ukkonen.cpp:35) */
    int32_t max_length_difference;  /* max |query - target| over the batch, computed by the host exactly as the
                                       reference does before its launch (aligner_global_ukkonen.cpp:66-72) */
    int32_t max_sequence_length;    /* max(query, target) over the batch (the reference's max_target_query_length) */
    int8_t* results;                /* alignment i: AlignmentState bytes, BACK TO FRONT (the host reverses,
                                       aligner_global.cpp:180), in the slot [sequence_starts[2i], sequence_starts[2i+2]) */
    int32_t* result_lengths;        /* [n] */
    void* workspace;                /* 256-byte aligned; holds the int16 band storage of every pair */
    size_t workspace_bytes;
} gwhip_ukkonen_args;

size_t gwhip_ukkonen_workspace_bytes(int32_t n_alignments, const int64_t* sequence_starts_host, int32_t ukkonen_p);
int gwhip_ukkonen(const gwhip_ukkonen_args* args, gwhip_stream_t stream);

/* ---- misc ---- */
/* Copies the last error text of the calling thread (NUL terminated) and returns its length. */
/* ---- cudaaligner unit hooks (device pointers): the production device functions on one pair ---- */
/* myers_preprocess (hirschberg_myers_gpu.cu:227-242; Test_HirschbergMyers.cu:94-148): patterns[w * 8 + c],
   c = 0..3 forward A, C, T, G, c = 4..7 the same for the query read back to front; ceil(len / 32) x 8 words. */
int gwhip_myers_test_patterns(const char* query_d, int32_t query_length, uint32_t* patterns_d, gwhip_stream_t stream);
/* get_query_pattern(patterns, word_index, shift, x, reverse) for the shifts 0..31 (hirschberg_myers_gpu.cu:244-276;
   Test_HirschbergMyers.cu:150-211). scratch: 4 * ceil(len / 32) words. */
int gwhip_myers_test_get_pattern(const char* query_d, int32_t query_length, int32_t word_index, char x, int32_t reverse,
                                 uint32_t* scratch_d, uint32_t* out32_d, gwhip_stream_t stream);
/* One band attempt of the banded Myers kernel with a given band_width and p (the reference's
   myers_compute_scores_edit_dist_banded_test_kernel, Test_MyersAlgorithm.cu:42-97). The workspace holds, lane-interleaved
   (element k at word k * 64): pv[nwb * (t + 1)] | mv[...] | score[...] | patterns, column-major (word w, column j at
   j * nwb + w); diagonals_d receives {diagonal_begin, diagonal_end}. */
size_t gwhip_myers_test_banded_matrices_words(int32_t query_length, int32_t target_length, int32_t band_width);
int gwhip_myers_test_banded_matrices(const char* query_d, const char* target_d, int32_t query_length, int32_t target_length,
                                     int32_t band_width, int32_t p, uint32_t* workspace_d, int32_t* diagonals_d, gwhip_stream_t stream);

int gwhip_last_error_string(char* buf, size_t len);
/* Compile-time facts a test can assert without a GPU. */
const char* gwhip_build_arch(void); /* "gfx950" */
int gwhip_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
