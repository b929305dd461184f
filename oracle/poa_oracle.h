/*
 * poa_oracle.h -- CPU restatement of the reference cudapoa algorithm.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under genomeworks_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 * Parity status: pinned by the reference's inline known-answer vectors:
 * NW1-NW5, the 493-node banded==full case, topsort x3, addAlignment x5, consensus x5,
 * batch-level cases; and by the reference's own cudapoa library run on the CPU (its CUDA sources under
 * the SIMT emulator of oracle/simt -> oracle/_ref/libref_cudapoa_simt.so): the 77 windows of
 * tests/golden/reference_simt_windows.json.gz and fresh random windows, tests/test_reference_simt.py.
 * spoa itself is an empty un-vendored submodule in the reference checkout and
 * cudapoa/data/sample-windows.txt is stripped, so parity vs spoa / the End2End golden cannot be
 * had here (stated in DESIGN.md).
 */
#ifndef POA_ORACLE_H
#define POA_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POA_MAX_NODE_EDGES 50      /* cudapoa_structs.cuh:24 */
#define POA_MAX_NODE_ALIGNMENTS 50 /* cudapoa_structs.cuh:27 */
#define POA_CELLS_PER_THREAD 4     /* cudapoa_structs.cuh:34 */
#define POA_MIN_BAND_WIDTH 128     /* cudapoa_structs.cuh:35 (4 * WARP_SIZE(32)); API constant */
#define POA_BAND_RIGHT_PADDING 8   /* cudapoa_structs.cuh:36 */
#define POA_MAX_ADAPTIVE_BAND 1536 /* cudapoa_structs.cuh:38 */
#define POA_SHIFT_LEFT (-10)       /* cudapoa_structs.cuh:41 */
#define POA_SHIFT_RIGHT (-11)      /* cudapoa_structs.cuh:42 */
#define POA_NW_LOOP_FAILED (-1)    /* cudapoa_structs.cuh:53 */
#define POA_NW_ADAPTIVE_STORAGE_FAILED (-2)
#define POA_NW_TRACEBACK_BUFFER_FAILED (-3)
#define POA_KERNEL_ERROR 0xFF /* cudapoa_structs.cuh:49 */

/* cudapoa.hpp:34-49 StatusType (order matters: written to consensus[1]) */
enum {
    POA_SUCCESS = 0,
    POA_EXCEEDED_MAXIMUM_POAS,
    POA_EXCEEDED_MAXIMUM_SEQUENCE_SIZE,
    POA_EXCEEDED_MAXIMUM_SEQUENCES_PER_POA,
    POA_NODE_COUNT_EXCEEDED,
    POA_EDGE_COUNT_EXCEEDED,
    POA_EXCEEDED_ADAPTIVE_BANDED_MATRIX_SIZE,
    POA_EXCEEDED_MAXIMUM_PREDECESSOR_DISTANCE,
    POA_LOOP_COUNT_EXCEEDED,
    POA_OUTPUT_TYPE_UNAVAILABLE,
    POA_ZERO_WEIGHTED_SEQUENCE,
    POA_EMPTY_POA_GROUP,
    POA_GENERIC_ERROR
};

/* cudapoa.hpp:68-75 BandMode */
enum { POA_FULL_BAND = 0, POA_STATIC_BAND, POA_ADAPTIVE_BAND, POA_STATIC_BAND_TB, POA_ADAPTIVE_BAND_TB };

typedef struct poa_cfg
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
    int32_t score32;       /* ScoreT = int32 (else int16)   cudapoa_limits.hpp:34-45 */
    int32_t trace16;       /* TraceT = int16 (else int8)    cudapoa_limits.hpp:56-59 */
    int32_t output_mask;   /* 1 consensus, 2 msa            cudapoa.hpp:81-85 */
    int32_t spoa_accurate; /* racon topsort inside graph build (SPOA_ACCURATE build flag) */
} poa_cfg;

/* Graph arrays in the reference's SoA layout (cudapoa_structs.cuh:129-195) with SizeT=int32. */
typedef struct poa_graph
{
    uint8_t* nodes;
    int32_t* node_alignments;
    uint16_t* node_alignment_count;
    int32_t* incoming_edges;
    uint16_t* incoming_edge_count;
    int32_t* outgoing_edges;
    uint16_t* outgoing_edge_count;
    uint16_t* incoming_edge_weights;
    int32_t* sorted_poa;
    int32_t* node_id_to_pos;
    uint16_t* local_incoming_edge_count;
    int32_t* consensus_scores;
    int32_t* consensus_predecessors;
    uint8_t* node_marks;
    uint8_t* check_aligned_nodes;
    int32_t* nodes_to_visit;
    uint16_t* node_coverage_counts;
    uint16_t* outgoing_edges_coverage;
    uint16_t* outgoing_edges_coverage_count;
    int32_t* node_id_to_msa_pos;
    int32_t* sequence_begin_nodes_ids;
} poa_graph;

typedef struct poa_workspace poa_workspace;

/* fp32 band placement, cudapoa_nw_banded.cuh:67-78. Exposed so tests can pin IEEE behaviour. */
int32_t poa_band_start_for_row(int32_t row, float gradient, int32_t band_width, int32_t band_shift, int32_t max_column);

/* BatchConfig ctor #1 (batch.cu:34-70) + type selection (cudapoa_limits.hpp:34-59). */
void poa_cfg_init(poa_cfg* c, int32_t max_seq_sz, int32_t max_seq_per_poa, int32_t band_width, int32_t band_mode,
                  float adaptive_storage_factor, float graph_length_factor, int32_t max_pred_dist,
                  int32_t gap, int32_t mismatch, int32_t match, int32_t output_mask);
/* Recompute score32/trace16 after fields were set by hand (BatchConfig ctor #2, batch.cu:73-104). */
void poa_cfg_select_types(poa_cfg* c);

poa_workspace* poa_workspace_create(const poa_cfg* cfg);
void poa_workspace_destroy(poa_workspace* ws);
poa_graph* poa_workspace_graph(poa_workspace* ws);
/* number of int16 narrowing overflows observed in score stores since creation */
int64_t poa_workspace_overflow_events(const poa_workspace* ws);

/*
 * One window end to end: generatePOAKernel (cudapoa_kernels.cuh:200-541) followed by
 * generateConsensus (cudapoa_generate_consensus.cuh:141-283) and/or generateMSA
 * (cudapoa_generate_msa.cuh:128-227), per output_mask.
 *
 * seqs/weights: reads packed back to back, each padded to a multiple of 4 bytes (cudapoa_batch.cuh:537).
 * seq_lens[num_seqs]; seq_lens[0] is overwritten with the final node count like the device does
 * (cudapoa_kernels.cuh:506).
 * consensus[max_consensus_size], coverage[max_consensus_size] (reversed, as the kernel emits them;
 * consensus[0]==0xFF => consensus[1] is the StatusType).
 * msa[max_seqs*max_consensus_size] or NULL.
 * cells (may be NULL): sum over reads of N_s * C_s (SURVEY 8(d) work unit).
 * Returns the StatusType of the window (0 = success).
 */
int32_t poa_process_window(poa_workspace* ws, const uint8_t* seqs, const int8_t* weights, int32_t* seq_lens,
                           int32_t num_seqs, size_t seq_buf_bytes, uint8_t* consensus, uint16_t* coverage,
                           uint8_t* msa, int64_t* cells);

/* ---- unit hooks mirroring the reference's test wrappers (SURVEY 2.1 last row) ---- */

/* runNW (cudapoa_nw.cuh:499): full-band NW on caller-provided graph arrays. Returns alignment length. */
int32_t poa_run_nw_full(const poa_cfg* cfg, const uint8_t* nodes, const int32_t* graph, const int32_t* node_id_to_pos,
                        int32_t graph_count, const uint16_t* incoming_edge_count, const int32_t* incoming_edges,
                        const uint16_t* outgoing_edge_count, const uint8_t* read, int32_t read_length,
                        int32_t* alignment_graph, int32_t* alignment_read);

/* runNWbanded (cudapoa_nw_banded.cuh:608) / runNWbandedTB (cudapoa_nw_tb_banded.cuh:727). */
int32_t poa_run_nw_banded(const poa_cfg* cfg, int32_t adaptive, int32_t traceback, const uint8_t* nodes,
                          const int32_t* graph, const int32_t* node_id_to_pos, int32_t graph_count,
                          const uint16_t* incoming_edge_count, const int32_t* incoming_edges,
                          const uint16_t* outgoing_edge_count, const uint8_t* read, int32_t read_length,
                          int32_t* alignment_graph, int32_t* alignment_read);

/* runTopSort (cudapoa_topsort.cuh:220) */
void poa_run_topsort(int32_t* sorted_poa, int32_t* node_id_to_pos, int32_t node_count,
                     const uint16_t* incoming_edge_count, const int32_t* outgoing_edges,
                     const uint16_t* outgoing_edge_count);

/* addAlignment (cudapoa_add_alignment.cuh:335), non-MSA. Returns status; *node_count updated. */
int32_t poa_run_add_alignment(uint8_t* nodes, int32_t* node_count, int32_t* node_alignments,
                              uint16_t* node_alignment_count, int32_t* incoming_edges,
                              uint16_t* incoming_edge_count, int32_t* outgoing_edges,
                              uint16_t* outgoing_edge_count, uint16_t* incoming_edge_w, int32_t alignment_length,
                              const int32_t* alignment_graph, const uint8_t* read, const int32_t* alignment_read,
                              uint16_t* node_coverage_counts, const int8_t* base_weights,
                              int32_t max_nodes_per_graph);

/* generateConsensusTestHost (cudapoa_generate_consensus.cuh:395) */
void poa_run_consensus(const uint8_t* nodes, int32_t node_count, const int32_t* graph, const int32_t* node_id_to_pos,
                       const int32_t* incoming_edges, const uint16_t* incoming_edge_count,
                       const int32_t* outgoing_edges, const uint16_t* outgoing_edge_count,
                       const uint16_t* incoming_edge_w, uint8_t* consensus, uint16_t* coverage,
                       const uint16_t* node_coverage_counts, const int32_t* node_alignments,
                       const uint16_t* node_alignment_count, int32_t max_consensus_size);

#ifdef __cplusplus
}
#endif
#endif
