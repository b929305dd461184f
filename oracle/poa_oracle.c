/*
 * poa_oracle.c -- CPU restatement of the reference cudapoa algorithm (plain C).
 * TEST INFRASTRUCTURE ONLY: see poa_oracle.h. Cites /root/reference file:line throughout.
 */
#include "poa_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

/* ---- fp32 band placement: cudapoa_nw_banded.cuh:67-78. IEEE fp32, no FMA contraction
 *      (built with -ffp-contract=off); volatile keeps the product a genuine float. ---- */
int32_t poa_band_start_for_row(int32_t row, float gradient, int32_t band_width, int32_t band_shift, int32_t max_column)
{
    volatile float prod    = (float)row * gradient;
    int32_t diagonal_index = (int32_t)prod;
    int32_t start_pos      = diagonal_index - band_shift;
    if (start_pos < 0) start_pos = 0;
    if (max_column < start_pos + band_width)
    {
        start_pos = max_column - band_width + POA_CELLS_PER_THREAD;
        if (start_pos < 0) start_pos = 0;
    }
    start_pos = start_pos - (start_pos % POA_CELLS_PER_THREAD);
    return start_pos;
}

#define SCORE_T int16_t
#define SCORE_MIN INT16_MIN
#define SFX(n) n##_s16
/* Optional observer of the banded traceback (one call per step: the cell left and the cell entered). NULL unless an
 * analysis tool installs one; it never changes a result. */
void (*poa_oracle_step_hook)(int32_t i, int32_t j, int32_t prev_i, int32_t prev_j) = 0;
/* Optional observer of the banded forward pass (one call per DP row: the row, its predecessor count, the distance in rows to
 * its farthest predecessor, its band start). NULL unless an analysis tool installs one; it never changes a result. */
void (*poa_oracle_row_hook)(int32_t row, int32_t pred_count, int32_t max_pred_distance, int32_t band_start) = 0;

#include "poa_nw.inc"
#include "poa_nw_tb.inc"
#undef SCORE_T
#undef SCORE_MIN
#undef SFX

#define SCORE_T int32_t
#define SCORE_MIN INT32_MIN
#define SFX(n) n##_s32
#include "poa_nw.inc"
#include "poa_nw_tb.inc"
#undef SCORE_T
#undef SCORE_MIN
#undef SFX

static int32_t align_i32(int32_t v, int32_t b) { return (v + b - 1) & ~(b - 1); } /* cudautils.hpp:104-111 */

/* cudapoa_limits.hpp:34-59 */
void poa_cfg_select_types(poa_cfg* c)
{
    int32_t upper_bound   = c->max_sequence_size * c->match_score;
    int32_t max_num_nodes = c->max_nodes_per_graph;
    int32_t gm            = c->gap_score > c->mismatch_score ? c->gap_score : c->mismatch_score;
    int32_t lower_bound   = c->max_sequence_size * gm + (max_num_nodes - c->max_sequence_size) * c->gap_score;
    c->score32            = (upper_bound > INT16_MAX || (-lower_bound) > (INT16_MAX + 1));
    c->trace16            = (c->max_banded_pred_distance > INT8_MAX);
}

/* BatchConfig::BatchConfig #1, batch.cu:34-70 */
void poa_cfg_init(poa_cfg* c, int32_t max_seq_sz, int32_t max_seq_per_poa, int32_t band_width, int32_t band_mode,
                  float adaptive_storage_factor, float graph_length_factor, int32_t max_pred_dist, int32_t gap,
                  int32_t mismatch, int32_t match, int32_t output_mask)
{
    memset(c, 0, sizeof(*c));
    c->max_sequence_size        = max_seq_sz;
    c->max_consensus_size       = 2 * max_seq_sz;
    c->alignment_band_width     = align_i32(band_width, POA_MIN_BAND_WIDTH);
    c->max_sequences_per_poa    = max_seq_per_poa;
    c->band_mode                = band_mode;
    c->max_banded_pred_distance = max_pred_dist > 0 ? max_pred_dist : 2 * align_i32(band_width, POA_MIN_BAND_WIDTH);
    c->max_nodes_per_graph      = align_i32((int32_t)(graph_length_factor * (float)max_seq_sz), POA_CELLS_PER_THREAD);
    if (band_mode == POA_FULL_BAND)
        c->matrix_sequence_dimension = align_i32(max_seq_sz, POA_CELLS_PER_THREAD);
    else if (band_mode == POA_STATIC_BAND || band_mode == POA_STATIC_BAND_TB)
        c->matrix_sequence_dimension = align_i32(c->alignment_band_width + POA_BAND_RIGHT_PADDING, POA_CELLS_PER_THREAD);
    else
        c->matrix_sequence_dimension =
            align_i32((int32_t)(adaptive_storage_factor * (float)(c->alignment_band_width + POA_BAND_RIGHT_PADDING)), POA_CELLS_PER_THREAD);
    c->gap_score      = gap;
    c->mismatch_score = mismatch;
    c->match_score    = match;
    c->output_mask    = output_mask;
    poa_cfg_select_types(c);
}

/* ------------------------------------------------------------------------------------------ */
struct poa_workspace
{
    poa_cfg cfg;
    poa_graph g;
    void* scores;       /* ScoreT[max_nodes * msd] (+slack), or the TB score ring */
    size_t scores_elems;
    void* traceback;    /* TraceT[max_nodes * msd] */
    size_t trace_elems;
    int32_t* alignment_graph;
    int32_t* alignment_read;
    int64_t overflow_events;
};

static void* xcalloc(size_t n, size_t sz)
{
    void* p = calloc(n ? n : 1, sz);
    if (!p) abort();
    return p;
}

poa_workspace* poa_workspace_create(const poa_cfg* cfg)
{
    poa_workspace* ws = (poa_workspace*)xcalloc(1, sizeof(*ws));
    ws->cfg           = *cfg;
    size_t n          = (size_t)cfg->max_nodes_per_graph;
    size_t ne         = n * POA_MAX_NODE_EDGES;
    poa_graph* g      = &ws->g;
    g->nodes                     = (uint8_t*)xcalloc(n, 1);
    g->node_alignments           = (int32_t*)xcalloc(n * POA_MAX_NODE_ALIGNMENTS, 4);
    g->node_alignment_count      = (uint16_t*)xcalloc(n, 2);
    g->incoming_edges            = (int32_t*)xcalloc(ne, 4);
    g->incoming_edge_count       = (uint16_t*)xcalloc(n, 2);
    g->outgoing_edges            = (int32_t*)xcalloc(ne, 4);
    g->outgoing_edge_count       = (uint16_t*)xcalloc(n, 2);
    g->incoming_edge_weights     = (uint16_t*)xcalloc(ne, 2);
    g->sorted_poa                = (int32_t*)xcalloc(n, 4);
    g->node_id_to_pos            = (int32_t*)xcalloc(n, 4);
    g->local_incoming_edge_count = (uint16_t*)xcalloc(n, 2);
    g->consensus_scores          = (int32_t*)xcalloc(n + 1, 4);
    g->consensus_predecessors    = (int32_t*)xcalloc(n, 4);
    g->node_marks                = (uint8_t*)xcalloc(n, 1);
    g->check_aligned_nodes       = (uint8_t*)xcalloc(n, 1);
    g->nodes_to_visit            = (int32_t*)xcalloc(n, 4);
    g->node_coverage_counts      = (uint16_t*)xcalloc(n, 2);
    g->sequence_begin_nodes_ids  = (int32_t*)xcalloc((size_t)cfg->max_sequences_per_poa + 1, 4);
    if (cfg->output_mask & 2)
    {
        g->outgoing_edges_coverage       = (uint16_t*)xcalloc(ne * (size_t)cfg->max_sequences_per_poa, 2);
        g->outgoing_edges_coverage_count = (uint16_t*)xcalloc(ne, 2);
        g->node_id_to_msa_pos            = (int32_t*)xcalloc(n, 4);
    }
    size_t width = (size_t)cfg->matrix_sequence_dimension;
    if (cfg->band_mode == POA_FULL_BAND)
    {
        /* per-window width = align4(L+1+4) <= align4(max_seq+5) (cudapoa_batch.cuh:502) */
        width = (size_t)align_i32(cfg->max_sequence_size + 1 + POA_CELLS_PER_THREAD, 4);
    }
    int tb = (cfg->band_mode == POA_STATIC_BAND_TB || cfg->band_mode == POA_ADAPTIVE_BAND_TB);
    if (tb)
    {
        ws->scores_elems = (size_t)cfg->max_banded_pred_distance * width;
        ws->trace_elems  = n * width;
        ws->traceback    = xcalloc(ws->trace_elems + 64, 2);
    }
    else
    {
        ws->scores_elems = n * width;
    }
    ws->scores          = xcalloc(ws->scores_elems + 64, 4);
    ws->alignment_graph = (int32_t*)xcalloc(n * 2 + 16, 4);
    ws->alignment_read  = (int32_t*)xcalloc(n * 2 + 16, 4);
    return ws;
}

void poa_workspace_destroy(poa_workspace* ws)
{
    if (!ws) return;
    poa_graph* g = &ws->g;
    free(g->nodes); free(g->node_alignments); free(g->node_alignment_count); free(g->incoming_edges);
    free(g->incoming_edge_count); free(g->outgoing_edges); free(g->outgoing_edge_count);
    free(g->incoming_edge_weights); free(g->sorted_poa); free(g->node_id_to_pos);
    free(g->local_incoming_edge_count); free(g->consensus_scores); free(g->consensus_predecessors);
    free(g->node_marks); free(g->check_aligned_nodes); free(g->nodes_to_visit); free(g->node_coverage_counts);
    free(g->sequence_begin_nodes_ids); free(g->outgoing_edges_coverage); free(g->outgoing_edges_coverage_count);
    free(g->node_id_to_msa_pos);
    free(ws->scores); free(ws->traceback); free(ws->alignment_graph); free(ws->alignment_read);
    free(ws);
}

poa_graph* poa_workspace_graph(poa_workspace* ws) { return &ws->g; }
int64_t poa_workspace_overflow_events(const poa_workspace* ws) { return ws->overflow_events; }

/* ------------------------------------------------------------------------------------------
 * addAlignmentToGraph: cudapoa_add_alignment.cuh:65-285
 * ------------------------------------------------------------------------------------------ */
static int32_t add_alignment_to_graph(int32_t* new_node_count, uint8_t* nodes, int32_t node_count,
                                      int32_t* node_alignments, uint16_t* node_alignment_count,
                                      int32_t* incoming_edges, uint16_t* incoming_edge_count,
                                      int32_t* outgoing_edges, uint16_t* outgoing_edge_count,
                                      uint16_t* incoming_edge_w, int32_t alignment_length,
                                      const int32_t* alignment_graph, const uint8_t* read,
                                      const int32_t* alignment_read, uint16_t* node_coverage_counts,
                                      const int8_t* base_weights, int msa, int32_t* sequence_begin_nodes_ids,
                                      uint16_t* outgoing_edges_coverage, uint16_t* outgoing_edges_coverage_count,
                                      uint16_t s, uint32_t max_sequences_per_poa, uint32_t max_limit_nodes_per_window)
{
    int32_t head_node_id = -1;
    int32_t curr_node_id = -1;
    uint16_t prev_weight = 0;

    for (int32_t pos = alignment_length - 1; pos >= 0; pos--)
    {
        int32_t read_pos = alignment_read[pos];
        if (read_pos != -1)
        {
            int8_t NODE_WEIGHT    = base_weights[read_pos];
            uint8_t read_base     = read[read_pos];
            int32_t graph_node_id = alignment_graph[pos];
            if (graph_node_id == -1)
            {
                curr_node_id = node_count++;
                if ((uint32_t)node_count >= max_limit_nodes_per_window) return POA_NODE_COUNT_EXCEEDED; /* :120-123 */
                nodes[curr_node_id]                = read_base;
                outgoing_edge_count[curr_node_id]  = 0;
                incoming_edge_count[curr_node_id]  = 0;
                node_alignment_count[curr_node_id] = 0;
                node_coverage_counts[curr_node_id] = 0;
            }
            else
            {
                uint8_t graph_base = nodes[graph_node_id];
                if (graph_base == read_base)
                {
                    curr_node_id = graph_node_id;
                }
                else
                {
                    uint16_t num_aligned_node = node_alignment_count[graph_node_id];
                    int32_t aligned_node_id   = -1;
                    for (int32_t n = 0; n < num_aligned_node; n++)
                    {
                        int32_t aid = node_alignments[graph_node_id * POA_MAX_NODE_ALIGNMENTS + n];
                        if (nodes[aid] == read_base)
                        {
                            aligned_node_id = aid;
                            break;
                        }
                    }
                    if (aligned_node_id != -1)
                    {
                        curr_node_id = aligned_node_id;
                    }
                    else
                    {
                        curr_node_id = node_count++;
                        if ((uint32_t)node_count >= max_limit_nodes_per_window) return POA_NODE_COUNT_EXCEEDED; /* :176-179 */
                        nodes[curr_node_id]                = read_base;
                        outgoing_edge_count[curr_node_id]  = 0;
                        incoming_edge_count[curr_node_id]  = 0;
                        node_alignment_count[curr_node_id] = 0;
                        node_coverage_counts[curr_node_id] = 0;
                        int32_t new_node_alignments        = 0;
                        for (int32_t n = 0; n < num_aligned_node; n++)
                        {
                            int32_t aid        = node_alignments[graph_node_id * POA_MAX_NODE_ALIGNMENTS + n];
                            uint16_t aid_count = node_alignment_count[aid];
                            node_alignments[aid * POA_MAX_NODE_ALIGNMENTS + aid_count]                    = curr_node_id;
                            node_alignment_count[aid]                                                     = aid_count + 1;
                            node_alignments[curr_node_id * POA_MAX_NODE_ALIGNMENTS + new_node_alignments] = aid;
                            new_node_alignments++;
                        }
                        node_alignments[graph_node_id * POA_MAX_NODE_ALIGNMENTS + num_aligned_node] = curr_node_id;
                        node_alignment_count[graph_node_id]                                         = num_aligned_node + 1;
                        node_alignments[curr_node_id * POA_MAX_NODE_ALIGNMENTS + new_node_alignments] = graph_node_id;
                        new_node_alignments++;
                        node_alignment_count[curr_node_id] = (uint16_t)new_node_alignments;
                    }
                }
            }

            if (msa && (read_pos == 0)) *sequence_begin_nodes_ids = curr_node_id; /* :215-219 */

            if (head_node_id != -1)
            {
                int edge_exists   = 0;
                uint16_t in_count = incoming_edge_count[curr_node_id];
                for (int32_t e = 0; e < in_count; e++)
                {
                    if (incoming_edges[curr_node_id * POA_MAX_NODE_EDGES + e] == head_node_id)
                    {
                        edge_exists = 1;
                        incoming_edge_w[curr_node_id * POA_MAX_NODE_EDGES + e] =
                            (uint16_t)(incoming_edge_w[curr_node_id * POA_MAX_NODE_EDGES + e] + (prev_weight + NODE_WEIGHT));
                    }
                }
                if (!edge_exists)
                {
                    incoming_edges[curr_node_id * POA_MAX_NODE_EDGES + in_count]  = head_node_id;
                    incoming_edge_w[curr_node_id * POA_MAX_NODE_EDGES + in_count] = (uint16_t)(prev_weight + NODE_WEIGHT);
                    incoming_edge_count[curr_node_id]                             = in_count + 1;
                    uint16_t out_count                                            = outgoing_edge_count[head_node_id];
                    outgoing_edges[head_node_id * POA_MAX_NODE_EDGES + out_count] = curr_node_id;
                    if (msa)
                    {
                        outgoing_edges_coverage_count[head_node_id * POA_MAX_NODE_EDGES + out_count] = 1;
                        outgoing_edges_coverage[(size_t)(head_node_id * POA_MAX_NODE_EDGES + out_count) * max_sequences_per_poa] = s;
                    }
                    outgoing_edge_count[head_node_id] = out_count + 1;
                    if (out_count + 1 >= POA_MAX_NODE_EDGES || in_count + 1 >= POA_MAX_NODE_EDGES)
                        return POA_EDGE_COUNT_EXCEEDED; /* :251-255 */
                }
                else if (msa)
                {
                    uint16_t out_count = outgoing_edge_count[head_node_id];
                    for (int32_t e = 0; e < out_count; e++)
                    {
                        if (outgoing_edges[head_node_id * POA_MAX_NODE_EDGES + e] == curr_node_id)
                        {
                            uint16_t cc = outgoing_edges_coverage_count[head_node_id * POA_MAX_NODE_EDGES + e];
                            outgoing_edges_coverage[(size_t)(head_node_id * POA_MAX_NODE_EDGES + e) * max_sequences_per_poa + cc] = s;
                            outgoing_edges_coverage_count[head_node_id * POA_MAX_NODE_EDGES + e] = cc + 1;
                            break;
                        }
                    }
                }
            }
            head_node_id = curr_node_id;
            node_coverage_counts[head_node_id]++;
            prev_weight = (uint16_t)NODE_WEIGHT;
        }
    }
    *new_node_count = node_count;
    return POA_SUCCESS;
}

/* ------------------------------------------------------------------------------------------
 * topologicalSortDeviceUtil: cudapoa_topsort.cuh:45-97
 * ------------------------------------------------------------------------------------------ */
static void topsort_kahn(int32_t* sorted_poa, int32_t* sorted_poa_node_map, int32_t node_count,
                         const uint16_t* incoming_edge_count, const int32_t* outgoing_edges,
                         const uint16_t* outgoing_edge_count, uint16_t* local_incoming_edge_count)
{
    int32_t sorted_poa_position = 0;
    for (int32_t n = 0; n < node_count; n++)
    {
        local_incoming_edge_count[n] = incoming_edge_count[n];
        if (local_incoming_edge_count[n] == 0)
        {
            sorted_poa_node_map[n]            = sorted_poa_position;
            sorted_poa[sorted_poa_position++] = n;
        }
    }
    for (int32_t n = 0; n < sorted_poa_position; n++)
    {
        int32_t node = sorted_poa[n];
        for (int32_t edge = 0; edge < outgoing_edge_count[node]; edge++)
        {
            int32_t out_node       = outgoing_edges[node * POA_MAX_NODE_EDGES + edge];
            uint16_t in_node_count = local_incoming_edge_count[out_node];
            if (--in_node_count == 0)
            {
                sorted_poa_node_map[out_node]     = sorted_poa_position;
                sorted_poa[sorted_poa_position++] = out_node;
            }
            local_incoming_edge_count[out_node] = in_node_count;
        }
    }
}

#include <stdio.h>
#include "topsort_incr_model.inc"
#include "topsort_incr_cnt8_model.inc"
static int32_t* tsm_sorted = NULL;
static int32_t* tsm_map    = NULL;
static uint16_t* tsm_meta  = NULL;
static int32_t tsm_cap = 0, tsm_nold = 0;
static void tsm_begin_window(int32_t max_nodes, int32_t len0)
{
    if (max_nodes > tsm_cap)
    {
        free(tsm_sorted); free(tsm_map); free(tsm_meta);
        tsm_sorted = (int32_t*)malloc(sizeof(int32_t) * (size_t)max_nodes);
        tsm_map    = (int32_t*)malloc(sizeof(int32_t) * (size_t)max_nodes);
        tsm_meta   = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)max_nodes);
        tsm_cap    = max_nodes;
    }
    for (int32_t n = 0; n < len0; n++) /* backbone chain: queue length 1 everywhere */
    {
        tsm_sorted[n] = n;
        tsm_map[n]    = n;
        tsm_meta[n]   = (uint16_t)(1 | ((n < len0 - 1 ? 1 : 0) << 4) | ((n > 0 ? 1 : 0) << 10));
    }
    tsm_nold = len0;
}
static int32_t* tsc_sorted = NULL;
static int32_t* tsc_map    = NULL;
static uint16_t* tsc_meta  = NULL;
static int32_t tsc_cap = 0, tsc_nold = 0;
static void tsc_begin_window(int32_t max_nodes, int32_t len0)
{
    if (max_nodes > tsc_cap)
    {
        free(tsc_sorted); free(tsc_map); free(tsc_meta);
        tsc_sorted = (int32_t*)malloc(sizeof(int32_t) * (size_t)max_nodes);
        tsc_map    = (int32_t*)malloc(sizeof(int32_t) * (size_t)max_nodes);
        tsc_meta   = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)max_nodes);
        tsc_cap    = max_nodes;
    }
    for (int32_t n = 0; n < len0; n++) /* backbone chain: queue length 1 everywhere */
    {
        tsc_sorted[n] = n;
        tsc_map[n]    = n;
        tsc_meta[n]   = (uint16_t)(1 | ((n < len0 - 1 ? 1 : 0) << 4) | ((n > 0 ? 1 : 0) << 10));
    }
    tsc_nold = len0;
}

/* raconTopologicalSortDeviceUtil: cudapoa_topsort.cuh:103-197 */
static void topsort_racon(int32_t* sorted_poa, int32_t* sorted_poa_node_map, int32_t node_count,
                          const uint16_t* incoming_edge_count, const int32_t* incoming_edges,
                          const uint16_t* aligned_node_count, const int32_t* aligned_nodes, uint8_t* node_marks,
                          uint8_t* check_aligned_nodes, int32_t* nodes_to_visit, int32_t max_nodes_per_graph)
{
    int32_t node_idx       = -1;
    int32_t sorted_poa_idx = 0;
    for (int32_t i = 0; i < max_nodes_per_graph; i++)
    {
        node_marks[i]          = 0;
        check_aligned_nodes[i] = 1;
    }
    for (int32_t i = 0; i < node_count; i++)
    {
        if (node_marks[i] != 0) continue;
        node_idx++;
        nodes_to_visit[node_idx] = i;
        while (node_idx != -1)
        {
            int32_t node_id = nodes_to_visit[node_idx];
            int valid       = 1;
            if (node_marks[node_id] != 2)
            {
                for (int32_t e = 0; e < incoming_edge_count[node_id]; e++)
                {
                    int32_t begin_node_id = incoming_edges[node_id * POA_MAX_NODE_EDGES + e];
                    if (node_marks[begin_node_id] != 2)
                    {
                        node_idx++;
                        nodes_to_visit[node_idx] = begin_node_id;
                        valid                    = 0;
                    }
                }
                if (check_aligned_nodes[node_id])
                {
                    for (int32_t a = 0; a < aligned_node_count[node_id]; a++)
                    {
                        int32_t aid = aligned_nodes[node_id * POA_MAX_NODE_ALIGNMENTS + a];
                        if (node_marks[aid] != 2)
                        {
                            node_idx++;
                            nodes_to_visit[node_idx] = aid;
                            check_aligned_nodes[aid] = 0;
                            valid                    = 0;
                        }
                    }
                }
                if (valid)
                {
                    node_marks[node_id] = 2;
                    if (check_aligned_nodes[node_id])
                    {
                        sorted_poa[sorted_poa_idx]   = node_id;
                        sorted_poa_node_map[node_id] = sorted_poa_idx;
                        sorted_poa_idx++;
                        for (int32_t a = 0; a < aligned_node_count[node_id]; a++)
                        {
                            int32_t aid                = aligned_nodes[node_id * POA_MAX_NODE_ALIGNMENTS + a];
                            sorted_poa[sorted_poa_idx] = aid;
                            sorted_poa_node_map[aid]   = sorted_poa_idx;
                            sorted_poa_idx++;
                        }
                    }
                }
                else
                {
                    node_marks[node_id] = 1;
                }
            }
            if (valid) node_idx--;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * branchCompletion / generateConsensus: cudapoa_generate_consensus.cuh:35-119,141-283
 * scores has one guard element in front (scores[-1] is indexable) because the reference evaluates
 * scores[predecessors[node]] with predecessors == -1 behind a short-circuit that never fires for
 * edge weights >= 0; the guard keeps the restatement literal without UB.
 * ------------------------------------------------------------------------------------------ */
static int32_t branch_completion(int32_t max_score_id_pos, int32_t node_count, const int32_t* graph,
                                 const int32_t* incoming_edges, const uint16_t* incoming_edge_count,
                                 const int32_t* outgoing_edges, const uint16_t* outgoing_edge_count,
                                 const uint16_t* incoming_edge_w, int32_t* scores, int32_t* predecessors)
{
    int32_t node_id    = graph[max_score_id_pos];
    uint16_t out_edges = outgoing_edge_count[node_id];
    for (int32_t oe = 0; oe < out_edges; oe++)
    {
        int32_t out_node_id        = outgoing_edges[node_id * POA_MAX_NODE_EDGES + oe];
        uint16_t out_node_in_edges = incoming_edge_count[out_node_id];
        for (int32_t ie = 0; ie < out_node_in_edges; ie++)
        {
            int32_t id = incoming_edges[out_node_id * POA_MAX_NODE_EDGES + ie];
            if (id != node_id) scores[id] = -1;
        }
    }
    int32_t max_score    = 0;
    int32_t max_score_id = 0;
    for (int32_t graph_pos = max_score_id_pos + 1; graph_pos < node_count; graph_pos++)
    {
        node_id               = graph[graph_pos];
        predecessors[node_id] = -1;
        int32_t score_node_id = -1;
        uint16_t in_edges     = incoming_edge_count[node_id];
        for (int32_t e = 0; e < in_edges; e++)
        {
            int32_t begin_node_id = incoming_edges[node_id * POA_MAX_NODE_EDGES + e];
            if (scores[begin_node_id] == -1) continue;
            int32_t edge_w = (int32_t)incoming_edge_w[node_id * POA_MAX_NODE_EDGES + e];
            if (score_node_id < edge_w ||
                (score_node_id == edge_w && scores[predecessors[node_id]] <= scores[begin_node_id]))
            {
                score_node_id         = edge_w;
                predecessors[node_id] = begin_node_id;
            }
        }
        if (predecessors[node_id] != -1) score_node_id += scores[predecessors[node_id]];
        if (max_score <= score_node_id)
        {
            max_score    = score_node_id;
            max_score_id = node_id;
        }
        scores[node_id] = score_node_id;
    }
    return max_score_id;
}

static void generate_consensus(const uint8_t* nodes, int32_t node_count, const int32_t* graph,
                               const int32_t* node_id_to_pos, const int32_t* incoming_edges,
                               const uint16_t* incoming_edge_count, const int32_t* outgoing_edges,
                               const uint16_t* outgoing_edge_count, const uint16_t* incoming_edge_w,
                               int32_t* predecessors, int32_t* scores_base, uint8_t* consensus, uint16_t* coverage,
                               const uint16_t* node_coverage_counts, const int32_t* node_alignments,
                               const uint16_t* node_alignment_count, int32_t max_limit_consensus_size)
{
    int32_t* scores = scores_base + 1; /* guard element at [-1] */
    scores[-1]      = -1;
    for (int32_t i = 0; i < node_count; i++)
    {
        predecessors[i] = -1;
        scores[i]       = -1;
    }
    int32_t max_score_id = 0;
    int32_t max_score    = -1;
    for (int32_t graph_pos = 0; graph_pos < node_count; graph_pos++)
    {
        int32_t node_id       = graph[graph_pos];
        uint16_t in_edges     = incoming_edge_count[node_id];
        int32_t score_node_id = scores[node_id];
        for (int32_t e = 0; e < in_edges; e++)
        {
            int32_t edge_w        = (int32_t)incoming_edge_w[node_id * POA_MAX_NODE_EDGES + e];
            int32_t begin_node_id = incoming_edges[node_id * POA_MAX_NODE_EDGES + e];
            if (score_node_id < edge_w ||
                (score_node_id == edge_w && scores[predecessors[node_id]] <= scores[begin_node_id]))
            {
                score_node_id         = edge_w;
                predecessors[node_id] = begin_node_id;
            }
        }
        if (predecessors[node_id] != -1) score_node_id += scores[predecessors[node_id]];
        if (max_score <= score_node_id)
        {
            max_score_id = node_id;
            max_score    = score_node_id;
        }
        scores[node_id] = score_node_id;
    }

    int32_t loop_count = 0;
    if (outgoing_edge_count[max_score_id] != 0)
    {
        while (outgoing_edge_count[max_score_id] != 0 && loop_count < node_count)
        {
            max_score_id = branch_completion(node_id_to_pos[max_score_id], node_count, graph, incoming_edges,
                                             incoming_edge_count, outgoing_edges, outgoing_edge_count, incoming_edge_w,
                                             scores, predecessors);
            loop_count++;
        }
    }
    if (loop_count >= node_count)
    {
        consensus[0] = POA_KERNEL_ERROR;
        consensus[1] = (uint8_t)POA_LOOP_COUNT_EXCEEDED;
        return;
    }

    int32_t consensus_pos   = 0;
    int32_t consensus_count = 0;
    while (predecessors[max_score_id] != -1)
    {
        consensus[consensus_pos] = nodes[max_score_id];
        uint16_t cov             = node_coverage_counts[max_score_id];
        for (int32_t a = 0; a < node_alignment_count[max_score_id]; a++)
            cov = (uint16_t)(cov + node_coverage_counts[node_alignments[max_score_id * POA_MAX_NODE_ALIGNMENTS + a]]);
        coverage[consensus_pos] = cov;
        max_score_id            = predecessors[max_score_id];
        consensus_pos           = (consensus_pos + 1) < (max_limit_consensus_size - 1) ? (consensus_pos + 1) : (max_limit_consensus_size - 1);
        consensus_count++;
    }
    consensus[consensus_pos] = nodes[max_score_id];
    uint16_t cov             = node_coverage_counts[max_score_id];
    for (int32_t a = 0; a < node_alignment_count[max_score_id]; a++)
        cov = (uint16_t)(cov + node_coverage_counts[node_alignments[max_score_id * POA_MAX_NODE_ALIGNMENTS + a]]);
    coverage[consensus_pos] = cov;
    if (consensus_count >= (max_limit_consensus_size - 1))
    {
        consensus[0] = POA_KERNEL_ERROR;
        consensus[1] = (uint8_t)POA_EXCEEDED_MAXIMUM_SEQUENCE_SIZE;
        return;
    }
    consensus_pos++;
    consensus[consensus_pos] = '\0';
}

#include "poa_msa.inc"

/* ------------------------------------------------------------------------------------------
 * Window driver: generatePOAKernel, cudapoa_kernels.cuh:200-541 (+ the consensus / MSA kernels)
 * ------------------------------------------------------------------------------------------ */
static int32_t window_error(uint8_t* consensus, int32_t status)
{
    consensus[0] = POA_KERNEL_ERROR;
    consensus[1] = (uint8_t)status;
    return status;
}

int32_t poa_process_window(poa_workspace* ws, const uint8_t* seqs, const int8_t* weights, int32_t* seq_lens,
                           int32_t num_seqs, size_t seq_buf_bytes, uint8_t* consensus, uint16_t* coverage,
                           uint8_t* msa_out, int64_t* cells)
{
    const poa_cfg* c = &ws->cfg;
    poa_graph* g     = &ws->g;
    const int msa    = (c->output_mask & 2) != 0;
    const int E      = POA_MAX_NODE_EDGES;

    const uint8_t* sequence    = seqs;
    const int8_t* base_weights = weights;
    size_t consumed            = 0;

    /* backbone :200-238 */
    g->nodes[0]                 = sequence[0];
    g->sorted_poa[0]            = 0;
    g->incoming_edge_count[0]   = 0;
    g->node_alignment_count[0]  = 0;
    g->node_id_to_pos[0]        = 0;
    g->outgoing_edge_count[seq_lens[0] - 1] = 0;
    g->incoming_edge_weights[0] = (uint16_t)base_weights[0];
    g->node_coverage_counts[0]  = 1;
    if (msa) g->sequence_begin_nodes_ids[0] = 0;
    for (int32_t n = 1; n < seq_lens[0]; n++)
    {
        g->nodes[n]                     = sequence[n];
        g->sorted_poa[n]                = n;
        g->outgoing_edges[(n - 1) * E]  = n;
        g->outgoing_edge_count[n - 1]   = 1;
        g->incoming_edges[n * E]        = n - 1;
        g->incoming_edge_weights[n * E] = (uint16_t)(base_weights[n - 1] + base_weights[n]);
        g->incoming_edge_count[n]       = 1;
        g->node_alignment_count[n]      = 0;
        g->node_id_to_pos[n]            = n;
        g->node_coverage_counts[n]      = 1;
        if (msa)
        {
            g->outgoing_edges_coverage[(size_t)(n - 1) * E * c->max_sequences_per_poa] = 0;
            g->outgoing_edges_coverage_count[(n - 1) * E]                              = 1;
        }
    }
    consensus[0] = 0;
    if (tsm_enabled && !c->spoa_accurate && c->max_nodes_per_graph <= 4095) tsm_begin_window(c->max_nodes_per_graph, seq_lens[0]);
    if (tsc_enabled && !c->spoa_accurate && c->max_nodes_per_graph <= 0xfffff) tsc_begin_window(c->max_nodes_per_graph, seq_lens[0]);

    float banded_buffer_size = (float)c->max_nodes_per_graph * (float)c->matrix_sequence_dimension; /* :149-162 */
    int32_t scores_width     = 0; /* full band: window_details.scores_width, cudapoa_batch.cuh:502-507 */
    for (int32_t s = 0; s < num_seqs; s++)
    {
        int32_t w = align_i32(seq_lens[s] + 1 + POA_CELLS_PER_THREAD, 4);
        if (w > scores_width) scores_width = w;
    }

    for (int32_t s = 1; s < num_seqs; s++)
    {
        int32_t seq_len = seq_lens[s];
        int32_t adv     = align_i32(s == 1 ? seq_lens[0] : seq_lens[s - 1], 4);
        /* NB: seq_lens[0] is overwritten with the node count after the first merge (:506), but the pointer
           advance for s==1 happens before that; for s>=2 it uses seq_lens[s-1] (:248-249). */
        sequence += adv;
        base_weights += adv;
        consumed += (size_t)adv;
        size_t read_avail = seq_buf_bytes > consumed ? seq_buf_bytes - consumed : 0;

        if (seq_lens[0] >= c->max_nodes_per_graph) return window_error(consensus, POA_NODE_COUNT_EXCEEDED); /* :253-265 */

        int32_t graph_count = seq_lens[0];
        int32_t alignment_length;
        int32_t bm = c->band_mode;

#define CALL_BANDED(ADAPT, RERUN)                                                                                      \
    (c->score32 ? nw_banded_s32(ADAPT, g->nodes, g->sorted_poa, g->node_id_to_pos, graph_count,                       \
                                g->incoming_edge_count, g->incoming_edges, g->outgoing_edge_count, sequence,           \
                                read_avail, seq_len, (int32_t*)ws->scores, ws->scores_elems, banded_buffer_size,       \
                                ws->alignment_graph, ws->alignment_read, c->alignment_band_width, c->gap_score,        \
                                c->mismatch_score, c->match_score, RERUN, &ws->overflow_events, cells)                 \
                : nw_banded_s16(ADAPT, g->nodes, g->sorted_poa, g->node_id_to_pos, graph_count,                       \
                                g->incoming_edge_count, g->incoming_edges, g->outgoing_edge_count, sequence,           \
                                read_avail, seq_len, (int16_t*)ws->scores, ws->scores_elems, banded_buffer_size,       \
                                ws->alignment_graph, ws->alignment_read, c->alignment_band_width, c->gap_score,        \
                                c->mismatch_score, c->match_score, RERUN, &ws->overflow_events, cells))
#define CALL_TB(ADAPT, RERUN)                                                                                          \
    (c->score32 ? nw_banded_tb_s32(ADAPT, c->trace16, g->nodes, g->sorted_poa, g->node_id_to_pos, graph_count,        \
                                   g->incoming_edge_count, g->incoming_edges, g->outgoing_edge_count, sequence,        \
                                   read_avail, seq_len, (int32_t*)ws->scores, ws->scores_elems,                  \
                                   (int16_t*)ws->traceback, ws->trace_elems, banded_buffer_size,                  \
                                   ws->alignment_graph, ws->alignment_read, c->alignment_band_width,                   \
                                   c->max_banded_pred_distance, c->gap_score, c->mismatch_score, c->match_score,       \
                                   RERUN, &ws->overflow_events, cells)                                                 \
                : nw_banded_tb_s16(ADAPT, c->trace16, g->nodes, g->sorted_poa, g->node_id_to_pos, graph_count,        \
                                   g->incoming_edge_count, g->incoming_edges, g->outgoing_edge_count, sequence,        \
                                   read_avail, seq_len, (int16_t*)ws->scores, ws->scores_elems,                  \
                                   (int16_t*)ws->traceback, ws->trace_elems, banded_buffer_size,                  \
                                   ws->alignment_graph, ws->alignment_read, c->alignment_band_width,                   \
                                   c->max_banded_pred_distance, c->gap_score, c->mismatch_score, c->match_score,       \
                                   RERUN, &ws->overflow_events, cells))

        if (bm == POA_ADAPTIVE_BAND_TB && c->alignment_band_width < POA_MAX_ADAPTIVE_BAND) /* :273-320 */
        {
            alignment_length = CALL_TB(1, 0);
            if (alignment_length == POA_SHIFT_LEFT || alignment_length == POA_SHIFT_RIGHT)
                alignment_length = CALL_TB(1, alignment_length);
        }
        else if (bm == POA_STATIC_BAND_TB || bm == POA_ADAPTIVE_BAND_TB) /* :323-346 */
        {
            alignment_length = CALL_TB(0, 0);
        }
        else if (bm == POA_ADAPTIVE_BAND && c->alignment_band_width < POA_MAX_ADAPTIVE_BAND) /* :351-396 */
        {
            alignment_length = CALL_BANDED(1, 0);
            if (alignment_length == POA_SHIFT_LEFT || alignment_length == POA_SHIFT_RIGHT)
                alignment_length = CALL_BANDED(1, alignment_length);
        }
        else if (bm == POA_STATIC_BAND || bm == POA_ADAPTIVE_BAND) /* :399-419 */
        {
            alignment_length = CALL_BANDED(0, 0);
        }
        else /* full band :422-441 */
        {
            if (cells) *cells += (int64_t)graph_count * seq_len;
            alignment_length =
                c->score32 ? nw_full_s32(g->nodes, g->sorted_poa, g->node_id_to_pos, graph_count, g->incoming_edge_count,
                                         g->incoming_edges, g->outgoing_edge_count, sequence, read_avail, seq_len,
                                         (int32_t*)ws->scores, scores_width, ws->alignment_graph, ws->alignment_read,
                                         c->gap_score, c->mismatch_score, c->match_score, &ws->overflow_events)
                           : nw_full_s16(g->nodes, g->sorted_poa, g->node_id_to_pos, graph_count, g->incoming_edge_count,
                                         g->incoming_edges, g->outgoing_edge_count, sequence, read_avail, seq_len,
                                         (int16_t*)ws->scores, scores_width, ws->alignment_graph, ws->alignment_read,
                                         c->gap_score, c->mismatch_score, c->match_score, &ws->overflow_events);
        }
#undef CALL_BANDED
#undef CALL_TB

        /* :444-473 */
        if (alignment_length == POA_NW_LOOP_FAILED) return window_error(consensus, POA_LOOP_COUNT_EXCEEDED);
        if (alignment_length == POA_NW_ADAPTIVE_STORAGE_FAILED) return window_error(consensus, POA_EXCEEDED_ADAPTIVE_BANDED_MATRIX_SIZE);
        if ((bm == POA_STATIC_BAND_TB || bm == POA_ADAPTIVE_BAND_TB) && alignment_length == POA_NW_TRACEBACK_BUFFER_FAILED)
            return window_error(consensus, POA_EXCEEDED_MAXIMUM_PREDECESSOR_DISTANCE);

        /* :475-531 */
        int32_t new_node_count = 0;
        int32_t err = add_alignment_to_graph(&new_node_count, g->nodes, seq_lens[0], g->node_alignments,
                                             g->node_alignment_count, g->incoming_edges, g->incoming_edge_count,
                                             g->outgoing_edges, g->outgoing_edge_count, g->incoming_edge_weights,
                                             alignment_length, ws->alignment_graph, sequence, ws->alignment_read,
                                             g->node_coverage_counts, base_weights, msa, g->sequence_begin_nodes_ids + s,
                                             g->outgoing_edges_coverage, g->outgoing_edges_coverage_count, (uint16_t)s,
                                             (uint32_t)c->max_sequences_per_poa, (uint32_t)c->max_nodes_per_graph);
        if (err != 0) return window_error(consensus, err);
        seq_lens[0] = new_node_count;
        if (c->spoa_accurate)
            topsort_racon(g->sorted_poa, g->node_id_to_pos, new_node_count, g->incoming_edge_count, g->incoming_edges,
                          g->node_alignment_count, g->node_alignments, g->node_marks, g->check_aligned_nodes,
                          g->nodes_to_visit, (int32_t)(uint16_t)c->max_nodes_per_graph /* (uint16_t) cast :519 */);
        else
        {
            topsort_kahn(g->sorted_poa, g->node_id_to_pos, new_node_count, g->incoming_edge_count, g->outgoing_edges,
                         g->outgoing_edge_count, g->local_incoming_edge_count);
            if (tsm_enabled && c->max_nodes_per_graph <= 4095) /* model of the kernel's incremental order, see the .inc */
            {
                topsort_incr_model(tsm_sorted, tsm_map, tsm_nold, new_node_count, g->incoming_edge_count,
                                   g->outgoing_edges, g->outgoing_edge_count, tsm_meta);
                tsm_nold = new_node_count;
                for (int32_t n = 0; n < new_node_count; n++)
                    if (tsm_sorted[n] != g->sorted_poa[n] || tsm_map[n] != g->node_id_to_pos[n]) { tsm_stats[TSM_MISMATCH]++; break; }
            }
            if (tsc_enabled && c->max_nodes_per_graph <= 0xfffff) /* model of the long-read kernel's order with its state in LDS */
            {
                if (topsort_incr_cnt8_model(tsc_sorted, tsc_map, tsc_nold, new_node_count, g->incoming_edge_count, g->outgoing_edges,
                                            g->outgoing_edge_count, tsc_meta))
                {
                    for (int32_t n = 0; n < new_node_count; n++)
                        if (tsc_sorted[n] != g->sorted_poa[n] || tsc_map[n] != g->node_id_to_pos[n]) { tsc_stats[TSC_MISMATCH]++; break; }
                }
                else /* gave up (queue ring): the kernel re-runs the HBM routine; the model restarts from the plain order */
                    for (int32_t n = 0; n < new_node_count; n++)
                    {
                        tsc_sorted[n] = g->sorted_poa[n];
                        tsc_map[n]    = g->node_id_to_pos[n];
                        tsc_meta[n]   = (uint16_t)(15 | ((uint32_t)g->outgoing_edge_count[n] << 4) | ((uint32_t)g->incoming_edge_count[n] << 10));
                    }
                tsc_nold = new_node_count;
            }
        }
    }

    /* output kernels: cudapoa_kernels.cuh:1023-1075 -- msa bit set => ONLY the MSA kernel runs */
    if (msa)
    {
        int32_t st = generate_msa_window(ws, seq_lens, num_seqs, consensus, msa_out);
        return st;
    }
    generate_consensus(g->nodes, seq_lens[0], g->sorted_poa, g->node_id_to_pos, g->incoming_edges,
                       g->incoming_edge_count, g->outgoing_edges, g->outgoing_edge_count, g->incoming_edge_weights,
                       g->consensus_predecessors, g->consensus_scores, consensus, coverage, g->node_coverage_counts,
                       g->node_alignments, g->node_alignment_count, c->max_consensus_size);
    return consensus[0] == POA_KERNEL_ERROR ? consensus[1] : POA_SUCCESS;
}

/* ------------------------------------------------------------------------------------------
 * Unit hooks (reference test wrappers)
 * ------------------------------------------------------------------------------------------ */
int32_t poa_run_nw_full(const poa_cfg* cfg, const uint8_t* nodes, const int32_t* graph, const int32_t* node_id_to_pos,
                        int32_t graph_count, const uint16_t* incoming_edge_count, const int32_t* incoming_edges,
                        const uint16_t* outgoing_edge_count, const uint8_t* read, int32_t read_length,
                        int32_t* alignment_graph, int32_t* alignment_read)
{
    /* runNW, cudapoa_nw.cuh:499-540: int16 scores, width = matrix_sequence_dimension (BatchConfig full band) */
    int64_t ovf          = 0;
    int32_t scores_width = align_i32(read_length + 1 + POA_CELLS_PER_THREAD, 4);
    if (cfg->matrix_sequence_dimension > scores_width) scores_width = cfg->matrix_sequence_dimension;
    size_t elems    = (size_t)(graph_count + 2) * (size_t)scores_width + 64;
    int16_t* scores = (int16_t*)xcalloc(elems, sizeof(int16_t));
    int32_t r = nw_full_s16(nodes, graph, node_id_to_pos, graph_count, incoming_edge_count, incoming_edges,
                            outgoing_edge_count, read, (size_t)read_length, read_length, scores, scores_width,
                            alignment_graph, alignment_read, cfg->gap_score, cfg->mismatch_score, cfg->match_score, &ovf);
    free(scores);
    return r;
}

int32_t poa_run_nw_banded(const poa_cfg* cfg, int32_t adaptive, int32_t traceback, const uint8_t* nodes,
                          const int32_t* graph, const int32_t* node_id_to_pos, int32_t graph_count,
                          const uint16_t* incoming_edge_count, const int32_t* incoming_edges,
                          const uint16_t* outgoing_edge_count, const uint8_t* read, int32_t read_length,
                          int32_t* alignment_graph, int32_t* alignment_read)
{
    /* runNWbanded cudapoa_nw_banded.cuh:560-680 / runNWbandedTB cudapoa_nw_tb_banded.cuh:646-779:
       int16 scores, buffer = max_nodes_per_graph * matrix_sequence_dimension, rerun = 0, no second pass. */
    int64_t ovf       = 0;
    float buffer_size = (float)cfg->max_nodes_per_graph * (float)cfg->matrix_sequence_dimension;
    size_t elems      = (size_t)cfg->max_nodes_per_graph * (size_t)cfg->matrix_sequence_dimension;
    int32_t r;
    if (!traceback)
    {
        int16_t* scores = (int16_t*)xcalloc(elems + 64, sizeof(int16_t));
        r = nw_banded_s16(adaptive, nodes, graph, node_id_to_pos, graph_count, incoming_edge_count, incoming_edges,
                          outgoing_edge_count, read, (size_t)read_length, read_length, scores, elems, buffer_size,
                          alignment_graph, alignment_read, cfg->alignment_band_width, cfg->gap_score,
                          cfg->mismatch_score, cfg->match_score, 0, &ovf, NULL);
        free(scores);
    }
    else
    {
        size_t selems   = (size_t)cfg->max_banded_pred_distance * (size_t)cfg->matrix_sequence_dimension;
        int16_t* scores = (int16_t*)xcalloc(selems + 64, sizeof(int16_t));
        int16_t* trace  = (int16_t*)xcalloc(elems + 64, 2);
        r = nw_banded_tb_s16(adaptive, cfg->trace16, nodes, graph, node_id_to_pos, graph_count, incoming_edge_count,
                             incoming_edges, outgoing_edge_count, read, (size_t)read_length, read_length, scores, selems,
                             trace, elems, buffer_size, alignment_graph, alignment_read, cfg->alignment_band_width,
                             cfg->max_banded_pred_distance, cfg->gap_score, cfg->mismatch_score, cfg->match_score, 0,
                             &ovf, NULL);
        free(scores);
        free(trace);
    }
    return r;
}

void poa_run_topsort(int32_t* sorted_poa, int32_t* node_id_to_pos, int32_t node_count,
                     const uint16_t* incoming_edge_count, const int32_t* outgoing_edges,
                     const uint16_t* outgoing_edge_count)
{
    uint16_t* local = (uint16_t*)xcalloc((size_t)node_count + 1, 2);
    topsort_kahn(sorted_poa, node_id_to_pos, node_count, incoming_edge_count, outgoing_edges, outgoing_edge_count, local);
    free(local);
}

int32_t poa_run_add_alignment(uint8_t* nodes, int32_t* node_count, int32_t* node_alignments,
                              uint16_t* node_alignment_count, int32_t* incoming_edges,
                              uint16_t* incoming_edge_count, int32_t* outgoing_edges,
                              uint16_t* outgoing_edge_count, uint16_t* incoming_edge_w, int32_t alignment_length,
                              const int32_t* alignment_graph, const uint8_t* read, const int32_t* alignment_read,
                              uint16_t* node_coverage_counts, const int8_t* base_weights,
                              int32_t max_nodes_per_graph)
{
    int32_t new_count = *node_count;
    int32_t st = add_alignment_to_graph(&new_count, nodes, *node_count, node_alignments, node_alignment_count,
                                        incoming_edges, incoming_edge_count, outgoing_edges, outgoing_edge_count,
                                        incoming_edge_w, alignment_length, alignment_graph, read, alignment_read,
                                        node_coverage_counts, base_weights, 0, NULL, NULL, NULL, 0, 0,
                                        (uint32_t)max_nodes_per_graph);
    *node_count = new_count;
    return st;
}

void poa_run_consensus(const uint8_t* nodes, int32_t node_count, const int32_t* graph, const int32_t* node_id_to_pos,
                       const int32_t* incoming_edges, const uint16_t* incoming_edge_count,
                       const int32_t* outgoing_edges, const uint16_t* outgoing_edge_count,
                       const uint16_t* incoming_edge_w, uint8_t* consensus, uint16_t* coverage,
                       const uint16_t* node_coverage_counts, const int32_t* node_alignments,
                       const uint16_t* node_alignment_count, int32_t max_consensus_size)
{
    int32_t* scores = (int32_t*)xcalloc((size_t)node_count + 2, 4);
    int32_t* preds  = (int32_t*)xcalloc((size_t)node_count + 1, 4);
    generate_consensus(nodes, node_count, graph, node_id_to_pos, incoming_edges, incoming_edge_count, outgoing_edges,
                       outgoing_edge_count, incoming_edge_w, preds, scores, consensus, coverage, node_coverage_counts,
                       node_alignments, node_alignment_count, max_consensus_size);
    free(scores);
    free(preds);
}
