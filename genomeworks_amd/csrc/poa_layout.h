// poa_layout.h -- per-window scratch layout shared by the host sizing code and the kernels.
// Our own layout (not the reference's BatchBlock carving, allocate_block.hpp:107-315): every per-window
// array lives in one contiguous, 256-byte aligned window slab so one window's working set is contiguous
// in HBM / L2, and the score matrix rows are 16-byte aligned for coalesced wave-wide stores.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/gwhip.h"

#ifdef __HIPCC__
#define GW_HD __host__ __device__
#else
#define GW_HD
#endif

namespace gwhip
{

constexpr int kEdges = GWHIP_MAX_NODE_EDGES;
constexpr int kAligns = GWHIP_MAX_NODE_ALIGNMENTS;
constexpr int kCellsPerLane = 4;   // CUDAPOA_CELLS_PER_THREAD: observable through the x4 band rounding
constexpr int kRightPad = 8;       // CUDAPOA_BANDED_MATRIX_RIGHT_PADDING: observable through matrix_sequence_dimension
constexpr int kMaxAdaptiveBand = 1536;
constexpr int kShiftLeft = -10, kShiftRight = -11;
constexpr int kNwLoopFailed = -1, kNwAdaptiveStorageFailed = -2, kNwTracebackBufferFailed = -3;
constexpr int kNwScoreWrapped = -5;   // a score did not fit the int16 matrix (narrow_chk, poa_device.h): the reference's result would depend on its relaxation order
constexpr int kNwNeedScoreRows = -6;  // the walk over move bytes needs a score row the forward pass kept out of HBM: nw_banded reruns the pass storing every row
constexpr int kNwPipelineFailed = -4; // the multi-wave forward pass gave up on a bounded hand-over wait (protocol error, never expected)
constexpr uint8_t kKernelError = 0xFF;

// StatusType values written to consensus[1] (cudapoa.hpp:34-49)
enum : uint8_t
{
    kNodeCountExceeded = 4,
    kEdgeCountExceeded = 5,
    kExceededAdaptiveBandedMatrixSize = 6,
    kExceededMaximumPredecessorDistance = 7,
    kLoopCountExceeded = 8,
    kExceededMaximumSequenceSize = 2,
    kGenericError = 12 // StatusType::generic_error
};

struct PoaLayout
{
    // byte offsets inside a window slab
    size_t nodes, in_cnt, out_cnt, aln_cnt, coverage, sorted, pos, local_cnt;
    size_t in_edges, in_w, out_edges, aligned;
    size_t cons_scores, cons_pred;
    size_t align_graph, align_read; // int32 each, max_nodes + max_sequence_size + 8 entries
    size_t rowinfo;                 // global fallback for the per-row predecessor table
    size_t marks, check, to_visit;  // racon topsort scratch
    size_t out_cov, out_cov_cnt, msa_pos, seq_begin; // MSA only
    size_t trace;                   // TraceT matrix (traceback modes)
    size_t codes;                   // one move code per score cell (banded score-matrix modes: int16 scores, or long reads), else 0
    size_t scores;                  // score matrix, or score ring in traceback modes
    size_t scores_elems, trace_elems;
    size_t per_window;              // slab size (banded modes; full band adds the variable score region)
    int32_t id_bytes, score_bytes, trace_bytes, rowinfo_bytes;
    int32_t align_capacity;
};

inline GW_HD size_t gw_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Host+device: compute the layout for a config. For full band the score matrix is not part of the fixed
// slab (its width is per window: cudapoa_batch.cuh:502-507); `scores` is then 0 and the caller adds it.
inline PoaLayout make_poa_layout(const gwhip_poa_config& c)
{
    PoaLayout L{};
    const size_t mn   = (size_t)c.max_nodes_per_graph;
    L.id_bytes        = c.size32 ? 4 : 2;
    L.score_bytes     = c.score32 ? 4 : 2;
    L.trace_bytes     = c.trace16 ? 2 : 1;
    L.rowinfo_bytes   = 24; // RowInfo<false>; the packed 8-byte flavour only ever lives in LDS
    L.align_capacity  = c.max_nodes_per_graph + c.max_sequence_size + 8;
    const bool msa    = (c.output_mask & 2) != 0;
    const bool tb     = c.band_mode == GWHIP_STATIC_BAND_TRACEBACK || c.band_mode == GWHIP_ADAPTIVE_BAND_TRACEBACK;
    size_t off        = 0;
    auto take         = [&](size_t bytes) { size_t o = off; off = gw_align_up(off + bytes, 16); return o; };
    L.nodes       = take(mn);
    L.in_cnt      = take(mn * 2);
    L.out_cnt     = take(mn * 2);
    L.aln_cnt     = take(mn * 2);
    L.coverage    = take(mn * 2);
    L.sorted      = take(mn * L.id_bytes);
    L.pos         = take(mn * L.id_bytes);
    L.local_cnt   = take(mn * 2);
    L.in_edges    = take(mn * kEdges * L.id_bytes);
    L.in_w        = take(mn * kEdges * 2);
    L.out_edges   = take(mn * kEdges * L.id_bytes);
    L.aligned     = take(mn * kAligns * L.id_bytes);
    L.cons_scores = take((mn + 4) * 4);
    L.cons_pred   = take(mn * L.id_bytes);
    L.align_graph = take((size_t)L.align_capacity * 4);
    L.align_read  = take((size_t)L.align_capacity * 4);
    L.rowinfo     = take((mn + 2) * L.rowinfo_bytes);
    L.marks       = take(mn);
    L.check       = take(mn);
    L.to_visit    = take(mn * L.id_bytes);
    if (msa)
    {
        L.out_cov     = take(mn * kEdges * (size_t)c.max_sequences_per_poa * 2);
        L.out_cov_cnt = take(mn * kEdges * 2);
        L.msa_pos     = take(mn * L.id_bytes);
        L.seq_begin   = take(((size_t)c.max_sequences_per_poa + 1) * L.id_bytes);
    }
    const size_t msd = (size_t)c.matrix_sequence_dimension;
    if (tb)
    {
        L.trace_elems  = mn * msd;
        L.trace        = take(L.trace_elems * L.trace_bytes + 64);
        L.scores_elems = (size_t)c.max_banded_pred_distance * msd;
        L.scores       = take(L.scores_elems * L.score_bytes + 64);
    }
    else if (c.band_mode != GWHIP_FULL_BAND)
    {
        // int16 scores: poa_forward_packed.h; 32-bit ids in the adaptive band mode (long reads): generic_forward_skew
        if (!c.score32 || (c.size32 && c.band_mode == GWHIP_ADAPTIVE_BAND)) L.codes = take(mn * msd + 64);
        L.scores_elems = mn * msd;
        L.scores       = take(L.scores_elems * L.score_bytes + 64);
    }
    L.per_window = gw_align_up(off, 256);
    return L;
}

} // namespace gwhip
