// poa_tb_device.h -- banded NW with a traceback matrix and a score ring of `H = max_banded_pred_distance`
// rows (reference: cudapoa_nw_tb_banded.cuh:35-643), wave64.
//
// This mode is kept memory-faithful to the reference: the score ring uses the reference's own element
// indices (row % H, relative index, stride band_width + 8), because the reference's
// set_score_tb(column = -1) stores at relative index band_start of the ring row (:47-69) and that stray
// store can land in a live ring row. Stores that fall outside this window's own region are dropped.
// Two indeterminate reads of the reference are pinned to 0 (same as oracle/poa_nw_tb.inc): the
// never-initialised per-cell trace (:453) and trace row 0 (:576-577).
#pragma once
#include "poa_device.h"

namespace gwhip
{

template <typename ScoreT> struct TbCtx
{
    ScoreT* scores;
    size_t scores_elems;
    int32_t H, stride, band_width, band_shift, max_column, min_score;
    float gradient;
};

template <typename ScoreT>
__device__ __forceinline__ int32_t tb_get_score(const TbCtx<ScoreT>& t, int32_t row, int32_t column)
{
    int32_t bs   = band_start_for_row(row, t.gradient, t.band_width, t.band_shift, t.max_column);
    int32_t bend = min(bs + t.band_width, t.max_column);
    if ((column > bend || column < bs) && column != -1) return t.min_score;
    int32_t col = column == -1 ? 0 : column - bs;
    return t.scores[(int64_t)col + (int64_t)(row % t.H) * t.stride];
}

template <typename ScoreT>
__device__ __forceinline__ void tb_set_score(const TbCtx<ScoreT>& t, int32_t row, int32_t column, int32_t value,
                                             int32_t band_start)
{
    int32_t col_idx = (column == -1) ? band_start : column - band_start;
    int64_t idx     = (int64_t)col_idx + (int64_t)(row % t.H) * t.stride;
    if (idx >= 0 && (size_t)idx < t.scores_elems) t.scores[idx] = (ScoreT)value;
}

template <typename ScoreT, typename IdT, typename RowT, typename TraceT, bool ADAPTIVE>
__device__ __forceinline__ int32_t nw_banded_tb(const GraphView<IdT>& g, RowT* rowinfo, int32_t graph_count,
                                const uint8_t* read, int32_t read_length, ScoreT* scores, size_t scores_elems,
                                TraceT* traceback, size_t trace_elems, float max_buffer_size, int32_t* alignment_graph,
                                int32_t* alignment_read, int32_t band_width, int32_t H, int32_t gap_score,
                                int32_t mismatch_score, int32_t match_score, int32_t rerun, uint64_t& cells)
{
    const int lane           = threadIdx.x & (kWave - 1);
    const int32_t min_score  = Limits<ScoreT>::min / 2;
    const float gradient     = __fdiv_rn((float)(read_length + 1), (float)(graph_count + 1));
    const int32_t max_column = read_length + 1;
    int32_t band_shift       = band_width / 2;
    if (ADAPTIVE) // :306-332 (no gradient widening rules in this variant)
    {
        if (rerun == kShiftLeft && band_width <= kMaxAdaptiveBand / 2)
        {
            band_width *= 2;
            band_shift = (int32_t)((double)band_shift * 2.5);
        }
        if (rerun == kShiftRight && band_width <= kMaxAdaptiveBand / 2)
        {
            band_width *= 2;
            band_shift = (int32_t)((double)band_shift * 1.5);
        }
        float required = __fmul_rn((float)graph_count, (float)(band_width + kRightPad));
        if (required > max_buffer_size) return kNwAdaptiveStorageFailed;
    }
    cells += (uint64_t)graph_count * (uint64_t)band_width;

    TbCtx<ScoreT> t;
    t.scores = scores; t.scores_elems = scores_elems; t.H = H; t.stride = band_width + kRightPad;
    t.band_width = band_width; t.band_shift = band_shift; t.max_column = max_column; t.min_score = min_score;
    t.gradient = gradient;
    const int32_t stride = t.stride;
    auto tr_store = [&](int64_t idx, int32_t v) {
        if (idx >= 0 && (size_t)idx < trace_elems) traceback[idx] = (TraceT)v;
    };

    for (int32_t j = lane; j < stride; j += kWave)
    {
        tr_store(j, 0);                               // pinned: trace row 0
        tb_set_score(t, 0, j, j * gap_score, 0);      // :335-338
    }
    wave_sync();

    for (int32_t r = 1; r <= graph_count; r++)
    {
        const RowT ri    = rowinfo[r];
        const int32_t pred_count = ri.cnt();
        const int32_t bs         = band_start_for_row(r, gradient, band_width, band_shift, max_column);
        const int32_t node_id    = (pred_count > 3) ? (int32_t)g.sorted_poa[r - 1] : 0;
        auto pred_row = [&](int32_t p) -> int32_t {
            if (pred_count == 0) return 0;
            if (p < 3) return ri.pred(p);
            return (int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + p]] + 1;
        };
        const int32_t pred_idx0 = pred_row(0);
        int32_t fe = 0;
        if (lane == 0)
        {
            // initialize_band_tb :84-103
            const int32_t band_end = bs + band_width;
            const int32_t bs1      = max(1, bs);
            tb_set_score(t, r, bs1, min_score, bs1);
            for (int32_t j = band_end; j < band_end + kRightPad; j++) tb_set_score(t, r, j, min_score, bs1);
            // boundary :362-434
            const int64_t tindex = (int64_t)r * stride;
            if (pred_count == 0)
            {
                int64_t index = (int64_t)(r % H) * stride;
                if ((size_t)index < scores_elems) scores[index] = (ScoreT)gap_score;
                tr_store(tindex, -r);
            }
            else
            {
                int32_t penalty;
                if (((r - 1) - pred_idx0) < H) // :382 uses graph_pos = r - 1
                {
                    tr_store(tindex, -(r - pred_idx0));
                    if (bs > kCellsPerLane && pred_count == 1)
                        fe = min_score + gap_score;
                    else
                    {
                        penalty = max(min_score, tb_get_score(t, pred_idx0, -1));
                        for (int32_t p = 1; p < pred_count; p++)
                        {
                            int32_t pit = pred_row(p);
                            if ((r - pit) < H)
                            {
                                int32_t st = tb_get_score(t, pit, -1);
                                if (penalty < st) { penalty = st; tr_store(tindex, -(r - pit)); }
                            }
                        }
                        fe = penalty + gap_score;
                        tb_set_score(t, r, -1, fe, bs);
                    }
                }
                else
                {
                    penalty = min_score;
                    for (int32_t p = 1; p < pred_count; p++)
                    {
                        int32_t pit = pred_row(p);
                        if ((r - pit) < H)
                        {
                            int32_t st = tb_get_score(t, pit, -1);
                            if (penalty < st) { penalty = st; tr_store(tindex, -(r - pit)); }
                        }
                    }
                    fe = penalty + gap_score;
                    tb_set_score(t, r, -1, fe, bs);
                }
            }
        }
        fe = wave_first(fe);
        wave_sync();

        // A slot-0 predecessor farther than H rows is still used (:456-457) and reads whatever occupies its
        // ring slot -- possibly this row's own slot. Keep the reference's 128-column pass order there.
        const int32_t pass_cols = (pred_count > 0 && (r - pred_idx0) >= H) ? 128 : 256;
        const int32_t npass     = (band_width + pass_cols - 1) / pass_cols;
        int32_t carry = fe;
        for (int32_t pass = 0; pass < npass; pass++)
        {
            const int32_t off = pass * pass_cols + 4 * lane;
            const bool active = (4 * lane < pass_cols) && (off < band_width);
            const int32_t c   = bs + off;
            int32_t s[4]  = {min_score, min_score, min_score, min_score};
            int32_t tr[4] = {0, 0, 0, 0};
            if (active)
            {
                const uint32_t rd4 = *reinterpret_cast<const uint32_t*>(read + c);
                int32_t cp[4];
                cp[0] = ((rd4 & 0xff) == (uint32_t)ri.base()) ? match_score : mismatch_score;
                cp[1] = (((rd4 >> 8) & 0xff) == (uint32_t)ri.base()) ? match_score : mismatch_score;
                cp[2] = (((rd4 >> 16) & 0xff) == (uint32_t)ri.base()) ? match_score : mismatch_score;
                cp[3] = ((rd4 >> 24) == (uint32_t)ri.base()) ? match_score : mismatch_score;
                const int32_t np = max(pred_count, 1);
                for (int32_t p = 0; p < np; p++)
                {
                    const int32_t prow = (p == 0) ? pred_idx0 : pred_row(p);
                    if (p > 0 && !((r - prow) < H)) continue;
                    const int32_t pbs  = band_start_for_row(prow, gradient, band_width, band_shift, max_column);
                    const int32_t pend = min(pbs + band_width - kCellsPerLane, max_column);
                    if (c > pend || c < pbs) continue;
                    const ScoreT* ps    = scores + (int64_t)(c - pbs) + (int64_t)(prow % H) * stride;
                    const int32_t delta = r - prow;
                    int32_t S[5];
#pragma unroll
                    for (int k = 0; k < 5; k++) S[k] = ps[k];
#pragma unroll
                    for (int k = 0; k < 4; k++)
                    {
                        int32_t d = S[k] + cp[k], v = S[k + 1] + gap_score;
                        if (d >= v) { if (d > s[k]) { s[k] = (ScoreT)d; tr[k] = (TraceT)delta; } }
                        else        { if (v > s[k]) { s[k] = (ScoreT)v; tr[k] = (TraceT)(-delta); } }
                    }
                }
            }
            const int32_t tb = 4 * lane;
            int32_t u[4], m[4];
#pragma unroll
            for (int k = 0; k < 4; k++) u[k] = active ? s[k] - (tb + k) * gap_score : INT32_MIN;
            m[0] = u[0]; m[1] = max(m[0], u[1]); m[2] = max(m[1], u[2]); m[3] = max(m[2], u[3]);
            const int32_t incl = wave_inclusive_max(m[3]);
            const int32_t excl = max(wave_shr1(incl, INT32_MIN), carry + gap_score);
            int32_t h[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                h[k] = (ScoreT)(max(m[k], excl) + (tb + k) * gap_score);
                if (h[k] != s[k]) tr[k] = 0; // strictly improved by the horizontal move (:476-513)
            }
            const int last_lane = min(pass_cols, band_width - pass * pass_cols) / 4 - 1;
            carry = wave_bcast(h[3], last_lane);
            wave_sync(); // all loads of this pass done before its stores (reference: same warp instruction order)
            if (active)
            {
                int64_t index = (int64_t)(off + 1) + (int64_t)(r % H) * stride;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if ((size_t)(index + k) < scores_elems) scores[index + k] = (ScoreT)h[k];
                index = (int64_t)(off + 1) + (int64_t)r * stride;
#pragma unroll
                for (int k = 0; k < 4; k++) tr_store(index + k, tr[k]);
            }
            wave_sync();
        }
    }

    // sink selection :535-568 (rows restricted to the last H)
    int32_t best = min_score, best_i = 0;
    for (int32_t idx = 1 + lane; idx <= graph_count; idx += kWave)
    {
        if (rowinfo[idx].sink() && (graph_count - idx) < H)
        {
            int32_t s = tb_get_score(t, idx, read_length);
            if (best < s) { best = s; best_i = idx; }
        }
    }
    for (int off = 32; off > 0; off >>= 1)
    {
        int32_t ob = __shfl_xor(best, off), oi = __shfl_xor(best_i, off);
        if (ob > best || (ob == best && oi != 0 && (best_i == 0 || oi < best_i))) { best = ob; best_i = oi; }
    }

    int32_t aligned_nodes = 0;
    if (lane == 0)
    {
        int32_t i = best_i, j = read_length;
        if (i == 0) { j = 0; aligned_nodes = kNwTracebackBufferFailed; }
        int32_t loop_count  = 0;
        const int32_t bound = read_length + graph_count + 2;
        bool off_matrix     = false;
        while (!(i == 0 && j == 0) && loop_count < bound)
        {
            loop_count++;
            int32_t bs    = band_start_for_row(i, gradient, band_width, band_shift, max_column);
            int64_t tidx  = (int64_t)(j - bs) + (int64_t)i * stride;
            int32_t trace = (tidx >= 0 && (size_t)tidx < trace_elems) ? (int32_t)traceback[tidx] : 0;
            if (trace == 0)
            {
                alignment_graph[aligned_nodes] = -1;
                alignment_read[aligned_nodes]  = j - 1;
                j--;
            }
            else if (trace < 0)
            {
                alignment_graph[aligned_nodes] = g.sorted_poa[i - 1];
                alignment_read[aligned_nodes]  = -1;
                i += trace;
            }
            else
            {
                alignment_graph[aligned_nodes] = g.sorted_poa[i - 1];
                alignment_read[aligned_nodes]  = j - 1;
                i -= trace;
                j--;
                if (ADAPTIVE && rerun == 0 && band_width < kMaxAdaptiveBand)
                {
                    int32_t threshold = max(1, max_column / 1024);
                    if (j > threshold && j < max_column - threshold)
                    {
                        int32_t b2 = band_start_for_row(i, gradient, band_width, band_shift, max_column);
                        if (j <= b2 + threshold) { aligned_nodes = kShiftLeft; break; }
                        if (j >= (b2 + band_width - threshold)) { aligned_nodes = kShiftRight; break; }
                    }
                }
            }
            aligned_nodes++;
            if (i < 0 || j < 0) { off_matrix = true; break; }
        }
        if (loop_count >= bound || off_matrix) aligned_nodes = kNwLoopFailed;
    }
    return wave_first(aligned_nodes);
}

// ------------------------------------------------------------------------------------------------
// The packed flavour (round 4): int16 scores, int16 trace region (two byte planes), row table and read in LDS, bands 128 /
// 256 / 384 / 512 -- poa_forward_moves_tb.h + the sheared-tile walk of poa_traceback_moves.h (MODE 1). `handled` comes back false when
// the configuration or this read's graph needs the memory-faithful routine above (nothing observable has been written then:
// nw_banded_tb initialises everything it reads).
// ------------------------------------------------------------------------------------------------
template <typename IdT, bool ADAPTIVE>
__device__ __forceinline__ int32_t nw_banded_tb_packed(const GraphView<IdT>& g, RowInfo<true>* rowinfo, int32_t graph_count,
                                                       const uint8_t* lds_read, int32_t read_length, int16_t* scores, size_t scores_elems,
                                                       int16_t* traceback, size_t trace_elems, float max_buffer_size,
                                                       int32_t* alignment_graph, int32_t* alignment_read, int32_t band_width, int32_t H,
                                                       int32_t gap_score, int32_t mismatch_score, int32_t match_score, int32_t rerun,
                                                       uint64_t& cells, uint8_t* ring_lds, int32_t ring_bytes, const uint64_t* xpred,
                                                       int32_t dbg, bool& handled)
{
    handled                  = false;
    const int lane           = threadIdx.x & (kWave - 1);
    const int32_t min_score  = Limits<int16_t>::min / 2;
    const float gradient     = __fdiv_rn((float)(read_length + 1), (float)(graph_count + 1));
    const int32_t max_column = read_length + 1;
    int32_t band_shift       = band_width / 2;
    if (ADAPTIVE) // :306-332
    {
        if (rerun == kShiftLeft && band_width <= kMaxAdaptiveBand / 2)
        {
            band_width *= 2;
            band_shift = (int32_t)((double)band_shift * 2.5);
        }
        if (rerun == kShiftRight && band_width <= kMaxAdaptiveBand / 2)
        {
            band_width *= 2;
            band_shift = (int32_t)((double)band_shift * 1.5);
        }
    }
    const int32_t u_span = max(band_width, 256) * abs(gap_score); // u-space offset of the last band cell
    const bool packed_ok = (band_width == 256 || band_width == 128 || band_width == 384 || band_width == 512) && max_column >= band_width && ring_bytes >= kPkSlots * kPkSlotBytes &&
                           ring_bytes >= kMtBytes && xpred != nullptr && H >= 16 && !(dbg & 256) &&
                           abs(gap_score) <= 30 && abs(match_score) <= 100 && abs(mismatch_score) <= 100 &&
                           // no packed operation can leave int16 (the bounds of nw_banded's packed pass)
                           max(match_score, 0) * min(read_length, graph_count) + u_span + abs(match_score) <= 32767 &&
                           (graph_count + read_length) * min(min(gap_score, mismatch_score), 0) >= -32768 + 256 &&
                           min_score + 4 * min(min(gap_score, mismatch_score), 0) - u_span >= -32768 &&
                           // both byte planes inside the int16 trace region, the ring rows inside the score region
                           (size_t)(graph_count + 1) * (size_t)(band_width + kRightPad) <= trace_elems &&
                           (size_t)min(H, graph_count + 1) * (size_t)(band_width + kRightPad) <= scores_elems;
    if (!packed_ok) return 0;
    if (ADAPTIVE)
    {
        float required = __fmul_rn((float)graph_count, (float)(band_width + kRightPad));
        if (required > max_buffer_size)
        {
            handled = true;
            return kNwAdaptiveStorageFailed;
        }
    }
    const int32_t stride = band_width + kRightPad;
    TbCtx<int16_t> t;
    t.scores = scores; t.scores_elems = scores_elems; t.H = H; t.stride = stride;
    t.band_width = band_width; t.band_shift = band_shift; t.max_column = max_column; t.min_score = min_score;
    t.gradient = gradient;
    // row 0 of the ring (:335-338) and the band start of every row into the row table
    for (int32_t j = lane; j < stride; j += kWave) tb_set_score(t, 0, j, j * gap_score, 0);
    for (int32_t r = 1 + lane; r <= graph_count; r += kWave)
        rowinfo[r].set_bs(band_start_for_row(r, gradient, band_width, band_shift, max_column));
    wave_sync();
    TbPlanes planes;
    planes.ring       = scores;
    planes.ring_elems = scores_elems;
    planes.H          = H;
    planes.plane0     = reinterpret_cast<uint8_t*>(traceback);
    planes.plane1     = reinterpret_cast<uint8_t*>(traceback) + trace_elems;
    bool ok;
    if (band_width == 256)
        ok = banded_forward_tb<IdT, 256>(g, rowinfo, graph_count, lds_read, ring_lds, xpred, max_column, gap_score, mismatch_score, match_score, planes, dbg);
    else if (band_width == 128)
        ok = banded_forward_tb<IdT, 128>(g, rowinfo, graph_count, lds_read, ring_lds, xpred, max_column, gap_score, mismatch_score, match_score, planes, dbg);
    else if (band_width == 384)
        ok = banded_forward_tb<IdT, 384>(g, rowinfo, graph_count, lds_read, ring_lds, xpred, max_column, gap_score, mismatch_score, match_score, planes, dbg);
    else
        ok = banded_forward_tb<IdT, 512>(g, rowinfo, graph_count, lds_read, ring_lds, xpred, max_column, gap_score, mismatch_score, match_score, planes, dbg);
    if (!ok) return 0; // the caller reruns this read with nw_banded_tb
    handled = true;
    cells += (uint64_t)graph_count * (uint64_t)band_width;
    wave_sync(); // ring and planes complete and visible

    // sink selection :535-568 (rows restricted to the last H)
    int32_t best = min_score, best_i = 0;
    for (int32_t idx = 1 + lane; idx <= graph_count; idx += kWave)
    {
        if (rowinfo[idx].sink() && (graph_count - idx) < H)
        {
            int32_t s = tb_get_score(t, idx, read_length);
            if (best < s) { best = s; best_i = idx; }
        }
    }
    for (int off = 32; off > 0; off >>= 1)
    {
        int32_t ob = __shfl_xor(best, off), oi = __shfl_xor(best_i, off);
        if (ob > best || (ob == best && oi != 0 && (best_i == 0 || oi < best_i))) { best = ob; best_i = oi; }
    }
    best_i = wave_first(best_i);
    if (best_i == 0) return kNwTracebackBufferFailed; // :570
    BandedCtx<int16_t> b;
    b.scores = nullptr; b.ring = nullptr; b.ring_rows = 0;
    b.stride = stride; b.band_width = band_width; b.band_shift = band_shift; b.max_column = max_column;
    b.gradient = gradient; b.min_score = min_score;
    const int32_t n = traceback_moves<int16_t, IdT, RowInfo<true>, ADAPTIVE, 1>(b, g, rowinfo, graph_count, lds_read, read_length, best_i, alignment_graph,
                                                                               alignment_read, gap_score, mismatch_score, match_score, rerun, ring_lds,
                                                                               planes.plane0, planes.plane1);
    return wave_first(n);
}

} // namespace gwhip
