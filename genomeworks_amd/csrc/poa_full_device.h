// poa_full_device.h -- full-band NW (reference: cudapoa_nw.cuh:149-454), wave64.
// Row layout: element e = column + 3 of a row of `scores_width` elements, so the 4 cells of a lane
// (columns 1+4l .. 4+4l of a 256-column pass) are one aligned Quad. Semantics differ from the banded mode:
// min is numeric_limits<ScoreT>::min() (not halved), column 0 holds the true vertical boundary and IS the
// carry-in, there is no band predicate.
#pragma once
#include "poa_device.h"

namespace gwhip
{

template <typename ScoreT, typename IdT, typename RowT>
__device__ __forceinline__ int32_t nw_full(const GraphView<IdT>& g, RowT* rowinfo, int32_t graph_count, const uint8_t* read,
                           int32_t read_length, ScoreT* scores, int32_t scores_width, ScoreT* ring_base,
                           int32_t ring_bytes, int32_t* alignment_graph, int32_t* alignment_read, int32_t gap_score,
                           int32_t mismatch_score, int32_t match_score, uint64_t& cells)
{
    const int lane       = threadIdx.x & (kWave - 1);
    const int32_t stride = scores_width;
    const int32_t npass  = (read_length + 255) / 256;
    int32_t ring_rows    = ring_bytes / (int32_t)(stride * sizeof(ScoreT));
    if (ring_rows < 2) ring_rows = 0;
    ScoreT* ring = ring_base;
    cells += (uint64_t)graph_count * (uint64_t)read_length;

    for (int32_t j = lane; j <= read_length; j += kWave)
    {
        ScoreT v = (ScoreT)(j * gap_score);
        scores[j + kRelShift] = v;
        if (ring_rows) ring[j + kRelShift] = v;
    }
    bool hbm_dirty = false;
    wave_sync();

    for (int32_t r = 1; r <= graph_count; r++)
    {
        const RowT ri    = rowinfo[r];
        const int32_t pred_count = ri.cnt();
        const int32_t node_id    = (pred_count > 3) ? (int32_t)g.sorted_poa[r - 1] : 0;
        auto pred_row = [&](int32_t p) -> int32_t {
            if (pred_count == 0) return 0;
            if (p < 3) return ri.pred(p);
            return (int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + p]] + 1;
        };
        auto row_ptr = [&](int32_t row) -> const ScoreT* {
            if (ring_rows && r - row < ring_rows) return ring + (row % ring_rows) * stride;
            if (hbm_dirty) { wave_sync(); hbm_dirty = false; }
            return scores + (int64_t)row * stride;
        };
        // column 0 (:186-216): gap for sources, else gap + max over predecessors' column 0
        int32_t col0;
        if (pred_count == 0)
            col0 = (ScoreT)gap_score;
        else
        {
            int32_t penalty = Limits<ScoreT>::min;
            for (int32_t p = 0; p < pred_count; p++) penalty = max(penalty, (int32_t)row_ptr(pred_row(p))[kRelShift]);
            col0 = (ScoreT)(penalty + gap_score);
        }
        int32_t carry = col0;
        for (int32_t pass = 0; pass < npass; pass++)
        {
            const int32_t c   = pass * 256 + 4 * lane; // cells are columns c+1..c+4, read chars c..c+3
            const bool active = c < read_length;
            const uint32_t rd4 = *reinterpret_cast<const uint32_t*>(read + c);
            const int32_t cp0 = ((rd4 & 0xff) == (uint32_t)ri.base()) ? match_score : mismatch_score;
            const int32_t cp1 = (((rd4 >> 8) & 0xff) == (uint32_t)ri.base()) ? match_score : mismatch_score;
            const int32_t cp2 = (((rd4 >> 16) & 0xff) == (uint32_t)ri.base()) ? match_score : mismatch_score;
            const int32_t cp3 = ((rd4 >> 24) == (uint32_t)ri.base()) ? match_score : mismatch_score;
            int32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            const int32_t np = max(pred_count, 1);
            for (int32_t p = 0; p < np; p++)
            {
                const ScoreT* rowp = row_ptr(pred_row(p));
                int32_t S0 = 0, S1 = 0, S2 = 0, S3 = 0, S4 = 0;
                if (active)
                {
                    S0 = rowp[c + kRelShift];
                    Quad<ScoreT> qd = *reinterpret_cast<const Quad<ScoreT>*>(rowp + c + kRelShift + 1);
                    S1 = qd.v[0]; S2 = qd.v[1]; S3 = qd.v[2]; S4 = qd.v[3];
                }
                if (p == 0)
                {
                    s0 = (ScoreT)max(S0 + cp0, S1 + gap_score);
                    s1 = (ScoreT)max(S1 + cp1, S2 + gap_score);
                    s2 = (ScoreT)max(S2 + cp2, S3 + gap_score);
                    s3 = (ScoreT)max(S3 + cp3, S4 + gap_score);
                }
                else
                {
                    s0 = (ScoreT)max(S0 + cp0, max(s0, S1 + gap_score));
                    s1 = (ScoreT)max(S1 + cp1, max(s1, S2 + gap_score));
                    s2 = (ScoreT)max(S2 + cp2, max(s2, S3 + gap_score));
                    s3 = (ScoreT)max(S3 + cp3, max(s3, S4 + gap_score));
                }
            }
            const int32_t tb = 4 * lane;
            int32_t u0 = s0 - (tb + 0) * gap_score, u1 = s1 - (tb + 1) * gap_score;
            int32_t u2 = s2 - (tb + 2) * gap_score, u3 = s3 - (tb + 3) * gap_score;
            if (!active) u0 = u1 = u2 = u3 = INT32_MIN;
            const int32_t m0 = u0, m1 = max(m0, u1), m2 = max(m1, u2), m3 = max(m2, u3);
            const int32_t incl = wave_inclusive_max(m3);
            const int32_t excl = max(wave_shr1(incl, INT32_MIN), carry + gap_score);
            const int32_t n0 = (ScoreT)(max(m0, excl) + (tb + 0) * gap_score);
            const int32_t n1 = (ScoreT)(max(m1, excl) + (tb + 1) * gap_score);
            const int32_t n2 = (ScoreT)(max(m2, excl) + (tb + 2) * gap_score);
            const int32_t n3 = (ScoreT)(max(m3, excl) + (tb + 3) * gap_score);
            carry = wave_bcast(n3, kWave - 1);
            if (active)
            {
                Quad<ScoreT> out;
                out.v[0] = (ScoreT)n0; out.v[1] = (ScoreT)n1; out.v[2] = (ScoreT)n2; out.v[3] = (ScoreT)n3;
                *reinterpret_cast<Quad<ScoreT>*>(scores + (int64_t)r * stride + c + 1 + kRelShift) = out;
                if (ring_rows) *reinterpret_cast<Quad<ScoreT>*>(ring + (r % ring_rows) * stride + c + 1 + kRelShift) = out;
            }
        }
        if (lane == 0)
        {
            scores[(int64_t)r * stride + kRelShift] = (ScoreT)col0;
            if (ring_rows) ring[(r % ring_rows) * stride + kRelShift] = (ScoreT)col0;
        }
        hbm_dirty = true;
    }
    wave_sync();

    auto H = [&](int32_t i, int32_t j) -> int32_t { return scores[(int64_t)i * stride + j + kRelShift]; };

    // sink selection (:320-337)
    int32_t best = Limits<ScoreT>::min, best_i = 0;
    for (int32_t idx = 1 + lane; idx <= graph_count; idx += kWave)
    {
        if (rowinfo[idx].sink())
        {
            int32_t s = H(idx, read_length);
            if (best < s) { best = s; best_i = idx; }
        }
    }
    for (int off = 32; off > 0; off >>= 1)
    {
        int32_t ob = __shfl_xor(best, off), oi = __shfl_xor(best_i, off);
        if (ob > best || (ob == best && oi != 0 && (best_i == 0 || oi < best_i))) { best = ob; best_i = oi; }
    }

    int32_t aligned_nodes = 0;
    if (lane == 0) // traceback by recomputation (:340-445)
    {
        int32_t i = best_i, j = read_length, prev_i = 0, prev_j = 0, loop_count = 0;
        const int32_t bound = read_length + graph_count + 2;
        while (!(i == 0 && j == 0) && loop_count < bound)
        {
            loop_count++;
            int32_t scores_ij = H(i, j);
            bool pred_found   = false;
            RowT ri{};
            int32_t pred_count = 0, node_id = 0;
            if (i != 0)
            {
                ri         = rowinfo[i];
                pred_count = ri.cnt();
                if (pred_count > 3) node_id = g.sorted_poa[i - 1];
            }
            auto pred_row = [&](int32_t p) -> int32_t {
                if (pred_count == 0) return 0;
                if (p < 3) return ri.pred(p);
                return (int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + p]] + 1;
            };
            const int32_t np = max(pred_count, 1);
            if (i != 0 && j != 0)
            {
                int32_t match_cost = (ri.base() == read[j - 1] ? match_score : mismatch_score);
                for (int32_t p = 0; p < np; p++)
                {
                    int32_t pi = pred_row(p);
                    if (scores_ij == H(pi, j - 1) + match_cost) { prev_i = pi; prev_j = j - 1; pred_found = true; break; }
                }
            }
            if (!pred_found && i != 0)
            {
                for (int32_t p = 0; p < np; p++)
                {
                    int32_t pi = pred_row(p);
                    if (scores_ij == H(pi, j) + gap_score) { prev_i = pi; prev_j = j; pred_found = true; break; }
                }
            }
            // horizontal; j == 0 is unreachable here because column 0 always has a vertical match
            if (!pred_found && j > 0 && scores_ij == H(i, j - 1) + gap_score) { prev_i = i; prev_j = j - 1; pred_found = true; }
            alignment_graph[aligned_nodes] = (i == prev_i ? -1 : (int32_t)g.sorted_poa[i - 1]);
            alignment_read[aligned_nodes]  = (j == prev_j ? -1 : j - 1);
            aligned_nodes++;
            i = prev_i;
            j = prev_j;
        }
        if (loop_count >= bound) aligned_nodes = kNwLoopFailed;
    }
    return wave_first(aligned_nodes);
}

} // namespace gwhip

#include "poa_forward_moves_full.h" // the packed pass for int16 scores (round 5); nw_full above stays the generic routine
