// poa_device.h -- gfx950 device code for the POA hot path (one 64-lane wavefront per window).
//
// What the reference computes is restated in oracle/ and SURVEY.md Appendix A; this file is how it is
// computed on CDNA4:
//   * the band row (256 cells = 64 lanes x 4 cells) is one wave-wide step; the previous row stays in
//     registers and is re-aligned to the new band start with DPP lane shifts, so the common predecessor
//     (distance 1) costs no memory round trip;
//   * nearby predecessors come from an LDS ring of recent rows, far ones from the HBM score matrix;
//   * the horizontal max-plus recurrence is a wave-level prefix-max (fixpoint of the reference's
//     shfl/any relaxation loop, identical unless an int16 store wraps -- see DESIGN.md);
//   * score rows stream to HBM as one aligned 8/16-byte store per lane (512 B / 1 KiB per wave-row);
//   * per-row predecessor info (base, predecessor rows, sink flag) is gathered once per read by all
//     lanes into an LDS table so neither the forward pass nor the traceback chases graph pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "poa_layout.h"

namespace gwhip
{

constexpr int kWave = 64;

template <typename T> struct Limits;
template <> struct Limits<int16_t> { static constexpr int32_t min = -32768; };
template <> struct Limits<int32_t> { static constexpr int32_t min = INT32_MIN; };

template <typename ScoreT> struct alignas(sizeof(ScoreT) * 4) Quad { ScoreT v[4]; };

// Per-row table entry. pred[] holds score-matrix rows (node_id_to_pos + 1) of the first 3 predecessor slots.
template <typename IdT> struct RowInfo;
template <> struct alignas(8) RowInfo<int16_t>
{
    uint8_t base;
    uint8_t cnt_sink; // bits 0-6: predecessor count (<= 50), bit 7: outgoing_edge_count == 0
    uint16_t pred[3];
};
template <> struct alignas(16) RowInfo<int32_t>
{
    uint8_t base;
    uint8_t cnt_sink;
    uint16_t pad;
    int32_t pred[3];
};

template <typename IdT> struct GraphView
{
    uint8_t* nodes;
    IdT* node_alignments;
    uint16_t* node_alignment_count;
    IdT* incoming_edges;
    uint16_t* incoming_edge_count;
    IdT* outgoing_edges;
    uint16_t* outgoing_edge_count;
    uint16_t* incoming_edge_w;
    IdT* sorted_poa;
    IdT* node_id_to_pos;
    uint16_t* local_cnt;
    uint16_t* coverage;
    int32_t* cons_scores;
    IdT* cons_pred;
    uint8_t* marks;
    uint8_t* check;
    IdT* to_visit;
    uint16_t* out_cov;
    uint16_t* out_cov_cnt;
    IdT* msa_pos;
    IdT* seq_begin;
};

// ---- band placement: IEEE fp32 multiply + truncation (cudapoa_nw_banded.cuh:67-78) ----
__device__ __forceinline__ int32_t band_start_for_row(int32_t row, float gradient, int32_t band_width,
                                                      int32_t band_shift, int32_t max_column)
{
    int32_t diagonal_index = (int32_t)(__fmul_rn((float)row, gradient));
    int32_t start_pos      = max(0, diagonal_index - band_shift);
    if (max_column < start_pos + band_width) start_pos = max(0, max_column - band_width + kCellsPerLane);
    start_pos = start_pos - (start_pos % kCellsPerLane);
    return start_pos;
}

// ---- wave-level helpers ----
__device__ __forceinline__ int32_t wave_bcast(int32_t v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int32_t wave_first(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// Inclusive prefix-max across the 64 lanes with the gfx9 DPP row-shift / row-broadcast sequence.
__device__ __forceinline__ int32_t wave_inclusive_max(int32_t v)
{
    constexpr int32_t ident = INT32_MIN;
#define GW_DPP_MAX(ctrl, rmask) v = max(v, __builtin_amdgcn_update_dpp(ident, v, ctrl, rmask, 0xf, false))
    GW_DPP_MAX(0x111, 0xf); // row_shr:1
    GW_DPP_MAX(0x112, 0xf); // row_shr:2
    GW_DPP_MAX(0x114, 0xf); // row_shr:4
    GW_DPP_MAX(0x118, 0xf); // row_shr:8
    GW_DPP_MAX(0x142, 0xa); // row_bcast:15 into rows 1,3
    GW_DPP_MAX(0x143, 0xc); // row_bcast:31 into rows 2,3
#undef GW_DPP_MAX
    return v;
}
// value of lane-1 (lane 0 receives `first`)
__device__ __forceinline__ int32_t wave_shr1(int32_t v, int32_t first)
{
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false); // wave_shr:1
}
// value of lane+1 (lane 63 receives `last`)
__device__ __forceinline__ int32_t wave_shl1(int32_t v, int32_t last)
{
    return __builtin_amdgcn_update_dpp(last, v, 0x130, 0xf, 0xf, false); // wave_shl:1
}

// ------------------------------------------------------------------------------------------------
// Banded score-matrix storage (our layout): row r occupies stride = band_width + 8 elements; the value
// the reference keeps at relative index `rel` (= column - band_start(r), 0 = left-boundary slot) is
// stored at element rel + 3, so each lane's 4 cells (rel 1+4l .. 4+4l) form one naturally aligned Quad.
// ------------------------------------------------------------------------------------------------
constexpr int kRelShift = 3;

template <typename ScoreT> struct BandedCtx
{
    ScoreT* scores;    // HBM score matrix of this window
    ScoreT* ring;      // LDS ring of recent rows (same row layout), ring_rows rows
    int32_t ring_rows; // 0 = no ring
    int32_t stride;
    int32_t band_width, band_shift, max_column;
    float gradient;
    int32_t min_score;
};

// scalar read with the reference's get_score() predicate (cudapoa_nw_banded.cuh:80-102), HBM copy
template <typename ScoreT>
__device__ __forceinline__ int32_t get_score(const BandedCtx<ScoreT>& b, int32_t row, int32_t column)
{
    int32_t bs   = band_start_for_row(row, b.gradient, b.band_width, b.band_shift, b.max_column);
    int32_t bend = min(bs + b.band_width, b.max_column);
    if ((column > bend || column < bs) && column != -1) return b.min_score;
    int32_t rel = column == -1 ? 0 : column - bs;
    return b.scores[(int64_t)row * b.stride + rel + kRelShift];
}

// ------------------------------------------------------------------------------------------------
// Lane-0 traceback by recomputation, exact restatement of cudapoa_nw_banded.cuh:428-549, reading the HBM
// score matrix and the LDS row table. Returns alignment length or an error / rerun code.
// ------------------------------------------------------------------------------------------------
template <typename ScoreT, typename IdT, bool ADAPTIVE>
__device__ int32_t traceback_banded(const BandedCtx<ScoreT>& b, const GraphView<IdT>& g, const RowInfo<IdT>* rowinfo,
                                    int32_t graph_count, const uint8_t* read, int32_t read_length, int32_t start_i,
                                    int32_t* alignment_graph, int32_t* alignment_read, int32_t gap_score,
                                    int32_t mismatch_score, int32_t match_score, int32_t rerun)
{
    int32_t aligned_nodes = 0;
    int32_t i = start_i, j = read_length;
    int32_t prev_i = 0, prev_j = 0;
    int32_t loop_count = 0;
    const int32_t bound = read_length + graph_count + 2;
    while (!(i == 0 && j == 0) && loop_count < bound)
    {
        loop_count++;
        int32_t scores_ij = get_score(b, i, j);
        bool pred_found   = false;
        RowInfo<IdT> ri{};
        int32_t pred_count = 0, node_id = 0;
        if (i != 0)
        {
            ri         = rowinfo[i];
            pred_count = ri.cnt_sink & 0x7f;
        }
        auto pred_row = [&](int32_t p) -> int32_t {
            if (pred_count == 0) return 0;
            if (p < 3) return (int32_t)ri.pred[p];
            return (int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + p]] + 1;
        };
        if (i != 0 && pred_count > 3) node_id = g.sorted_poa[i - 1];
        if (i != 0 && j != 0)
        {
            if (ADAPTIVE)
            {
                if (rerun == 0 && b.band_width < kMaxAdaptiveBand)
                {
                    int32_t threshold = max(1, b.max_column / 1024);
                    if (j > threshold && j < b.max_column - threshold)
                    {
                        int32_t bs = band_start_for_row(i, b.gradient, b.band_width, b.band_shift, b.max_column);
                        if (j <= bs + threshold) { aligned_nodes = kShiftLeft; break; }
                        if (j >= (bs + b.band_width - threshold)) { aligned_nodes = kShiftRight; break; }
                    }
                }
            }
            int32_t match_cost = (ri.base == read[j - 1] ? match_score : mismatch_score);
            int32_t np         = max(pred_count, 1);
            for (int32_t p = 0; p < np; p++)
            {
                int32_t pi = pred_row(p);
                if (scores_ij == get_score(b, pi, j - 1) + match_cost)
                {
                    prev_i = pi; prev_j = j - 1; pred_found = true;
                    break;
                }
            }
        }
        if (!pred_found && i != 0)
        {
            int32_t np = max(pred_count, 1);
            for (int32_t p = 0; p < np; p++)
            {
                int32_t pi = pred_row(p);
                if (scores_ij == get_score(b, pi, j) + gap_score)
                {
                    prev_i = pi; prev_j = j; pred_found = true;
                    break;
                }
            }
        }
        if (!pred_found && scores_ij == get_score(b, i, j - 1) + gap_score)
        {
            prev_i = i; prev_j = j - 1; pred_found = true;
        }
        alignment_graph[aligned_nodes] = (i == prev_i ? -1 : (int32_t)g.sorted_poa[i - 1]);
        alignment_read[aligned_nodes]  = (j == prev_j ? -1 : j - 1);
        aligned_nodes++;
        i = prev_i;
        j = prev_j;
    }
    if (loop_count >= bound) aligned_nodes = kNwLoopFailed;
    return aligned_nodes;
}

// ------------------------------------------------------------------------------------------------
// Per-read row table: all lanes gather (base, predecessor rows, sink flag) for rows 1..N.
// ------------------------------------------------------------------------------------------------
template <typename IdT>
__device__ void build_rowinfo(const GraphView<IdT>& g, int32_t graph_count, RowInfo<IdT>* rowinfo, int lane)
{
    for (int32_t r = 1 + lane; r <= graph_count; r += kWave)
    {
        int32_t node = g.sorted_poa[r - 1];
        RowInfo<IdT> ri{};
        ri.base     = g.nodes[node];
        int32_t cnt = g.incoming_edge_count[node];
        int32_t oc  = g.outgoing_edge_count[node];
        ri.cnt_sink = (uint8_t)((cnt & 0x7f) | (oc == 0 ? 0x80 : 0));
        for (int32_t p = 0; p < 3; p++)
        {
            int32_t pr = 0;
            if (p < cnt) pr = (int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node * kEdges + p]] + 1;
            ri.pred[p] = (decltype(ri.pred[0]))pr;
        }
        rowinfo[r] = ri;
    }
}

// ------------------------------------------------------------------------------------------------
// Banded NW (score-matrix modes): forward pass wave-wide, then sink selection (wave reduction with the
// reference's first-maximum tie rule) and the lane-0 traceback.
// ------------------------------------------------------------------------------------------------
template <typename ScoreT, typename IdT, bool ADAPTIVE>
__device__ int32_t nw_banded(const GraphView<IdT>& g, RowInfo<IdT>* rowinfo, int32_t graph_count, const uint8_t* read,
                             int32_t read_length, ScoreT* scores, ScoreT* ring_base, int32_t ring_bytes,
                             float max_buffer_size, int32_t* alignment_graph, int32_t* alignment_read,
                             int32_t band_width, int32_t gap_score, int32_t mismatch_score, int32_t match_score,
                             int32_t rerun, uint64_t& cells)
{
    const int lane              = threadIdx.x & (kWave - 1);
    const int32_t min_score     = Limits<ScoreT>::min / 2;
    const float gradient        = __fdiv_rn((float)(read_length + 1), (float)(graph_count + 1));
    const int32_t max_column    = read_length + 1;

    if (ADAPTIVE) // cudapoa_nw_banded.cuh:213-234
    {
        if ((double)gradient > 1.1)
            band_width = max(band_width, ((int32_t)((double)max_column * 0.08 * (double)gradient) + 127) & ~127);
        if ((double)gradient < 0.8)
            band_width = max(band_width, ((int32_t)((double)max_column * 0.1 / (double)gradient) + 127) & ~127);
        band_width = min(band_width, kMaxAdaptiveBand);
        if (band_width == kMaxAdaptiveBand && rerun != 0) return rerun;
    }
    int32_t band_shift = band_width / 2;
    if (ADAPTIVE) // :239-265
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

    BandedCtx<ScoreT> b;
    b.scores     = scores;
    b.stride     = band_width + kRightPad;
    b.band_width = band_width;
    b.band_shift = band_shift;
    b.max_column = max_column;
    b.gradient   = gradient;
    b.min_score  = min_score;
    b.ring       = ring_base;
    b.ring_rows  = ring_bytes / (int32_t)(b.stride * sizeof(ScoreT));
    if (b.ring_rows < 2) b.ring_rows = 0;
    const int32_t stride = b.stride;
    const int32_t npass  = (band_width + 255) / 256;
    const bool reg_path  = (npass == 1); // previous row carried in registers

    // row 0: H[0][rel] = rel * gap for rel < stride (:269-272); only rel <= band_width is ever read.
    for (int32_t rel = lane; rel <= band_width; rel += kWave)
    {
        ScoreT v = (ScoreT)(rel * gap_score);
        scores[rel + kRelShift] = v;
        if (b.ring_rows) b.ring[rel + kRelShift] = v;
    }
    // registers: previous row's cells (columns prev_bs+1+4*lane .. +4), valid when prev_row == r-1
    int32_t P0 = 0, P1 = 0, P2 = 0, P3 = 0;
    int32_t prev_bs   = 0;         // band start of the row held in P*
    int32_t prev_rel0 = 0;         // its relative-0 slot value
    if (reg_path)
    {
        int32_t c = 4 * lane; // row 0, band start 0: cells rel 1+4l..4+4l
        P0 = (ScoreT)((c + 1) * gap_score);
        P1 = (ScoreT)((c + 2) * gap_score);
        P2 = (ScoreT)((c + 3) * gap_score);
        P3 = (ScoreT)((c + 4) * gap_score);
        prev_rel0 = 0; // row 0, rel 0 = 0 * gap
    }
    bool hbm_dirty = true; // stores since the last workgroup sync (needed before reading the HBM matrix)
    __syncthreads();
    hbm_dirty = false;

    for (int32_t r = 1; r <= graph_count; r++)
    {
        const RowInfo<IdT> ri    = rowinfo[r];
        const int32_t pred_count = ri.cnt_sink & 0x7f;
        const int32_t bs         = band_start_for_row(r, gradient, band_width, band_shift, max_column);
        const int32_t node_id    = (pred_count > 3) ? (int32_t)g.sorted_poa[r - 1] : 0;
        auto pred_row = [&](int32_t p) -> int32_t {
            if (pred_count == 0) return 0;
            if (p < 3) return (int32_t)ri.pred[p];
            return (int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + p]] + 1;
        };
        // relative-0 slot of an arbitrary earlier row (get_score(row, -1): reads rel 0 unconditionally)
        auto rel0_of = [&](int32_t row) -> int32_t {
            if (reg_path && row == r - 1) return prev_rel0;
            if (b.ring_rows && r - row < b.ring_rows) return b.ring[(row % b.ring_rows) * stride + kRelShift];
            if (hbm_dirty) { __syncthreads(); hbm_dirty = false; }
            return scores[(int64_t)row * stride + kRelShift];
        };

        // ---- left boundary / carry-in (:293-326), wave-uniform ----
        int32_t fe       = 0;          // first_element_prev_score
        int32_t rel0_val = min_score;  // value kept in this row's relative-0 slot
        const int32_t pred_idx0 = pred_row(0);
        if (pred_count == 0)
        {
            if (bs == 0) rel0_val = (ScoreT)gap_score; // carry-in stays 0: reference quirk
        }
        else
        {
            if (bs > kCellsPerLane && pred_count == 1)
                fe = min_score + gap_score;
            else
            {
                int32_t penalty = max(min_score, rel0_of(pred_idx0));
                for (int32_t p = 0; p < pred_count; p++) penalty = max(penalty, rel0_of(pred_row(p)));
                fe = penalty + gap_score;
            }
            if (bs == 0) rel0_val = (ScoreT)fe;
        }

        int32_t carry = fe;
        int32_t N0 = 0, N1 = 0, N2 = 0, N3 = 0; // this row's cells of the (single) register pass
        for (int32_t pass = 0; pass < npass; pass++)
        {
            const int32_t c      = bs + pass * 256 + 4 * lane; // chunk anchor column (cells c+1..c+4)
            const bool active    = (pass * 256 + 4 * lane) < band_width;
            // read characters c .. c+3 (positions past the read are never consumed; buffer has zero slack)
            const uint32_t rd4   = *reinterpret_cast<const uint32_t*>(read + c);
            const int32_t cp0    = ((rd4 & 0xff) == ri.base) ? match_score : mismatch_score;
            const int32_t cp1    = (((rd4 >> 8) & 0xff) == ri.base) ? match_score : mismatch_score;
            const int32_t cp2    = (((rd4 >> 16) & 0xff) == ri.base) ? match_score : mismatch_score;
            const int32_t cp3    = ((rd4 >> 24) == ri.base) ? match_score : mismatch_score;
            int32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            const int32_t np = max(pred_count, 1);
            for (int32_t p = 0; p < np; p++)
            {
                const int32_t prow = (p == 0) ? pred_idx0 : pred_row(p);
                const int32_t pbs  = band_start_for_row(prow, gradient, band_width, band_shift, max_column);
                const int32_t pend = min(pbs + band_width - kCellsPerLane, max_column);
                const bool valid   = !(c > pend || c < pbs);
                int32_t S0, S1, S2, S3, S4; // predecessor row columns c .. c+4
                if (reg_path && prow == r - 1)
                {
                    const int32_t q = (bs - prev_bs) >> 2; // lane shift between the two bands
                    if (q == 0)
                    {
                        S0 = wave_shr1(P3, prev_rel0);
                        S1 = P0; S2 = P1; S3 = P2; S4 = P3;
                    }
                    else if (q == 1)
                    {
                        S0 = P3;
                        S1 = wave_shl1(P0, 0); S2 = wave_shl1(P1, 0); S3 = wave_shl1(P2, 0); S4 = wave_shl1(P3, 0);
                    }
                    else
                    {
                        int src = lane + q;
                        S0 = __shfl(P3, src - 1);
                        S1 = __shfl(P0, src); S2 = __shfl(P1, src); S3 = __shfl(P2, src); S4 = __shfl(P3, src);
                    }
                }
                else
                {
                    const ScoreT* rowp;
                    if (b.ring_rows && r - prow < b.ring_rows)
                        rowp = b.ring + (prow % b.ring_rows) * stride;
                    else
                    {
                        if (hbm_dirty) { __syncthreads(); hbm_dirty = false; }
                        rowp = scores + (int64_t)prow * stride;
                    }
                    if (valid)
                    {
                        const int32_t rel = c - pbs; // multiple of 4
                        S0 = rowp[rel + kRelShift];
                        Quad<ScoreT> qd = *reinterpret_cast<const Quad<ScoreT>*>(rowp + rel + kRelShift + 1);
                        S1 = qd.v[0]; S2 = qd.v[1]; S3 = qd.v[2]; S4 = qd.v[3];
                    }
                    else { S0 = S1 = S2 = S3 = S4 = 0; }
                }
                int32_t t0, t1, t2, t3;
                if (valid)
                {
                    t0 = (ScoreT)max(S0 + cp0, S1 + gap_score);
                    t1 = (ScoreT)max(S1 + cp1, S2 + gap_score);
                    t2 = (ScoreT)max(S2 + cp2, S3 + gap_score);
                    t3 = (ScoreT)max(S3 + cp3, S4 + gap_score);
                }
                else { t0 = t1 = t2 = t3 = min_score; }
                if (p == 0) { s0 = t0; s1 = t1; s2 = t2; s3 = t3; }
                else { s0 = max(s0, t0); s1 = max(s1, t1); s2 = max(s2, t2); s3 = max(s3, t3); }
            }
            // ---- horizontal max-plus scan: H[t] = max(v[t], H[t-1] + gap), H[-1] = carry ----
            // with u[t] = v[t] - t*gap this is a prefix maximum; t = 4*lane + k inside the pass.
            const int32_t tb = 4 * lane;
            int32_t u0 = s0 - (tb + 0) * gap_score;
            int32_t u1 = s1 - (tb + 1) * gap_score;
            int32_t u2 = s2 - (tb + 2) * gap_score;
            int32_t u3 = s3 - (tb + 3) * gap_score;
            if (!active) u0 = u1 = u2 = u3 = INT32_MIN;
            const int32_t m0 = u0, m1 = max(m0, u1), m2 = max(m1, u2), m3 = max(m2, u3);
            const int32_t incl  = wave_inclusive_max(m3);
            const int32_t cu    = carry + gap_score; // carry as element t = -1: carry - (-1)*gap
            const int32_t excl  = max(wave_shr1(incl, INT32_MIN), cu);
            N0 = (ScoreT)(max(m0, excl) + (tb + 0) * gap_score);
            N1 = (ScoreT)(max(m1, excl) + (tb + 1) * gap_score);
            N2 = (ScoreT)(max(m2, excl) + (tb + 2) * gap_score);
            N3 = (ScoreT)(max(m3, excl) + (tb + 3) * gap_score);
            // carry into the next pass = last cell of the last lane
            carry = wave_bcast(N3, kWave - 1);
            if (active)
            {
                Quad<ScoreT> out;
                out.v[0] = (ScoreT)N0; out.v[1] = (ScoreT)N1; out.v[2] = (ScoreT)N2; out.v[3] = (ScoreT)N3;
                const int32_t rel = pass * 256 + 4 * lane + 1;
                *reinterpret_cast<Quad<ScoreT>*>(scores + (int64_t)r * stride + rel + kRelShift) = out;
                if (b.ring_rows)
                    *reinterpret_cast<Quad<ScoreT>*>(b.ring + (r % b.ring_rows) * stride + rel + kRelShift) = out;
            }
        }
        if (lane == 0)
        {
            scores[(int64_t)r * stride + kRelShift] = (ScoreT)rel0_val;
            if (b.ring_rows) b.ring[(r % b.ring_rows) * stride + kRelShift] = (ScoreT)rel0_val;
        }
        hbm_dirty = true;
        if (reg_path)
        {
            P0 = N0; P1 = N1; P2 = N2; P3 = N3;
            prev_bs   = bs;
            prev_rel0 = rel0_val;
        }
    }
    __syncthreads(); // score matrix complete and visible to lane 0's traceback

    // ---- sink selection (:410-426): first row with the strictly greatest H(row, L) among sink rows ----
    int32_t best = min_score, best_i = 0;
    for (int32_t idx = 1 + lane; idx <= graph_count; idx += kWave)
    {
        if (rowinfo[idx].cnt_sink & 0x80)
        {
            int32_t s = get_score(b, idx, read_length);
            if (best < s) { best = s; best_i = idx; }
        }
    }
    for (int off = 32; off > 0; off >>= 1)
    {
        int32_t ob = __shfl_xor(best, off), oi = __shfl_xor(best_i, off);
        // strictly greater wins; on ties the lower row index wins; rows that never beat min_score stay 0
        if (ob > best || (ob == best && oi != 0 && (best_i == 0 || oi < best_i))) { best = ob; best_i = oi; }
    }

    int32_t aligned_nodes = 0;
    if (lane == 0)
        aligned_nodes = traceback_banded<ScoreT, IdT, ADAPTIVE>(b, g, rowinfo, graph_count, read, read_length, best_i,
                                                                alignment_graph, alignment_read, gap_score,
                                                                mismatch_score, match_score, rerun);
    aligned_nodes = wave_first(aligned_nodes);
    return aligned_nodes;
}

} // namespace gwhip
