// poa_traceback_moves.h -- traceback over the move bytes the forward pass of poa_forward_moves.h left (256-column band,
// int16 scores, row table in LDS). Same decision sequence as cudapoa_nw_banded.cuh:428-549 (diagonal through
// predecessor 0..n-1, then vertical through predecessor 0..n-1, then horizontal; first equality wins) -- the forward
// pass evaluated exactly those comparisons on the stored 16-bit operands and recorded their outcome per cell; cells
// where it could not (move byte 0) are stepped by exact recomputation from the HBM score matrix.
//
// The walk is one chain of dependent steps on a lone wavefront, so what counts is instructions and LDS round trips
// per step:
//   * a SHEARED tile of move bytes in LDS (the forward pass's ring, dead by now): tile row t holds matrix row
//     top - t, byte c of it column L - t + c, rows 72 bytes apart (64 columns + 8 zero bytes). A move of `rows up`
//     d and `columns left` e changes the byte address by 73 d - e, nothing else: no window arithmetic, no row table,
//     no band starts. A byte outside the window (pad bytes, eight zero rows below the tile) reads 0, the same value
//     as "undecided here", so the walk has ONE exit test; the slow path then tells the two apart.
//   * RUN SKIPPING: 80 % / 60 % (early / late reads) of all steps go one row up and one column left, and they come in
//     runs (profiles/r03_traceback_run_length_model.json: 5.6 / 2.6 steps per run incl. the step that ends it).
//     Lane k reads the byte 72 k further (the cell k such steps ahead); a ballot of "not that move" and a
//     count-trailing-zeros give the run length, the ending move is fetched with v_readlane and taken in the same
//     iteration: one LDS round trip per run instead of per step.
//   * the (graph position, read position) pairs of a run are written straight to HBM by the lanes (the loop has no
//     global load, so a store never makes it wait); graph positions are translated to node ids by all lanes afterwards.
//   * the NEXT tile (96 rows further up, at the column the band's slope predicts) is requested as soon as the current
//     one is in place and waits in registers.
// In the adaptive band mode the cells inside the band-edge margins of cudapoa_nw_banded.cuh:442-465 are committed
// to the tile as 0, so the shift-left / shift-right tests run in the slow path only.
//
// Two geometries. Row table in LDS (RowInfo<true>: the 256 / 128-column int16 pass): a predecessor is at most 7 rows up, so
// 8 pad bytes per tile row and 8 zero rows catch every move that leaves the window. Row table in HBM (RowInfo<false>: long
// reads, generic_forward_skew, 16- or 32-bit scores, bands up to 1 536 columns): moves go up to 63 rows up, so a tile row is
// 64 columns + 72 zero bytes (136: still an even number of dwords that is not a multiple of 4, i.e. a two-way bank conflict
// at worst for the 64 lanes of a run probe) above 64 zero rows; band starts come from the band formula instead of the table.
#pragma once

namespace gwhip
{

constexpr int kMtRows = 96, kMtCols = 64, kMtFront = 8, kMtPasses = kMtRows / 16;
template <bool WIDE> struct MtGeometry
{
    static constexpr int kStride   = WIDE ? 136 : 72; // bytes between tile rows: 64 columns + zero bytes
    static constexpr int kZeroRows = WIDE ? 64 : 8;   // zero rows below the tile
    static constexpr int kMaxUp    = WIDE ? 63 : 7;   // rows a move may go up
    static constexpr int kBytes    = kMtFront + (kMtRows + kZeroRows) * kStride;
};
constexpr int kMtBytes     = MtGeometry<false>::kBytes;
constexpr int kMtBytesWide = MtGeometry<true>::kBytes;

__device__ __forceinline__ uint32_t lds_load_u8(uint32_t addr)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) uint8_t*>(addr);
}
__device__ __forceinline__ void lds_store_zero_u64(uint32_t addr)
{
    u32x2 v;
    v.x = 0; v.y = 0;
    *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(addr) = v;
}

// MODE 0: score-matrix modes -- a byte 0 is a cell the forward pass left undecided, stepped by recomputation from the HBM score
// matrix. MODE 1: traceback-buffer modes (poa_forward_moves_tb.h) -- `moves` is plane 0 and `plane1` the exact int8 traces of
// the general rows; a byte 0 is a cell of a general row (or of a band-edge margin), stepped from the planes the way the
// reference walks its trace matrix (cudapoa_nw_tb_banded.cuh:572-636), and the band-edge tests of the adaptive mode run where
// the reference runs them: after a diagonal move, on the cell it arrives at.
template <typename ScoreT, typename IdT, typename RowT, bool ADAPTIVE, int MODE = 0>
__device__ __forceinline__ int32_t traceback_moves(const BandedCtx<ScoreT>& b, const GraphView<IdT>& g, const RowT* rowinfo,
                                                   int32_t graph_count, const uint8_t* read, int32_t read_length, int32_t start_i,
                                                   int32_t* alignment_graph, int32_t* alignment_read, int32_t gap_score,
                                                   int32_t mismatch_score, int32_t match_score, int32_t rerun, uint8_t* tile_region,
                                                   const uint8_t* moves, const uint8_t* plane1 = nullptr, int32_t score_rows = 1)
{
    constexpr bool kWide    = !std::is_same<RowT, RowInfo<true>>::value;
    constexpr int kMtStride = MtGeometry<kWide>::kStride, kMtZeroRows = MtGeometry<kWide>::kZeroRows;
    constexpr int kHalf = 31;
    const int lane      = threadIdx.x & (kWave - 1);
    const int32_t bound = read_length + graph_count + 2;
    int32_t n = 0; // steps taken = entries written (the reference's loop counter and aligned_nodes at once)
    int32_t i = start_i, j = read_length;
    const uint32_t T0 = lds_addr(tile_region) + kMtFront;

    // zero bytes the loader never touches: the front pad, the 8 pad bytes of every row, the rows below the tile
    wave_sync();
    for (int32_t q = lane; q < kMtRows * ((kMtStride - kMtCols) / 8); q += kWave)
        lds_store_zero_u64(T0 + (uint32_t)(q / ((kMtStride - kMtCols) / 8)) * kMtStride + kMtCols + (uint32_t)(q % ((kMtStride - kMtCols) / 8)) * 8);
    for (int32_t q = lane; q < kMtZeroRows * kMtStride / 8; q += kWave) lds_store_zero_u64(T0 + kMtRows * kMtStride + (uint32_t)q * 8);
    if (lane == 0) lds_store_zero_u64(T0 - kMtFront);

    // ---- tile loader: 4 lanes per tile row (16 bytes each), 16 rows per pass ----
    // Window byte kk of tile row t is the move of (row top - t, column L - t + kk), byte L - t + kk - bs(row) + 3 of
    // that row of the HBM matrix: any alignment, so five aligned dwords are loaded and funnel-shifted.
    struct Seg { uint32_t d[5]; };
    const int seg = lane & 3;
    // relative columns (column - band start) whose bytes are committed; the boundary cell (relative column 0) is a cell of the
    // traceback-buffer modes' trace matrix
    int32_t lo_rel = MODE == 1 ? 0 : 1, hi_rel = b.band_width;
    bool last_diag = false; // MODE 1, adaptive: the last step taken by the walk over move bytes was diagonal
    if (ADAPTIVE)
    {
        if (rerun == 0 && b.band_width < kMaxAdaptiveBand)
        {
            const int32_t threshold = max(1, b.max_column / 1024);
            lo_rel = threshold + 1;
            hi_rel = b.band_width - threshold - 1;
        }
    }
    auto band_start_of = [&](int32_t row) -> int32_t { // row >= 1
        if constexpr (kWide)
            return band_start_for_row(row, b.gradient, b.band_width, b.band_shift, b.max_column);
        else
            return rowinfo[row].bs();
    };
    auto issue_tile = [&](int32_t top, int32_t L, Seg (&v)[kMtPasses]) {
#pragma unroll
        for (int pass = 0; pass < kMtPasses; pass++)
        {
            const int32_t t    = pass * 16 + (lane >> 2);
            const int32_t rowc = max(top - t, 1);
            const int32_t e0   = (L - t) + 16 * seg - band_start_of(rowc) + kRelShift;
            const uint32_t* p  = reinterpret_cast<const uint32_t*>(moves + (int64_t)rowc * b.stride + (e0 & ~3));
#pragma unroll
            for (int k = 0; k < 5; k++) v[pass].d[k] = p[k];
        }
    };
    int32_t ctop = -(1 << 20), cL = 0; // tile anchor (far below any row while no tile is loaded)
    auto commit_tile = [&](int32_t top, int32_t L, Seg (&v)[kMtPasses]) {
        ctop = top;
        cL   = L;
#pragma unroll
        for (int pass = 0; pass < kMtPasses; pass++)
        {
            const int32_t t   = pass * 16 + (lane >> 2);
            const int32_t row = top - t;
            const int32_t bsr = band_start_of(max(row, 1));
            const int32_t e0  = (L - t) + 16 * seg - bsr + kRelShift;
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = __builtin_amdgcn_alignbyte(v[pass].d[k + 1], v[pass].d[k], (uint32_t)e0 & 3u);
            // window bytes that may be committed: columns band start + lo_rel .. + hi_rel of rows >= 1
            const int32_t klo = row >= 1 ? bsr + lo_rel - (L - t) : 1;
            const int32_t khi = row >= 1 ? bsr + hi_rel - (L - t) : 0;
            if (__ballot(!(klo <= 0 && khi >= kMtCols - 1)) != 0)
            {
#pragma unroll
                for (int d = 0; d < 4; d++)
                {
                    const int32_t k0 = seg * 16 + d * 4;
                    const int32_t lo = min(max(klo - k0, 0), 4), hi = min(max(khi - k0 + 1, 0), 4);
                    const uint32_t mhi = hi >= 4 ? 0xffffffffu : ((1u << (8 * hi)) - 1u);
                    const uint32_t mlo = lo >= 4 ? 0xffffffffu : ((1u << (8 * lo)) - 1u);
                    w[d] &= hi > lo ? (mhi & ~mlo) : 0u;
                }
            }
            const uint32_t dst = T0 + (uint32_t)t * kMtStride + (uint32_t)seg * 16;
            lds_store_u64(dst, w[0], w[1]);
            lds_store_u64(dst + 8, w[2], w[3]);
        }
        wave_sync();
    };
    Seg ahead[kMtPasses] = {};
    int32_t atop = -1, aL = 0; // anchor of the tile in `ahead` (atop < 0: none)
    // A window is 64 columns wide and sheared by one column per row; the path moves `gradient` columns per row, so it
    // drifts (1 - gradient) columns to the right per row inside the window: enter where the drift over the tile stays in.
    const int32_t c_in     = min(max(32 - (int32_t)((1.0f - b.gradient) * (float)(kMtRows / 2)), 8), 56);
    const int32_t slope_q8 = (int32_t)(b.gradient * 256.0f);
    const int32_t ahead_dL = (slope_q8 * kMtRows) >> 8; // columns the path is expected to move over one tile of rows
    auto load_tile = [&](int32_t row, int32_t col) {
        bool hit = false;
        if (atop >= 0)
        {
            const int32_t t = atop - row;
            const int32_t c = col - aL + t;
            hit = ((uint32_t)t < 16u) & ((uint32_t)(c - 6) < (uint32_t)(kMtCols - 12));
        }
        if (hit)
            commit_tile(atop, aL, ahead);
        else
        {
            Seg now[kMtPasses];
            issue_tile(row, col - c_in, now);
            commit_tile(row, col - c_in, now);
        }
        atop = ctop - kMtRows;
        aL   = cL - ahead_dL;
        if (atop >= 1) issue_tile(atop, aL, ahead);
        else atop = -1;
    };

    // lane roles of a recomputed step (VGPR constants): lanes 0..30 diagonal through predecessor slot p, 31..61 vertical,
    // 62 horizontal, 63 the cell itself
    const int kind        = lane < kHalf ? 0 : (lane < 2 * kHalf ? 1 : (lane == 2 * kHalf ? 2 : 3));
    const int p           = kind == 0 ? lane : lane - kHalf;
    const int psh         = 24 + 12 * min(p, 2);
    const bool is_diag    = kind == 0, is_vert = kind == 1, is_horiz = kind == 2, is_self = kind == 3;
    const int32_t col_dec = (is_vert | is_self) ? 0 : 1; // candidate column = j - col_dec
    const uint32_t lane_step = (uint32_t)lane * kMtStride;  // byte distance of the cell `lane` steps (1 up, 1 left) ahead

    while (!(i == 0 && j == 0) && n < bound)
    {
        if (i > 0)
        {
            int32_t t = ctop - i;
            int32_t c = j - cL + t;
            if (!(((uint32_t)t < (uint32_t)kMtRows) & ((uint32_t)c < (uint32_t)kMtCols)))
            {
                load_tile(i, j);
                t = ctop - i;
                c = j - cL + t;
            }
            // ---------------- the walk over move bytes ----------------
            uint32_t sa = T0 + (uint32_t)t * kMtStride + (uint32_t)c;
            for (;;)
            {
                const uint32_t m    = lds_load_u8(sa + lane_step);
                const uint64_t brk  = __ballot(m != 3u) | (1ull << 63);
                const int32_t nrun  = (int32_t)__builtin_ctzll(brk);                       // steps one up, one left
                const uint32_t mb   = (uint32_t)__builtin_amdgcn_readlane((int32_t)m, nrun); // the move after them
                const int32_t drow  = (int32_t)(mb >> 1), dcol = (int32_t)(mb & 1u);
                const int32_t take  = nrun + (mb != 0u ? 1 : 0);
                // entries n .. n + take - 1: lane k = the cell k steps ahead, (i - k, j - k)
                int32_t og = i - 1 - lane, orr = j - 1 - lane;
                if (lane == nrun)
                {
                    og  = drow != 0 ? og : -1;
                    orr = dcol != 0 ? orr : -1;
                }
                if (lane < take)
                {
                    alignment_graph[n + lane] = og;
                    alignment_read[n + lane]  = orr;
                }
                n += take;
                i -= nrun + drow;
                j -= nrun + dcol;
                sa += (uint32_t)(nrun * kMtStride + drow * (kMtStride + 1) - dcol);
                if constexpr (MODE == 1 && ADAPTIVE) last_diag = mb != 0u ? (drow != 0 && dcol != 0) : (nrun > 0 ? true : last_diag);
                if (mb == 0u) break;
            }
            // a byte 0: a cell the forward pass left undecided -- or a cell outside the tile
            t = ctop - i;
            c = j - cL + t;
            if (!(((uint32_t)t < (uint32_t)kMtRows) & ((uint32_t)c < (uint32_t)kMtCols)) && i > 0) continue;
            if ((i == 0 && j == 0) || n >= bound) continue; // the outer condition ends the walk
        }
        if constexpr (MODE == 1)
        {
            // ---------------- one step of the reference's walk over its trace matrix (cudapoa_nw_tb_banded.cuh:572-636) ----------------
            auto edge_test = [&](int32_t ii, int32_t jj) -> int32_t { // :603-626, on the cell a diagonal move arrived at
                if (rerun == 0 && b.band_width < kMaxAdaptiveBand)
                {
                    const int32_t threshold = max(1, b.max_column / 1024);
                    if (jj > threshold && jj < b.max_column - threshold)
                    {
                        const int32_t bs2 = band_start_for_row(ii, b.gradient, b.band_width, b.band_shift, b.max_column);
                        if (jj <= bs2 + threshold) return kShiftLeft;
                        if (jj >= (bs2 + b.band_width - threshold)) return kShiftRight;
                    }
                }
                return 0;
            };
            if constexpr (ADAPTIVE)
            {
                if (last_diag)
                {
                    const int32_t e = edge_test(i, j);
                    if (e != 0) { n = e; break; }
                }
                last_diag = false;
            }
            int32_t drow = 0, dcol = 1; // horizontal: trace row 0, and (pinned) whatever lies outside the band
            bool to_row0 = false;
            if (i > 0)
            {
                const int32_t rel = j - band_start_of(i);
                if (rel >= 0 && rel <= b.band_width)
                {
                    const int64_t idx = (int64_t)i * b.stride + rel + kRelShift;
                    const uint32_t m  = (uint32_t)wave_first((int32_t)moves[idx]);
                    if (m != 0u)
                    {
                        drow = (int32_t)(m >> 1);
                        dcol = (int32_t)(m & 1u);
                    }
                    else
                    {
                        const int32_t t = (int32_t)(int8_t)wave_first((int32_t)plane1[idx]);
                        to_row0         = t == kTbVertToRow0 || t == kTbDiagToRow0;
                        if (t > 0) { drow = t; dcol = 1; }
                        else if (t < 0) { drow = -t; dcol = 0; }
                    }
                }
            }
            if (to_row0) drow = i;
            if (lane == 0)
            {
                alignment_graph[n] = drow != 0 ? i - 1 : -1; // sorted position; node ids are filled in below
                alignment_read[n]  = dcol != 0 ? j - 1 : -1;
            }
            i -= drow;
            j -= dcol;
            if constexpr (ADAPTIVE)
            {
                if (drow != 0 && dcol != 0)
                {
                    const int32_t e = edge_test(i, j);
                    if (e != 0) { n = e; break; }
                }
            }
            n++;
            if (i < 0 || j < 0) { n = bound; break; } // walked off the matrix: the reference's loop failure
            continue;
        }
        // ---------------- one step by recomputation (exact restatement of :428-549), one candidate per lane ----------------
        // the row's record, wave-uniform: base, predecessor count, the rows of predecessor slots 0..2
        uint64_t riw = 0;
        int32_t pred_count = 0, pr0 = 0, pr1 = 0, pr2 = 0;
        uint32_t row_base = 0;
        if constexpr (!kWide)
        {
            riw        = i != 0 ? wave_first64(rowinfo[i].w) : 0;
            row_base   = (uint32_t)riw & 0xffu;
            pred_count = (int32_t)(((uint32_t)riw >> 8) & 0x3f);
        }
        else if (i != 0)
        {
            const RowT ri = uniform_row(rowinfo[i]);
            row_base      = (uint32_t)ri.base();
            pred_count    = ri.cnt();
            pr0 = ri.pred(0); pr1 = ri.pred(1); pr2 = ri.pred(2);
        }
        // score_rows: 1 = every score row is in HBM; 0 (round 5, the 256 / 128-column pass) = the forward pass kept the rows
        // nobody was known to read out of HBM (mark_score_rows). A cell of such a row that must be stepped by recomputation --
        // the band's first cell, a chunk outside a predecessor's band: the walk rarely passes through either outside the rows
        // whose band starts at column 0 -- ends this walk; nw_banded reruns the forward pass with every row stored and walks
        // again. 2 = as 0, but EVERY recomputed step below row 0 ends the walk (test arm: the rerun path on every read).
        if constexpr (!kWide && MODE == 0)
        {
            if (score_rows != 1 && i != 0 && (score_rows == 2 || !row_recompute_safe(riw)))
            {
                n = kNwNeedScoreRows;
                break;
            }
        }
        const uint32_t rch = j > 0 ? (uint32_t)wave_first((int32_t)read[j - 1]) : 0u;
        const int32_t np   = max(pred_count, 1);
        if (ADAPTIVE)
        {
            if (i != 0 && j != 0 && rerun == 0 && b.band_width < kMaxAdaptiveBand)
            {
                int32_t threshold = max(1, b.max_column / 1024);
                if (j > threshold && j < b.max_column - threshold)
                {
                    int32_t bs = band_start_for_row(i, b.gradient, b.band_width, b.band_shift, b.max_column);
                    if (j <= bs + threshold) { n = kShiftLeft; break; }
                    if (j >= (bs + b.band_width - threshold)) { n = kShiftRight; break; }
                }
            }
        }
        const int32_t match_cost = (row_base == rch ? match_score : mismatch_score);
        // the reference keeps the previous step's target when no candidate matches: that is the current cell (or (0, 0)
        // on the first step)
        int32_t next_i = n > 0 ? i : 0, next_j = n > 0 ? j : 0;
        int32_t scores_ij;
        if (np <= kHalf)
        {
            const bool en = (is_diag & (i != 0) & (j != 0) & (p < np)) | (is_vert & (i != 0) & (p < np)) | is_horiz;
            int32_t crow;
            if constexpr (!kWide)
                crow = pred_count != 0 ? (int32_t)((riw >> psh) & 0xfff) : 0;
            else
                crow = pred_count != 0 ? (p == 0 ? pr0 : (p == 1 ? pr1 : pr2)) : 0;
            crow          = (is_horiz | is_self) ? i : crow;
            if (pred_count > 3) // predecessor slots beyond the three packed ones live in the HBM edge list
            {
                if (en && !is_horiz && p >= 3)
                {
                    const int32_t node_id = g.sorted_poa[i - 1];
                    crow = (int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + p]] + 1;
                }
            }
            const int32_t ccol = j - col_dec;
            const int32_t cost = is_diag ? match_cost : gap_score;
            int32_t val        = 0;
            if (en | is_self) val = get_score(b, crow, ccol);
            scores_ij        = __builtin_amdgcn_readlane(val, kWave - 1);
            const bool hit   = en & (scores_ij == val + cost);
            const uint64_t m = __ballot(hit);
            if (m != 0)
            {
                const int sel = __ffsll((unsigned long long)m) - 1;
                next_i        = __builtin_amdgcn_readlane(crow, sel);
                next_j        = __builtin_amdgcn_readlane(ccol, sel);
            }
        }
        else // more predecessors than candidate lanes: the reference's sequential order, wave-uniform
        {
            scores_ij             = wave_first(get_score(b, i, j));
            const int32_t node_id = g.sorted_poa[i - 1];
            auto pred_row = [&](int32_t q) -> int32_t {
                if constexpr (!kWide)
                {
                    RowInfo<true> ri;
                    ri.w = riw;
                    if (q < 3) return ri.pred(q);
                }
                else if (q < 3)
                    return q == 0 ? pr0 : (q == 1 ? pr1 : pr2);
                return wave_first((int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + q]] + 1);
            };
            bool f = false;
            if (j != 0)
                for (int32_t q = 0; q < np && !f; q++)
                {
                    const int32_t pi = pred_row(q);
                    if (scores_ij == wave_first(get_score(b, pi, j - 1)) + match_cost) { next_i = pi; next_j = j - 1; f = true; }
                }
            for (int32_t q = 0; q < np && !f; q++)
            {
                const int32_t pi = pred_row(q);
                if (scores_ij == wave_first(get_score(b, pi, j)) + gap_score) { next_i = pi; next_j = j; f = true; }
            }
            if (!f && scores_ij == wave_first(get_score(b, i, j - 1)) + gap_score) { next_i = i; next_j = j - 1; f = true; }
        }
        if (lane == 0)
        {
            alignment_graph[n] = i == next_i ? -1 : i - 1; // sorted position; node ids are filled in below
            alignment_read[n]  = j == next_j ? -1 : j - 1;
        }
        n++;
        i = next_i;
        j = next_j;
    }
    if (n >= bound) n = kNwLoopFailed;
    wave_sync();
    if (n == kNwNeedScoreRows) return n;
    for (int32_t k0 = lane; k0 < n; k0 += 4 * kWave) // 4 independent load chains per lane in flight
    {
        int32_t pos[4], node[4];
#pragma unroll
        for (int u = 0; u < 4; u++) pos[u] = (k0 + u * kWave < n) ? alignment_graph[k0 + u * kWave] : -1;
#pragma unroll
        for (int u = 0; u < 4; u++) node[u] = (int32_t)g.sorted_poa[max(pos[u], 0)];
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (pos[u] >= 0) alignment_graph[k0 + u * kWave] = node[u];
    }
    wave_sync();
    return n;
}

} // namespace gwhip
