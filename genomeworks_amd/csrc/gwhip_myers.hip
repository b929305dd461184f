// gwhip_myers.hip -- banded Myers / Ukkonen global alignment for gfx950 (replaces myers_banded_gpu,
// cudaaligner/src/myers_gpu.cu:1149-1204 and the kernel :862-1032).
//
// MI355X design: ONE LANE PER PAIR. The reference gives a 32-lane warp to every pair and spreads the band's
// 32-bit words over the lanes (cross-lane add-with-carry and shifts); at the benchmark sizes the band is 1-4
// words, so 28-31 of its 32 lanes idle. Here each lane walks its own pair with the classic blocked form of
// Myers' algorithm: the band's words are advanced in order, chained by the horizontal delta (-1/0/+1) of the
// previous word. Both forms compute the same DP column, hence the same pv/mv bit-vectors, scores and
// backtrace (checked bit for bit against oracle/aligner_oracle.c, which keeps the reference's own warp
// decomposition). 64 pairs per wavefront, no cross-lane traffic at all; pairs are scheduled longest first so
// the lanes of a wave have similar trip counts.
//
// Per pair the kernel keeps pv, mv (uint32) and score (int32) for every (band word, target column) in a
// column-major workspace -- interleaved across the 64 pairs of a wavefront, so that the lanes' stores of one
// (word, column) step form one contiguous 256-byte write -- because the backtrace needs every column (12 B per 32-cell word-column: the
// algorithmic bytes of SURVEY.md 8(d)). Results are first written into the pair's own slot
// [sequence_starts[2i], sequence_starts[2i+2]) -- a run-length encoding never exceeds |q|+|t| entries -- and a
// second pass packs them behind an exclusive scan of the run counts, so the packed order is the input order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/gwhip.h"

namespace gwhip
{
extern thread_local std::string g_last_error;

namespace myers
{
constexpr int kWord        = 32;
constexpr int8_t kMatch = 0, kMismatch = 1, kInsertion = 2, kDeletion = 3; // AlignmentState, cudaaligner.hpp:52-58

struct Band
{
    uint32_t* pv;
    uint32_t* mv;
    int32_t* score;
    int32_t n_rows; // words in the band
    // one-lane kernels: the arrays are interleaved across the 64 lanes of a wave, element k of a lane at word k * 64 (+ lane).
    // Group kernel: a pair's elements are contiguous records {pv, mv, score} (stride 3, mv = pv + 1, score = pv + 2), so
    // the eight lanes of a group store one 96-byte run per column instead of 24 words scattered over 24 cache lines.
    int32_t stride = 64;
    __device__ __forceinline__ size_t at(int32_t w, int32_t t) const { return ((size_t)t * n_rows + w) * stride; }
};

// Per-lane arrays in LDS, element e of lane l at word e * 64 + l (conflict-free across the lanes of a wave).
// LDS_STATE kernels keep the previous DP column (pv, mv, score per band word) and the query's pattern table there:
// the forward pass then never reads the HBM workspace (it only streams the columns out for the backtrace), so a
// word step costs LDS latency instead of an L2 / HBM round trip on the column it has just written.
struct LaneArray
{
    uint32_t* base; // already offset by the lane
    __device__ __forceinline__ uint32_t& operator[](int32_t e) const { return base[e * 64]; }
};
struct ColumnState
{
    LaneArray pv, mv, score;
};

__device__ __forceinline__ int32_t ceil_div(int32_t a, int32_t b) { return (a + b - 1) / b; }

// workspace need of one pair: `me` words in each of the pv / mv / score arrays (band words of the widest attempt x
// (target + 1) columns, compute_matrix_size_for_alignment, aligner_global_myers_banded.cpp:47-55) and `pw` words of
// query patterns
__host__ __device__ inline void pair_ws_dims(int32_t q, int32_t t, int32_t max_bw, int64_t& me, int32_t& pw)
{
    me = 0;
    pw = 0;
    if (q == 0 || t == 0) return;
    const int32_t pmax = (max_bw + 1) / 2;
    const int32_t bw   = (1 + 2 * pmax) < q ? (1 + 2 * pmax) : q;
    me                 = (int64_t)((bw + 31) / 32) * ((int64_t)t + 1);
    pw                 = ((q + 31) / 32) * 4;
}

// bit pattern of query[offset .. offset+32) == x   (myers_gpu.cu:196-208)
__device__ __forceinline__ uint32_t make_pattern(char x, const char* query, int32_t query_size, int32_t offset)
{
    const int32_t n = min(query_size - offset, kWord);
    uint32_t r      = 0;
    for (int32_t i = 0; i < n; ++i) r |= (uint32_t)(query[offset + i] == x) << i;
    return r;
}

// shifted view on the pattern table (myers_gpu.cu:210-241); index (c >> 1) & 3 => A, C, T, G
template <typename Table>
__device__ __forceinline__ uint32_t get_pattern(const Table& patterns, int32_t n_words, int32_t idx, int32_t begin, char x)
{
    const int32_t ci     = ((unsigned char)x >> 1) & 3;
    const int32_t io     = begin / kWord;
    const int32_t shift  = begin % kWord;
    uint32_t r           = (idx + io < n_words) ? patterns[(idx + io) * 4 + ci] : 0u;
    if (shift != 0)
    {
        r >>= shift;
        if (idx + io + 1 < n_words) r |= patterns[(idx + io + 1) * 4 + ci] << (kWord - shift);
    }
    return r;
}

// score of cell (i, j), i in [1, band rows] (myers_gpu.cu:243-255)
__device__ __forceinline__ int32_t cell_score(const Band& b, int32_t i, int32_t j, uint32_t last_mask)
{
    const int32_t w   = (i - 1) / kWord;
    const int32_t bit = (i - 1) % kWord;
    int32_t s         = b.score[b.at(w, j)];
    uint32_t mask     = bit == 31 ? 0u : ((~1u) << bit);
    if (w == b.n_rows - 1) mask &= last_mask;
    s -= __popc(mask & b.pv[b.at(w, j)]);
    s += __popc(mask & b.mv[b.at(w, j)]);
    return s;
}

// One word of one column. hin: horizontal delta entering the word's first row. Returns the delta at `hbit`
// (and at hbit << 1 through out_y, used by the sliding-band step).
__device__ __forceinline__ int32_t advance_word(uint32_t hbit, uint32_t eq, uint32_t& pv, uint32_t& mv, int32_t hin, int32_t* out_y)
{
    const uint32_t xv = eq | mv;
    if (hin < 0) eq |= 1u;
    uint32_t xh = ((eq & pv) + pv) ^ pv;
    xh |= eq;
    uint32_t ph = mv | ~(xh | pv);
    uint32_t mh = pv & xh;
    const int32_t out_x = ((ph & hbit) ? 1 : 0) - ((mh & hbit) ? 1 : 0);
    if (out_y) *out_y = ((ph & (hbit << 1)) ? 1 : 0) - ((mh & (hbit << 1)) ? 1 : 0);
    ph = (ph << 1) | (hin > 0 ? 1u : 0u);
    mh = (mh << 1) | (hin < 0 ? 1u : 0u);
    pv = mh | ~(xv | ph);
    mv = ph & xv;
    return out_x;
}

// The target characters of a pair, 64 at a time through a per-lane LDS buffer (LDS_STATE kernels). A per-column global
// load -- even one requested a column ahead -- has to wait for the column's own pv / mv / score stores (loads and stores
// share the in-order vmcnt counter), i.e. for an HBM write acknowledgement per column; with the buffer that wait
// happens once per 64 columns. Lanes of a wave walk their targets in step, so they refill together.
struct TargetStream
{
    const char* target;
    int32_t size;
    LaneArray buf; // 16 words per lane
    int32_t lo;    // buf holds target[lo, lo + 64)
    __device__ __forceinline__ void refill(int32_t idx)
    {
        lo = idx;
#pragma unroll
        for (int k = 0; k < 16; ++k)
        {
            uint32_t w = 0;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
            {
                const int32_t p = idx + 4 * k + bb;
                const uint32_t c = p < size ? (uint32_t)(unsigned char)target[p] : 0u;
                w |= c << (8 * bb);
            }
            buf[k] = w;
        }
    }
    __device__ __forceinline__ char at(int32_t idx)
    {
        if (idx < lo || idx >= lo + 64) refill(idx);
        const uint32_t w = buf[(idx - lo) >> 2];
        return (char)((w >> (8 * ((idx - lo) & 3))) & 0xffu);
    }
};

// horizontal stripe: columns [t_begin, t_end), fixed rows (myers_gpu.cu:629-674)
template <bool LDS_STATE, typename Table>
__device__ __forceinline__ void horizontal_band(Band& b, const ColumnState& cs, const Table& patterns, int32_t n_words_query, const char* target,
                                int32_t t_begin, int32_t t_end, int32_t width, int32_t n_words, int32_t pattern_offset,
                                TargetStream* ts = nullptr)
{
    char tc_next = (!LDS_STATE && t_begin < t_end) ? target[t_begin - 1] : 0;
    for (int32_t t = t_begin; t < t_end; ++t)
    {
        int32_t h = 1; // worst case for the top border of the band
        char tc;
        if (LDS_STATE)
            tc = ts->at(t - 1);
        else
        {
            tc = tc_next;
            if (t + 1 < t_end) tc_next = target[t]; // next column's character: its latency overlaps this column
        }
        for (int32_t w = 0; w < n_words; ++w)
        {
            uint32_t pv, mv;
            int32_t sc;
            if (LDS_STATE) { pv = cs.pv[w]; mv = cs.mv[w]; sc = (int32_t)cs.score[w]; }
            else { pv = b.pv[b.at(w, t - 1)]; mv = b.mv[b.at(w, t - 1)]; sc = b.score[b.at(w, t - 1)]; }
            const uint32_t hbit = 1u << (w == n_words - 1 ? width - (n_words - 1) * kWord - 1 : kWord - 1);
            const uint32_t eq   = get_pattern(patterns, n_words_query, w, pattern_offset, tc);
            h                   = advance_word(hbit, eq, pv, mv, h, nullptr);
            sc += h;
            b.score[b.at(w, t)] = sc;
            b.pv[b.at(w, t)]    = pv;
            b.mv[b.at(w, t)]    = mv;
            if (LDS_STATE) { cs.pv[w] = pv; cs.mv[w] = mv; cs.score[w] = (uint32_t)sc; }
        }
    }
}

// diagonal part: the band slides one row per column (myers_gpu.cu:676-751)
template <bool LDS_STATE, typename Table>
__device__ __forceinline__ void diagonal_band(Band& b, const ColumnState& cs, const Table& patterns, int32_t n_words_query, const char* target,
                              int32_t t_begin, int32_t t_end, int32_t band_width, int32_t n_words, int32_t pattern_offset,
                              TargetStream* ts = nullptr)
{
    char tc_next = (!LDS_STATE && t_begin < t_end) ? target[t_begin - 1] : 0;
    for (int32_t t = t_begin; t < t_end; ++t)
    {
        int32_t h = 1;
        char tc;
        if (LDS_STATE)
            tc = ts->at(t - 1);
        else
        {
            tc = tc_next;
            if (t + 1 < t_end) tc_next = target[t];
        }
        // word w of the new column needs words w and w + 1 of the previous one; the column state is updated in
        // place in increasing w, so word w + 1 still holds the previous column when word w is computed
        uint32_t cur_pv, cur_mv;
        if (LDS_STATE) { cur_pv = cs.pv[0]; cur_mv = cs.mv[0]; }
        else { cur_pv = b.pv[b.at(0, t - 1)]; cur_mv = b.mv[b.at(0, t - 1)]; }
        for (int32_t w = 0; w < n_words; ++w)
        {
            uint32_t pv = cur_pv >> 1, mv = cur_mv >> 1;
            int32_t sc;
            if (LDS_STATE) sc = (int32_t)cs.score[w];
            else sc = b.score[b.at(w, t - 1)];
            if (w + 1 < n_words)
            {
                if (LDS_STATE) { cur_pv = cs.pv[w + 1]; cur_mv = cs.mv[w + 1]; }
                else { cur_pv = b.pv[b.at(w + 1, t - 1)]; cur_mv = b.mv[b.at(w + 1, t - 1)]; }
                pv |= cur_pv << (kWord - 1);
                mv |= cur_mv << (kWord - 1);
            }
            const uint32_t eq  = get_pattern(patterns, n_words_query, w, pattern_offset + t - t_begin + 1, tc);
            const uint32_t drb = 1u << (w == n_words - 1 ? band_width - (n_words - 1) * kWord - 2 : kWord - 2);
            const uint32_t ddb = drb << 1;
            if (w == n_words - 1)
            {
                pv |= ddb; // bottom bit has no left neighbour: assume the worst case (+1)
                mv &= ~ddb;
            }
            int32_t hy;
            const int32_t hx   = advance_word(drb, eq, pv, mv, h, &hy);
            const int32_t down = ((pv & ddb) ? 1 : 0) - ((mv & ddb) ? 1 : 0);
            sc += hx + down;
            b.score[b.at(w, t)] = sc;
            b.pv[b.at(w, t)]    = pv;
            b.mv[b.at(w, t)]    = mv;
            if (LDS_STATE) { cs.pv[w] = pv; cs.mv[w] = mv; cs.score[w] = (uint32_t)sc; }
            h = hy; // the horizontal delta of the word's last row enters the next word
        }
    }
}

// the three stripes of one band attempt (myers_compute_scores_edit_dist_banded, myers_gpu.cu:753-846): column 0 is
// already in place; fills diagonal_begin / diagonal_end for the backtrace
template <bool LDS_STATE, typename Table>
__device__ __forceinline__ void banded_stripes(Band& b, const ColumnState& cs, const Table& patterns, int32_t n_words, const char* target,
                                               int32_t query_size, int32_t target_size, int32_t p, int32_t n_words_band,
                                               int32_t band_width, int32_t& diagonal_begin, int32_t& diagonal_end,
                                               TargetStream* ts = nullptr)
{
    const int32_t dlen = abs(target_size - query_size);
    if (band_width >= query_size)
    {
        diagonal_begin = target_size + 1;
        diagonal_end   = target_size + 1;
        horizontal_band<LDS_STATE>(b, cs, patterns, n_words, target, 1, target_size + 1, query_size, n_words_band, 0, ts);
    }
    else
    {
        const int32_t symmetric = (band_width - min(1 + 2 * p + dlen, query_size) == 0) ? 1 : 0;
        diagonal_begin = query_size < target_size ? target_size - query_size + p + 2 : p + 2 + (1 - symmetric);
        diagonal_end   = query_size < target_size ? query_size - p + symmetric : query_size - (query_size - target_size) - p + 1;
        horizontal_band<LDS_STATE>(b, cs, patterns, n_words, target, 1, diagonal_begin, band_width, n_words_band, 0, ts);
        diagonal_band<LDS_STATE>(b, cs, patterns, n_words, target, diagonal_begin, diagonal_end, band_width, n_words_band, 0, ts);
        horizontal_band<LDS_STATE>(b, cs, patterns, n_words, target, diagonal_end, target_size + 1, band_width, n_words_band,
                                   query_size - band_width, ts);
    }
}

// the four forward (A, C, T, G) and four back-to-front pattern words of query word w (myers_preprocess,
// hirschberg_myers_gpu.cu:227-242; make_pattern / make_pattern_reverse in one sweep over the characters)
__device__ __forceinline__ void pattern_words(const char* query, int32_t query_size, int32_t w, uint32_t f[4], uint32_t r[4])
{
    uint32_t pa = 0, pc = 0, pt = 0, pg = 0, ra = 0, rc = 0, rt = 0, rg = 0;
    const int32_t nchar = min(query_size - w * kWord, kWord);
    for (int32_t i = 0; i < nchar; ++i)
    {
        const char a = query[w * kWord + i], z = query[query_size - 1 - (w * kWord + i)];
        pa |= (uint32_t)(a == 'A') << i; pc |= (uint32_t)(a == 'C') << i; pt |= (uint32_t)(a == 'T') << i; pg |= (uint32_t)(a == 'G') << i;
        ra |= (uint32_t)(z == 'A') << i; rc |= (uint32_t)(z == 'C') << i; rt |= (uint32_t)(z == 'T') << i; rg |= (uint32_t)(z == 'G') << i;
    }
    f[0] = pa; f[1] = pc; f[2] = pt; f[3] = pg;
    r[0] = ra; r[1] = rc; r[2] = rt; r[3] = rg;
}

#define GW_EMIT(R)                                                                                                     \
    do                                                                                                                 \
    {                                                                                                                  \
        const int8_t r_ = (R);                                                                                         \
        if (prev_r != r_)                                                                                              \
        {                                                                                                              \
            if (prev_r != -1)                                                                                          \
            {                                                                                                          \
                if (writer)                                                                                            \
                {                                                                                                      \
                    path[pos]   = prev_r;                                                                              \
                    counts[pos] = r_count;                                                                             \
                }                                                                                                      \
                ++pos;                                                                                                 \
            }                                                                                                          \
            prev_r  = r_;                                                                                              \
            r_count = 0;                                                                                               \
        }                                                                                                              \
        ++r_count;                                                                                                     \
    } while (0)

// three-phase backtrace with the implicit worst-case row 0 (myers_gpu.cu:444-627); emits reversed RLE
// `tile` (LDS_STATE kernels; tile_words per lane, 0 = none): the walk reads columns j and j - 1 only and never moves right,
// so the band's pv / mv / score words of the next few columns are fetched together -- one HBM round trip per window of
// columns instead of one per step (a step's nine loads were one dependent round trip each time). All lanes of the wave
// refill together (each for its own position) as soon as one of them runs out.
// G > 1: the G lanes of a pair's group (myers_banded_group_kernel) run the walk together, in lockstep on identical values;
// what they share is the refill of the column window (the loads of a refill are dealt to the lanes), only `writer` stores.
template <int G = 1>
__device__ int32_t backtrace_banded(int8_t* path, int32_t* counts, const Band& b, int32_t diagonal_begin, int32_t diagonal_end,
                                    int32_t band_width, int32_t target_size, LaneArray tile = LaneArray{nullptr}, int32_t tile_words = 0,
                                    bool writer = true)
{
    // One lane: only for narrow bands -- a refill issues 6 instructions per cached (word, column), which a walk through a
    // 6-word band does not win back (measured on configs[1]: 1.04 ms with the window, 0.85 ms without); short-read bands of
    // 1-3 words do. A group shares that cost among its lanes.
    int32_t tile_cols = (tile_words > 0 && b.n_rows > 0) ? tile_words / (3 * b.n_rows) : 0;
    if (tile_cols < (G > 1 ? 8 : 24)) tile_cols = 0;
    int32_t tile_lo = 1, tile_hi = 0; // cached columns [tile_lo, tile_hi] (empty)
    auto refill = [&](int32_t j) {
        tile_hi = j;
        tile_lo = max(0, j - tile_cols + 1);
        const int32_t n  = (tile_hi - tile_lo + 1) * b.n_rows;
        const size_t src = b.at(0, tile_lo); // columns are contiguous in (column, word) order
        constexpr int kU = 8;                // 24 independent loads in flight per lane, then their LDS stores
        for (int32_t e0 = G > 1 ? (int32_t)((threadIdx.x & 63) % G) * kU : 0; e0 < n; e0 += G * kU)
        {
            uint32_t p[kU], m[kU], sc[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u)
            {
                const size_t at = src + (size_t)min(e0 + u, n - 1) * b.stride;
                p[u]  = b.pv[at];
                m[u]  = b.mv[at];
                sc[u] = (uint32_t)b.score[at];
            }
#pragma unroll
            for (int u = 0; u < kU; ++u)
                if (e0 + u < n)
                {
                    tile[3 * (e0 + u) + 0] = p[u];
                    tile[3 * (e0 + u) + 1] = m[u];
                    tile[3 * (e0 + u) + 2] = sc[u];
                }
        }
        if (G > 1) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // the other lanes' part of the window
    };
    // before a step at column j: columns j - 1 and j must be cached (wave-uniform decision over the lanes still walking)
    auto ensure = [&](int32_t j) {
        if (tile_cols < 2) return;
        const bool need = j > tile_hi || max(j - 1, 0) < tile_lo;
        if (__any(need)) refill(j);
    };
    const int32_t out_of_band = INT32_MAX - 1;
    int32_t i = band_width, j = target_size;
    const uint32_t last_mask = band_width % kWord != 0 ? ((1u << (band_width % kWord)) - 1) : ~0u;
    const int32_t last_diag  = diagonal_end < 2 ? out_of_band : cell_score(b, 1, diagonal_end - 2, last_mask) + 2;
    int32_t myscore          = i > 0 ? b.score[b.at((i - 1) / kWord, j)] : 0;
    int32_t pos = 0, r_count = 0;
    int8_t prev_r = -1;
    // The three neighbour scores of a step from five loads instead of nine (round 3). `left` = cell (i2, j2) is read as the
    // reference reads it (word score minus / plus the vertical deltas above the row: cell_score). `diag` is the cell above it
    // in the same column, (i2 - 1, j2): left minus the vertical delta of row i2, a bit of the two words already loaded.
    // `above` is the cell above the current one, whose score the walk carries: myscore minus the vertical delta of the
    // current row in column j_cur (two loads). Within a word these are algebraic identities of cell_score; across a word
    // boundary they hold because the stored word scores and the bit vectors describe the same column (the 10 000- and
    // 1 000 000-pair goldens and the reference vectors check every walk). Neighbours that the reference derives from a
    // formula instead of the matrix are clamped for the fetch and replaced by the callers afterwards; when (i2, j2) itself is
    // outside the band the anchor is (i2 - 1, j2).
    auto fetch3 = [&](int32_t i_cur, int32_t j_cur, int32_t i2, int32_t j2, int32_t& above, int32_t& diag, int32_t& left) {
        const int32_t rows = band_width;
        const int32_t ia = min(max(i_cur, 1), rows), wa = (ia - 1) / kWord, ba = (ia - 1) % kWord;
        const bool in2   = i2 >= 1 && i2 <= rows;
        const int32_t il = in2 ? i2 : min(max(i2 - 1, 1), rows), wl = (il - 1) / kWord, bl = (il - 1) % kWord;
        uint32_t mask    = bl == 31 ? 0u : ((~1u) << bl);
        if (wl == b.n_rows - 1) mask &= last_mask;
        const int32_t ja = max(j_cur, 0), jl = max(j2, 0);
        const size_t xa = b.at(wa, ja), xl = b.at(wl, jl);
        uint32_t pa, na, pl, nl;
        int32_t cl;
        const bool cached = tile_cols >= 2 && min(ja, jl) >= tile_lo && max(ja, jl) <= tile_hi;
        if (cached)
        {
            // element of (word w, column j) in the tile: ((j - tile_lo) * n_rows + w) * 3
            const int32_t ea = ((ja - tile_lo) * b.n_rows + wa) * 3, el = ((jl - tile_lo) * b.n_rows + wl) * 3;
            pa = tile[ea]; na = tile[ea + 1];
            pl = tile[el]; nl = tile[el + 1]; cl = (int32_t)tile[el + 2];
        }
        else
        {
            pa = b.pv[xa]; na = b.mv[xa];
            pl = b.pv[xl]; nl = b.mv[xl]; cl = b.score[xl];
        }
        const int32_t s   = cl - __popc(mask & pl) + __popc(mask & nl);
        const int32_t dvl = (int32_t)((pl >> bl) & 1u) - (int32_t)((nl >> bl) & 1u);
        left  = s;
        diag  = in2 ? s - dvl : s;
        above = myscore - ((int32_t)((pa >> ba) & 1u) - (int32_t)((na >> ba) & 1u));
    };
    while (j >= diagonal_end)
    {
        int32_t above, diag, left;
        ensure(j);
        fetch3(i, j, i, j - 1, above, diag, left);
        if (i <= 1) { above = last_diag + j - diagonal_end; diag = last_diag + j - 1 - diagonal_end; }
        if (i < 1) left = last_diag + j - 1 - diagonal_end;
        int8_t r;
        if (left + 1 == myscore) { r = kInsertion; myscore = left; --j; }
        else if (above + 1 == myscore) { r = kDeletion; myscore = above; --i; }
        else { r = diag == myscore ? kMatch : kMismatch; myscore = diag; --i; --j; }
        GW_EMIT(r);
    }
    while (j >= diagonal_begin)
    {
        int32_t above, diag, left;
        ensure(j);
        fetch3(i, j, i + 1, j - 1, above, diag, left);
        if (i <= 1) above = out_of_band;
        if (i <= 0) diag = j - 1;
        if (i >= band_width) left = out_of_band;
        int8_t r;
        if (left + 1 == myscore) { r = kInsertion; myscore = left; ++i; --j; }
        else if (above + 1 == myscore) { r = kDeletion; myscore = above; --i; }
        else { r = diag == myscore ? kMatch : kMismatch; myscore = diag; --j; }
        GW_EMIT(r);
    }
    while (i > 0 && j > 0)
    {
        int32_t above, diag, left;
        ensure(j);
        fetch3(i, j, i, j - 1, above, diag, left);
        if (i == 1) { above = j; diag = j - 1; }
        if (i > band_width) left = out_of_band;
        int8_t r;
        if (left + 1 == myscore) { r = kInsertion; myscore = left; --j; }
        else if (above + 1 == myscore) { r = kDeletion; myscore = above; --i; }
        else { r = diag == myscore ? kMatch : kMismatch; myscore = diag; --i; --j; }
        GW_EMIT(r);
    }
    auto flush_run = [&]() {
        if (writer)
        {
            path[pos]   = prev_r;
            counts[pos] = r_count;
        }
        ++pos;
    };
    if (i > 0)
    {
        if (prev_r != kDeletion)
        {
            if (prev_r != -1) flush_run();
            prev_r  = kDeletion;
            r_count = 0;
        }
        r_count += i;
    }
    if (j > 0)
    {
        if (prev_r != kInsertion)
        {
            if (prev_r != -1) flush_run();
            prev_r  = kInsertion;
            r_count = 0;
        }
        r_count += j;
    }
    if (r_count != 0) flush_run();
    return pos;
}
#undef GW_EMIT

struct KernelArgs
{
    int32_t n;
    const char* sequences;
    const int64_t* starts;
    const int32_t* max_bandwidths;
    const int32_t* order;       // scheduling order (longest first)
    const int64_t* ws_offsets;  // per wave of 64 slots: first uint32 element of its interleaved workspace region
    int64_t ws_capacity_words;  // words available behind ws (guards against a workspace sized for another order)
    uint32_t* ws;               // [pv | mv | score | patterns] per pair
    int8_t* slot_ops;           // per-pair slots, indexed by sequence offset
    int32_t* slot_counts;
    int32_t* run_counts;        // [n] runs per pair, -1: no result
    uint32_t* metadata;         // [n]
    uint64_t* band_cells;       // optional [n]
    int32_t lds_pattern_words;  // LDS_STATE kernels: per-lane words of the pattern table / of one column-state array
    int32_t lds_band_words;
    int32_t debug_skip;         // profiling ablations (GWHIP_MYERS_SKIP): 1 = no backtrace, 2 = no forward stripes, 4 = no pattern build
    int32_t index_base;         // chunked batches (gwhip_myers_args::index_base): added to the pair index kept in metadata
    int64_t slot_base;          // sequence offset of the (chunk's) first pair: the per-pair result slots are indexed relative to it
};

// per-alignment body of myers_banded_kernel (myers_gpu.cu:897-1021).
// LDS_STATE: dynamic LDS holds, per lane, the query's pattern table (4 words per query word) followed by the column
// state (pv, mv, score for a.lds_band_words band words); the launcher only picks it when every pair of the batch fits.
template <bool LDS_STATE>
__global__ __launch_bounds__(64) void myers_banded_kernel(KernelArgs a)
{
    extern __shared__ uint32_t myers_lds[];
    const int32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    // the wave's workspace region is sized by its largest pair: reduce before any lane leaves
    int64_t me_max = 0;
    int32_t pw_max = 0;
    if (slot < a.n)
    {
        const int32_t i = a.order[slot];
        pair_ws_dims((int32_t)(a.starts[2 * i + 1] - a.starts[2 * i]), (int32_t)(a.starts[2 * i + 2] - a.starts[2 * i + 1]),
                     a.max_bandwidths[i], me_max, pw_max);
    }
    for (int off = 32; off > 0; off >>= 1)
    {
        me_max = max(me_max, (int64_t)__shfl_xor((long long)me_max, off));
        pw_max = max(pw_max, __shfl_xor(pw_max, off));
    }
    if (slot >= a.n) return;
    const int32_t idx        = a.order[slot];
    const char* query        = a.sequences + a.starts[2 * idx];
    const char* target       = a.sequences + a.starts[2 * idx + 1];
    const int32_t query_size  = (int32_t)(a.starts[2 * idx + 1] - a.starts[2 * idx]);
    const int32_t target_size = (int32_t)(a.starts[2 * idx + 2] - a.starts[2 * idx + 1]);
    const int32_t max_bw     = a.max_bandwidths[idx];
    int8_t* path             = a.slot_ops + (a.starts[2 * idx] - a.slot_base);
    int32_t* counts          = a.slot_counts + (a.starts[2 * idx] - a.slot_base);
    const int32_t dlen       = abs(target_size - query_size);
    uint64_t cells           = 0;

    if (max_bw - 1 < dlen && query_size != 0 && target_size != 0)
    {
        a.run_counts[idx] = -1;
        a.metadata[idx]   = (uint32_t)(idx + a.index_base);
        return;
    }
    if (target_size == 0 || query_size == 0)
    {
        if (query_size == 0 && target_size == 0)
            a.run_counts[idx] = 0;
        else
        {
            path[0]           = query_size == 0 ? kInsertion : kDeletion;
            counts[0]         = query_size + target_size;
            a.run_counts[idx] = 1;
        }
        a.metadata[idx] = (uint32_t)(idx + a.index_base) | (1u << 31);
        return;
    }

    // workspace of this pair: sized by the host like compute_matrix_size_for_alignment (aligner_global_myers_banded.cpp:47-55)
    const int32_t n_words   = ceil_div(query_size, kWord);
    const int32_t pmax      = (max_bw + 1) / 2;
    const int64_t max_elems = (int64_t)ceil_div(min(1 + 2 * pmax, query_size), kWord) * ((int64_t)target_size + 1);
    const int64_t region    = a.ws_offsets[blockIdx.x];
    if (region + 64 * (3 * me_max + pw_max) > a.ws_capacity_words) // workspace was sized for a different order
    {
        a.run_counts[idx] = -1;
        a.metadata[idx]   = (uint32_t)(idx + a.index_base);
        return;
    }
    uint32_t* base          = a.ws + region + (threadIdx.x & 63);
    Band b;
    b.pv            = base;
    b.mv            = base + 64 * me_max;
    b.score         = reinterpret_cast<int32_t*>(base + 128 * me_max);
    const LaneArray hbm_patterns{base + 192 * me_max};
    b.n_rows        = 0;
    // pattern table: LDS (per lane) or the pair's HBM workspace
    const LaneArray lds_patterns{myers_lds + (threadIdx.x & 63)};
    ColumnState cs{};
    if (LDS_STATE)
    {
        uint32_t* state = myers_lds + (size_t)a.lds_pattern_words * 64 + (threadIdx.x & 63);
        cs.pv    = LaneArray{state};
        cs.mv    = LaneArray{state + (size_t)a.lds_band_words * 64};
        cs.score = LaneArray{state + (size_t)a.lds_band_words * 128};
    }
    for (int32_t w = 0; w < ((a.debug_skip & 4) ? 0 : n_words); ++w)
    {
        // the four patterns of a query word in one sweep over its characters (make_pattern x 4, one load per character)
        uint32_t pa = 0, pc = 0, pt = 0, pg = 0;
        {
            const int32_t nchar = min(query_size - w * kWord, kWord);
            const char* qw      = query + w * kWord;
            for (int32_t i = 0; i < nchar; ++i)
            {
                const char ch = qw[i];
                pa |= (uint32_t)(ch == 'A') << i;
                pc |= (uint32_t)(ch == 'C') << i;
                pt |= (uint32_t)(ch == 'T') << i;
                pg |= (uint32_t)(ch == 'G') << i;
            }
        }
        if (LDS_STATE)
        {
            lds_patterns[w * 4 + 0] = pa; lds_patterns[w * 4 + 1] = pc; lds_patterns[w * 4 + 2] = pt; lds_patterns[w * 4 + 3] = pg;
        }
        else
        {
            hbm_patterns[w * 4 + 0] = pa; hbm_patterns[w * 4 + 1] = pc; hbm_patterns[w * 4 + 2] = pt; hbm_patterns[w * 4 + 3] = pg;
        }
    }
    int32_t estimate = max(1, dlen + min(target_size, query_size) / 20);
    int32_t diagonal_begin = -1, diagonal_end = -1, band_width = 0;
    for (;;)
    {
        int32_t p      = min(min(target_size, query_size), (estimate - dlen) / 2);
        int32_t bw_new = min(1 + 2 * p + dlen, query_size);
        if (bw_new % kWord == 1 && bw_new != query_size) // at least two bits in the last word
        {
            p += 1;
            bw_new = min(1 + 2 * p + dlen, query_size);
        }
        if (bw_new > max_bw)
        {
            bw_new = max_bw;
            p      = (bw_new - 1 - dlen) / 2;
        }
        const int32_t n_words_band = ceil_div(bw_new, kWord);
        if ((int64_t)n_words_band * (int64_t)(target_size + 1) > max_elems)
        {
            band_width = -band_width;
            break;
        }
        band_width = bw_new;
        b.n_rows   = n_words_band;
        cells += (uint64_t)n_words_band * kWord * (uint64_t)target_size;
        // myers_compute_scores_edit_dist_banded (:753-846)
        for (int32_t w = 0; w < n_words_band; ++w)
        {
            const int32_t s0    = min((w + 1) * kWord, band_width);
            b.pv[b.at(w, 0)]    = ~0u;
            b.mv[b.at(w, 0)]    = 0u;
            b.score[b.at(w, 0)] = s0;
            if (LDS_STATE) { cs.pv[w] = ~0u; cs.mv[w] = 0u; cs.score[w] = (uint32_t)s0; }
        }
        if (a.debug_skip & 2) {}
        else if (LDS_STATE)
        {
            TargetStream ts{target, target_size,
                            LaneArray{myers_lds + (size_t)(a.lds_pattern_words + 3 * a.lds_band_words) * 64 + (threadIdx.x & 63)}, -(1 << 30)};
            banded_stripes<LDS_STATE>(b, cs, lds_patterns, n_words, target, query_size, target_size, p, n_words_band, band_width, diagonal_begin,
                                      diagonal_end, &ts);
        }
        else
            banded_stripes<LDS_STATE>(b, cs, hbm_patterns, n_words, target, query_size, target_size, p, n_words_band, band_width, diagonal_begin, diagonal_end);
        const int32_t dist = n_words_band > 0 ? b.score[b.at(n_words_band - 1, target_size)] : target_size;
        if (dist <= estimate || band_width == query_size) break;
        if (band_width == max_bw)
        {
            band_width = -band_width;
            break;
        }
        estimate *= 2;
    }
    if (band_width != 0 && (a.debug_skip & 1))
    {
        a.run_counts[idx] = 0;
        a.metadata[idx]   = (uint32_t)(idx + a.index_base);
    }
    else if (band_width != 0)
    {
        if (LDS_STATE) // every LDS word of the lane is free after the forward pass: the backtrace's column window
            a.run_counts[idx] = backtrace_banded(path, counts, b, diagonal_begin, diagonal_end, abs(band_width), target_size,
                                                 LaneArray{myers_lds + (threadIdx.x & 63)}, a.lds_pattern_words + 3 * a.lds_band_words + 16);
        else
            a.run_counts[idx] = backtrace_banded(path, counts, b, diagonal_begin, diagonal_end, abs(band_width), target_size);
        a.metadata[idx]   = (uint32_t)(idx + a.index_base) | (band_width > 0 ? (1u << 31) : 0u);
    }
    else
    {
        a.run_counts[idx] = -1;
        a.metadata[idx]   = (uint32_t)(idx + a.index_base);
    }
    if (a.band_cells) a.band_cells[idx] = cells;
}

// ------------------------------------------------------------------------------------------------
// G lanes per pair (small batches of long pairs: BASELINE configs[1] is 10 000 pairs -- 157 wavefronts at one lane per
// pair on 1024 SIMDs, and a pair is one chain of (columns x band words) dependent word steps). Here the band's words of
// a column are advanced by G neighbouring lanes at once, as in the reference's warp decomposition (myers_gpu.cu:
// 257-442: add with carry and shifts across the lanes), so a column costs one word step instead of n_words:
//   * lane k of the group owns band word k; pv / mv / score of the previous column stay in its registers;
//   * the multi-word addition of Myers' Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq is a carry-lookahead over the group: every lane
//     adds its word alone and reports "generates a carry" / "would propagate one"; the two ballots are added as integers
//     (an integer addition ripples carries through runs of propagate bits), which yields every lane's carry-in at once;
//     groups are cut apart by clearing both bits at each group's last lane. The horizontal deltas that the one-lane form
//     chains from word to word (advance_word's hin) are exactly these carries and the bits shifted across the word
//     boundary, so both forms fill identical pv / mv / score columns;
//   * a block is G wavefronts and still owns 64 pairs and the same interleaved workspace region as a wave of the
//     one-lane kernel (pair slot s at offset s), so the host's sizing and the planning kernels do not change;
//   * the pattern table (shared by the group), 64 target characters per pair and the backtrace's column window are in
//     LDS, per pair; the backtrace itself is the one-lane walk, run by the group's first lane.
// Attempts whose band needs more than G words fall back to the one-lane stripes on the group's first lane.
// ------------------------------------------------------------------------------------------------
struct PairTable // pattern table of one pair in LDS
{
    uint32_t* base;
    __device__ __forceinline__ uint32_t operator[](int32_t e) const { return base[e]; }
};

template <int G> struct GroupCtx
{
    int gl;              // lane within the group
    uint64_t top_mask;   // last lane of every group
    uint32_t* tbuf;      // 16 words: target[lo, lo + 64)
    const char* target;
    int32_t target_size;
    int32_t lo;
};

// one column of the band for the whole group: lane k holds word k (pv, mv as they enter the column: for the sliding band
// already shifted down by one row). Returns ph / mh (before the shift) for the deltas.
template <int G>
__device__ __forceinline__ void group_advance(const GroupCtx<G>& c, uint32_t eq, uint32_t& pv, uint32_t& mv, uint32_t& ph_out, uint32_t& mh_out)
{
    const int lane     = threadIdx.x & 63;
    const uint32_t xv  = eq | mv;
    const uint32_t a   = eq & pv;
    const uint32_t s0  = a + pv;
    const uint64_t gen = __ballot(s0 < a), prp = __ballot(s0 == 0xffffffffu);
    const uint64_t A   = (gen | prp) & ~c.top_mask, B = gen & ~c.top_mask;
    const uint64_t cin = (A + B) ^ (prp & ~c.top_mask);
    const uint32_t sum = s0 + (uint32_t)((cin >> lane) & 1u);
    const uint32_t xh  = (sum ^ pv) | eq;
    const uint32_t ph  = mv | ~(xh | pv);
    const uint32_t mh  = pv & xh;
    // bits shifted in from the word below (the band's top border is the worst case: +1)
    // (G = 8: groups lie inside the 16-lane rows; other widths cross them and take the wave-wide shift)
    constexpr int kShr = (16 % G == 0) ? 0x111 : 0x138; // row_shr:1 / wave_shr:1
    uint32_t ph_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)(ph >> 31), kShr, 0xf, 0xf, false);
    uint32_t mh_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)(mh >> 31), kShr, 0xf, 0xf, false);
    if (c.gl == 0)
    {
        ph_lo = 1u;
        mh_lo = 0u;
    }
    const uint32_t phs = (ph << 1) | ph_lo, mhs = (mh << 1) | mh_lo;
    pv     = mhs | ~(xv | phs);
    mv     = phs & xv;
    ph_out = ph;
    mh_out = mh;
}

// The pattern words of a band word, cached in registers (round 3). Band word k of a stripe looks at the query through a
// window that starts `begin` bits into the pattern table; its 32 bits for base ci are the funnel shift of table words
// k + begin / 32 and k + begin / 32 + 1 by begin % 32 (get_pattern above). A lane keeps those two words for all four bases
// (eight registers), so a column costs no LDS round trip: the horizontal stripes (fixed window) shift them once, the sliding
// stripe shifts them by one more bit per column and reloads a word every 32 columns.
struct PatternWindow
{
    uint32_t lo[4], hi[4];
    int32_t word; // table word in lo
    __device__ __forceinline__ void load(const PairTable& patterns, int32_t n_words, int32_t w)
    {
        word = w;
#pragma unroll
        for (int ci = 0; ci < 4; ci++)
        {
            lo[ci] = (w >= 0 && w < n_words) ? patterns[w * 4 + ci] : 0u;
            hi[ci] = (w + 1 >= 0 && w + 1 < n_words) ? patterns[(w + 1) * 4 + ci] : 0u;
        }
    }
};
__device__ __forceinline__ uint32_t select4(uint32_t ci, uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3)
{
    const uint32_t a = (ci & 1u) ? e1 : e0, b = (ci & 1u) ? e3 : e2;
    return (ci & 2u) ? b : a;
}
// target characters four at a time: word (idx >> 2) of the group's 64-character buffer, refilled by the group together
template <int G> struct TargetWords
{
    uint32_t w;
    int32_t q; // idx >> 2 of the cached word
};
template <int G> __device__ __forceinline__ uint32_t group_target_code(GroupCtx<G>& c, TargetWords<G>& tw, int32_t idx)
{
    if ((idx >> 2) != tw.q)
    {
        if (idx < c.lo || idx >= c.lo + 64)
        {
            c.lo = idx & ~3;
            for (int k = c.gl; k < 16; k += G)
            {
                uint32_t w = 0;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb)
                {
                    const int32_t q   = c.lo + 4 * k + bb;
                    const uint32_t ch = (q >= 0 && q < c.target_size) ? (uint32_t)(unsigned char)c.target[q] : 0u;
                    w |= ch << (8 * bb);
                }
                c.tbuf[k] = w;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        tw.q = idx >> 2;
        tw.w = c.tbuf[(idx - c.lo) >> 2];
    }
    return (tw.w >> (8 * (idx & 3) + 1)) & 3u; // (character >> 1) & 3: A, C, T, G
}

template <int G>
__device__ __forceinline__ void group_horizontal_band(Band& b, GroupCtx<G>& c, const PairTable& patterns, int32_t n_words_query, int32_t t_begin,
                                                      int32_t t_end, int32_t width, int32_t n_words, int32_t pattern_offset, uint32_t& pv,
                                                      uint32_t& mv, int32_t& sc)
{
    const int32_t k     = c.gl;
    const uint32_t hbit = 1u << (k == n_words - 1 ? width - (n_words - 1) * kWord - 1 : kWord - 1);
    // the window does not move in this stripe: one funnel shift per base
    PatternWindow pw;
    pw.load(patterns, n_words_query, k + pattern_offset / kWord);
    const uint32_t sh = (uint32_t)(pattern_offset % kWord);
    const uint32_t e0 = __builtin_amdgcn_alignbit(pw.hi[0], pw.lo[0], sh), e1 = __builtin_amdgcn_alignbit(pw.hi[1], pw.lo[1], sh),
                   e2 = __builtin_amdgcn_alignbit(pw.hi[2], pw.lo[2], sh), e3 = __builtin_amdgcn_alignbit(pw.hi[3], pw.lo[3], sh);
    TargetWords<G> tw{0u, INT32_MIN};
    size_t at = b.at(k, t_begin);
    const size_t step = (size_t)b.n_rows * b.stride;
    for (int32_t t = t_begin; t < t_end; ++t, at += step)
    {
        const uint32_t ci = group_target_code<G>(c, tw, t - 1);
        const uint32_t eq = select4(ci, e0, e1, e2, e3);
        uint32_t ph, mh;
        group_advance<G>(c, eq, pv, mv, ph, mh);
        sc += ((ph & hbit) ? 1 : 0) - ((mh & hbit) ? 1 : 0);
        if (k < n_words)
        {
            b.pv[at]    = pv; // one record: a 12-byte store per lane
            b.mv[at]    = mv;
            b.score[at] = sc;
        }
    }
}

template <int G>
__device__ __forceinline__ void group_diagonal_band(Band& b, GroupCtx<G>& c, const PairTable& patterns, int32_t n_words_query, int32_t t_begin,
                                                    int32_t t_end, int32_t band_width, int32_t n_words, int32_t pattern_offset, uint32_t& pv,
                                                    uint32_t& mv, int32_t& sc)
{
    const int32_t k    = c.gl;
    const uint32_t drb = 1u << (k == n_words - 1 ? band_width - (n_words - 1) * kWord - 2 : kWord - 2);
    const uint32_t ddb = drb << 1;
    // the window slides one bit per column: begin = pattern_offset + (t - t_begin) + 1
    int32_t begin = pattern_offset + 1;
    PatternWindow pw;
    pw.load(patterns, n_words_query, k + begin / kWord);
    TargetWords<G> tw{0u, INT32_MIN};
    size_t at = b.at(k, t_begin);
    const size_t step = (size_t)b.n_rows * b.stride;
    for (int32_t t = t_begin; t < t_end; ++t, ++begin, at += step)
    {
        const uint32_t ci = group_target_code<G>(c, tw, t - 1);
        // the band slides one row down: word k takes the low bit of word k + 1 as its top bit
        constexpr int kShl = (16 % G == 0) ? 0x101 : 0x130; // row_shl:1 / wave_shl:1
        const uint32_t pv_up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)pv, kShl, 0xf, 0xf, false);
        const uint32_t mv_up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)mv, kShl, 0xf, 0xf, false);
        pv >>= 1;
        mv >>= 1;
        if (k + 1 < n_words)
        {
            pv |= pv_up << (kWord - 1);
            mv |= mv_up << (kWord - 1);
        }
        if (k == n_words - 1)
        {
            pv |= ddb; // bottom bit has no left neighbour: assume the worst case (+1)
            mv &= ~ddb;
        }
        if (k + begin / kWord != pw.word) pw.load(patterns, n_words_query, k + begin / kWord); // every 32 columns
        const uint32_t sh = (uint32_t)(begin % kWord);
        const uint32_t eq = __builtin_amdgcn_alignbit(select4(ci, pw.hi[0], pw.hi[1], pw.hi[2], pw.hi[3]),
                                                      select4(ci, pw.lo[0], pw.lo[1], pw.lo[2], pw.lo[3]), sh);
        uint32_t ph, mh;
        group_advance<G>(c, eq, pv, mv, ph, mh);
        const int32_t hx   = ((ph & drb) ? 1 : 0) - ((mh & drb) ? 1 : 0);
        const int32_t down = ((pv & ddb) ? 1 : 0) - ((mv & ddb) ? 1 : 0);
        sc += hx + down;
        if (k < n_words)
        {
            b.pv[at]    = pv; // one record: a 12-byte store per lane
            b.mv[at]    = mv;
            b.score[at] = sc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Bands wider than the group (round 6; one pair per wavefront, G = 64): W ROUNDS per column. Lane l owns band words l, 64 + l,
// 128 + l, ...; round j advances words 64 j .. 64 j + 63 across the lanes exactly as group_advance does, and what the one-lane
// form chains from word to word crosses the round boundary as three wave-uniform values: the carry of the multi-word addition
// (out of lane 63 into lane 0 of the next round: the look-ahead sum takes it as its carry-in) and the top bits of ph / mh that the
// shift moves into the next word. The sliding stripe shifts every word down by one bit first, word w taking the low bit of word
// w + 1 -- lane 63 of round j that of lane 0 of round j + 1 -- from the values of the previous column. State (pv, mv, score of W
// words) stays in registers; the pattern words come from the pair's table in LDS (two loads per word and column, all of a
// column's independent of each other). Before this a band of more than 64 words ran on ONE lane: the full-matrix Myers class
// on 32 pairs of 65 536 bases (band attempts of 103 words) took 2.8 s.
// ------------------------------------------------------------------------------------------------
template <int W> struct MultiState
{
    uint32_t pv[W], mv[W];
    int32_t sc[W];
};

// one round: like group_advance<64>, with the chain values of the round below (cin01: carry into lane 0; ph_in / mh_in: bits
// shifted into lane 0's word) and those for the round above
__device__ __forceinline__ void multi_advance(uint32_t eq, uint32_t& pv, uint32_t& mv, uint32_t& ph_out, uint32_t& mh_out, uint32_t& carry,
                                              uint32_t& ph_top, uint32_t& mh_top)
{
    const int lane     = threadIdx.x & 63;
    const uint32_t xv  = eq | mv;
    const uint32_t a   = eq & pv;
    const uint32_t s0  = a + pv;
    const uint64_t gen = __ballot(s0 < a), prp = __ballot(s0 == 0xffffffffu);
    // carries into every lane at once: (gen | prp) + gen + carry-in ripples like the word carries do; bit 63 is kept out of the
    // addition (its carry-out is the next round's carry-in, taken separately)
    const uint64_t top = 1ull << 63;
    const uint64_t A = (gen | prp) & ~top, B = gen & ~top;
    const uint64_t cin = (A + B + (uint64_t)carry) ^ (prp & ~top);
    const uint32_t c_me = (uint32_t)((cin >> lane) & 1u);
    const uint32_t sum = s0 + c_me;
    // carry out of lane 63: generated there, or propagated from its carry-in
    carry = (uint32_t)(((gen >> 63) | ((prp >> 63) & (cin >> 63))) & 1u);
    const uint32_t xh  = (sum ^ pv) | eq;
    const uint32_t ph  = mv | ~(xh | pv);
    const uint32_t mh  = pv & xh;
    uint32_t ph_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)(ph >> 31), 0x138, 0xf, 0xf, false); // wave_shr:1
    uint32_t mh_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)(mh >> 31), 0x138, 0xf, 0xf, false);
    if (lane == 0)
    {
        ph_lo = ph_top;
        mh_lo = mh_top;
    }
    ph_top = (uint32_t)__builtin_amdgcn_readlane((int32_t)(ph >> 31), 63);
    mh_top = (uint32_t)__builtin_amdgcn_readlane((int32_t)(mh >> 31), 63);
    const uint32_t phs = (ph << 1) | ph_lo, mhs = (mh << 1) | mh_lo;
    pv     = mhs | ~(xv | phs);
    mv     = phs & xv;
    ph_out = ph;
    mh_out = mh;
}

// the 32 pattern bits of band word w for base ci when the band's window starts `begin` bits into the query
__device__ __forceinline__ uint32_t multi_eq(const PairTable& patterns, int32_t n_words_query, int32_t w, int32_t begin, uint32_t ci)
{
    const int32_t q   = w + begin / kWord;
    const uint32_t lo = (q >= 0 && q < n_words_query) ? patterns[q * 4 + (int32_t)ci] : 0u;
    const uint32_t hi = (q + 1 >= 0 && q + 1 < n_words_query) ? patterns[(q + 1) * 4 + (int32_t)ci] : 0u;
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(begin % kWord));
}

template <int W>
__device__ __forceinline__ void multi_horizontal_band(Band& b, GroupCtx<64>& c, const PairTable& patterns, int32_t n_words_query, int32_t t_begin,
                                                      int32_t t_end, int32_t width, int32_t n_words, int32_t pattern_offset, MultiState<W>& st)
{
    const int lane = threadIdx.x & 63;
    TargetWords<64> tw{0u, INT32_MIN};
    const size_t step = (size_t)b.n_rows * b.stride;
    size_t at0        = b.at(lane, t_begin);
    for (int32_t t = t_begin; t < t_end; ++t, at0 += step)
    {
        const uint32_t ci = group_target_code<64>(c, tw, t - 1);
        uint32_t carry = 0u, ph_top = 1u, mh_top = 0u; // the band's top border is the worst case: +1
#pragma unroll
        for (int j = 0; j < W; j++)
        {
            const int32_t w   = j * 64 + lane;
            const uint32_t eq = multi_eq(patterns, n_words_query, w, pattern_offset, ci);
            uint32_t ph, mh;
            multi_advance(eq, st.pv[j], st.mv[j], ph, mh, carry, ph_top, mh_top);
            const uint32_t hbit = 1u << (w == n_words - 1 ? width - (n_words - 1) * kWord - 1 : kWord - 1);
            st.sc[j] += ((ph & hbit) ? 1 : 0) - ((mh & hbit) ? 1 : 0);
            if (w < n_words)
            {
                const size_t at = at0 + (size_t)(j * 64) * b.stride;
                b.pv[at]    = st.pv[j];
                b.mv[at]    = st.mv[j];
                b.score[at] = st.sc[j];
            }
        }
    }
}

template <int W>
__device__ __forceinline__ void multi_diagonal_band(Band& b, GroupCtx<64>& c, const PairTable& patterns, int32_t n_words_query, int32_t t_begin,
                                                    int32_t t_end, int32_t band_width, int32_t n_words, int32_t pattern_offset, MultiState<W>& st)
{
    const int lane = threadIdx.x & 63;
    TargetWords<64> tw{0u, INT32_MIN};
    const size_t step = (size_t)b.n_rows * b.stride;
    size_t at0        = b.at(lane, t_begin);
    int32_t begin     = pattern_offset + 1; // the window slides one bit per column
    for (int32_t t = t_begin; t < t_end; ++t, ++begin, at0 += step)
    {
        const uint32_t ci = group_target_code<64>(c, tw, t - 1);
        // the band slides one row down: word w takes the low bit of word w + 1 (of the previous column) as its top bit
#pragma unroll
        for (int j = 0; j < W; j++)
        {
            const int32_t w = j * 64 + lane;
            uint32_t pv_up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)st.pv[j], 0x130, 0xf, 0xf, false); // wave_shl:1
            uint32_t mv_up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)st.mv[j], 0x130, 0xf, 0xf, false);
            if (j + 1 < W) // lane 63's neighbour is lane 0 of the round above (still the previous column's value)
            {
                const int jn      = j + 1 < W ? j + 1 : j; // (a constant once the loop is unrolled)
                const uint32_t pn = (uint32_t)__builtin_amdgcn_readlane((int32_t)st.pv[jn], 0);
                const uint32_t mn = (uint32_t)__builtin_amdgcn_readlane((int32_t)st.mv[jn], 0);
                if (lane == 63)
                {
                    pv_up = pn;
                    mv_up = mn;
                }
            }
            uint32_t pv = st.pv[j] >> 1, mv = st.mv[j] >> 1;
            if (w + 1 < n_words)
            {
                pv |= pv_up << (kWord - 1);
                mv |= mv_up << (kWord - 1);
            }
            if (w == n_words - 1)
            {
                const uint32_t ddb = 2u << (band_width - (n_words - 1) * kWord - 2);
                pv |= ddb; // bottom bit has no left neighbour: assume the worst case (+1)
                mv &= ~ddb;
            }
            st.pv[j] = pv;
            st.mv[j] = mv;
        }
        uint32_t carry = 0u, ph_top = 1u, mh_top = 0u;
#pragma unroll
        for (int j = 0; j < W; j++)
        {
            const int32_t w   = j * 64 + lane;
            const uint32_t eq = multi_eq(patterns, n_words_query, w, begin, ci);
            uint32_t ph, mh;
            multi_advance(eq, st.pv[j], st.mv[j], ph, mh, carry, ph_top, mh_top);
            const uint32_t drb = 1u << (w == n_words - 1 ? band_width - (n_words - 1) * kWord - 2 : kWord - 2);
            const uint32_t ddb = drb << 1;
            const int32_t hx   = ((ph & drb) ? 1 : 0) - ((mh & drb) ? 1 : 0);
            const int32_t down = ((st.pv[j] & ddb) ? 1 : 0) - ((st.mv[j] & ddb) ? 1 : 0);
            st.sc[j] += hx + down;
            if (w < n_words)
            {
                const size_t at = at0 + (size_t)(j * 64) * b.stride;
                b.pv[at]    = st.pv[j];
                b.mv[at]    = st.mv[j];
                b.score[at] = st.sc[j];
            }
        }
    }
}

// one band attempt of n_words_band <= 64 W words on the whole wavefront; returns the last word's score in the last column
template <int W>
__device__ __forceinline__ int32_t multi_attempt(Band& b, GroupCtx<64>& c, const PairTable& patterns, int32_t n_words, int32_t query_size,
                                                 int32_t target_size, int32_t p, int32_t dlen, int32_t n_words_band, int32_t band_width,
                                                 int32_t& diagonal_begin, int32_t& diagonal_end)
{
    const int lane = threadIdx.x & 63;
    MultiState<W> st;
#pragma unroll
    for (int j = 0; j < W; j++)
    {
        const int32_t w = j * 64 + lane;
        st.pv[j] = ~0u;
        st.mv[j] = 0u;
        st.sc[j] = min((w + 1) * kWord, band_width);
        if (w < n_words_band)
        {
            const size_t at = b.at(w, 0);
            b.pv[at]    = st.pv[j];
            b.mv[at]    = st.mv[j];
            b.score[at] = st.sc[j];
        }
    }
    c.lo = -(1 << 30);
    if (band_width >= query_size)
    {
        diagonal_begin = target_size + 1;
        diagonal_end   = target_size + 1;
        multi_horizontal_band<W>(b, c, patterns, n_words, 1, target_size + 1, query_size, n_words_band, 0, st);
    }
    else
    {
        const int32_t symmetric = (band_width - min(1 + 2 * p + dlen, query_size) == 0) ? 1 : 0;
        diagonal_begin = query_size < target_size ? target_size - query_size + p + 2 : p + 2 + (1 - symmetric);
        diagonal_end   = query_size < target_size ? query_size - p + symmetric : query_size - (query_size - target_size) - p + 1;
        multi_horizontal_band<W>(b, c, patterns, n_words, 1, diagonal_begin, band_width, n_words_band, 0, st);
        multi_diagonal_band<W>(b, c, patterns, n_words, diagonal_begin, diagonal_end, band_width, n_words_band, 0, st);
        multi_horizontal_band<W>(b, c, patterns, n_words, diagonal_end, target_size + 1, band_width, n_words_band, query_size - band_width, st);
    }
    // the last word's score in the last column: word n_words_band - 1 = round (n - 1) / 64, lane (n - 1) % 64
    int32_t dist = 0;
#pragma unroll
    for (int j = 0; j < W; j++)
    {
        const int32_t v = __shfl(st.sc[j], (n_words_band - 1) % 64); // (every lane takes part in every shuffle)
        if ((n_words_band - 1) / 64 == j) dist = v;
    }
    return dist;
}

// The group kernel's door to it: only a whole wavefront per pair (G = 64) can take a band of more than G words this way. A
// function of its own, NOT inlined: with the 32-round instantiation's ~100 state registers inside the kernel's body the
// ordinary one-word-per-lane stripes of the same kernel lost 20 % (myers_banded at 64 kbp: 67 -> 82 ms); everything goes in and
// out by value.
struct MultiResult
{
    int32_t dist, diagonal_begin, diagonal_end;
};
__device__ __attribute__((noinline)) MultiResult multi_attempt_any(Band b, GroupCtx<64> c, PairTable patterns, int32_t n_words, int32_t query_size,
                                                                   int32_t target_size, int32_t p, int32_t dlen, int32_t n_words_band,
                                                                   int32_t band_width)
{
    MultiResult r{0, -1, -1};
    const int32_t rounds = (n_words_band + 63) / 64;
#define GW_MULTI(W) r.dist = multi_attempt<W>(b, c, patterns, n_words, query_size, target_size, p, dlen, n_words_band, band_width, r.diagonal_begin, r.diagonal_end)
    if (rounds <= 2) GW_MULTI(2);
    else if (rounds <= 4) GW_MULTI(4);
    else if (rounds <= 8) GW_MULTI(8);
    else if (rounds <= 16) GW_MULTI(16);
    else GW_MULTI(32);
#undef GW_MULTI
    return r;
}
template <int G>
__device__ __forceinline__ bool multi_attempt_for_group(const Band& b, const GroupCtx<G>& ctx, const PairTable& patterns, int32_t n_words,
                                                        int32_t query_size, int32_t target_size, int32_t p, int32_t dlen, int32_t n_words_band,
                                                        int32_t band_width, int32_t& diagonal_begin, int32_t& diagonal_end, int32_t& dist)
{
    if constexpr (G == 64)
    {
        if (n_words_band > 64 * 32) return false;
        const MultiResult r = multi_attempt_any(b, ctx, patterns, n_words, query_size, target_size, p, dlen, n_words_band, band_width);
        dist           = r.dist;
        diagonal_begin = r.diagonal_begin;
        diagonal_end   = r.diagonal_end;
        return true;
    }
    else
        return false;
}

// PAIRS pairs per block (a divisor of 64): the workspace regions stay those of 64 slots -- block j works on slots
// [PAIRS j, PAIRS j + PAIRS) of region PAIRS j / 64 -- but a block is PAIRS G / 64 wavefronts, so that a small batch spreads
// over all CUs with one busy wavefront per SIMD (the column step is vector work back to back: two such wavefronts on one
// SIMD halve each other).
// G need not divide 64: six lanes per pair are ten pairs per wavefront (four lanes idle), which puts BASELINE configs[1]'s
// 10 000 pairs on 1 000 wavefronts -- one per SIMD -- where eight lanes per pair make 1 250 and a fifth of the SIMDs carry two.
template <int G, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void myers_banded_group_kernel(KernelArgs a)
{
    extern __shared__ uint32_t myers_lds[];
    constexpr int kPairsPerWave = 64 / G;
    constexpr int PAIRS         = WAVES * kPairsPerWave;
    const int lane  = threadIdx.x & 63;
    const int wave  = threadIdx.x >> 6;
    const int gl    = lane % G;
    const int gbase = lane - gl;                                  // first lane of the group
    const int sb    = wave * kPairsPerWave + min(lane / G, kPairsPerWave - 1); // pair of the block: 0 .. PAIRS - 1
    const int32_t slot = blockIdx.x * PAIRS + sb;
    const int32_t region_index = slot / 64;
    const int s     = slot - region_index * 64;                   // slot inside its 64-slot workspace region
    // the region is sized by its largest pair: every wave reduces over all 64 slots of the region -- of both regions when
    // its pairs straddle a boundary (ten pairs per wavefront do; the wavefront's slots are consecutive, so two at most)
    int64_t me_max = 0;
    int32_t pw_max = 0;
    {
        const int32_t first_slot = blockIdx.x * PAIRS + wave * kPairsPerWave;
        const int32_t r0 = first_slot / 64, r1 = (first_slot + kPairsPerWave - 1) / 64;
        for (int32_t r = r0; r <= r1; ++r) // wave-uniform
        {
            int64_t me = 0;
            int32_t pw = 0;
            const int32_t sj = r * 64 + lane;
            if (sj < a.n)
            {
                const int32_t i = a.order[sj];
                pair_ws_dims((int32_t)(a.starts[2 * i + 1] - a.starts[2 * i]), (int32_t)(a.starts[2 * i + 2] - a.starts[2 * i + 1]),
                             a.max_bandwidths[i], me, pw);
            }
            for (int off = 32; off > 0; off >>= 1)
            {
                me = max(me, (int64_t)__shfl_xor((long long)me, off));
                pw = max(pw, __shfl_xor(pw, off));
            }
            if (r == region_index)
            {
                me_max = me;
                pw_max = pw;
            }
        }
    }
    if (slot >= a.n || lane >= kPairsPerWave * G) return; // (lanes beyond the last whole group are idle)
    const bool leader        = gl == 0;
    const int32_t idx        = a.order[slot];
    const char* query        = a.sequences + a.starts[2 * idx];
    const char* target       = a.sequences + a.starts[2 * idx + 1];
    const int32_t query_size  = (int32_t)(a.starts[2 * idx + 1] - a.starts[2 * idx]);
    const int32_t target_size = (int32_t)(a.starts[2 * idx + 2] - a.starts[2 * idx + 1]);
    const int32_t max_bw     = a.max_bandwidths[idx];
    int8_t* path             = a.slot_ops + (a.starts[2 * idx] - a.slot_base);
    int32_t* counts          = a.slot_counts + (a.starts[2 * idx] - a.slot_base);
    const int32_t dlen       = abs(target_size - query_size);
    uint64_t cells           = 0;

    if (max_bw - 1 < dlen && query_size != 0 && target_size != 0)
    {
        if (leader)
        {
            a.run_counts[idx] = -1;
            a.metadata[idx]   = (uint32_t)(idx + a.index_base);
        }
        return;
    }
    if (target_size == 0 || query_size == 0)
    {
        if (leader)
        {
            if (query_size == 0 && target_size == 0)
                a.run_counts[idx] = 0;
            else
            {
                path[0]           = query_size == 0 ? kInsertion : kDeletion;
                counts[0]         = query_size + target_size;
                a.run_counts[idx] = 1;
            }
            a.metadata[idx] = (uint32_t)(idx + a.index_base) | (1u << 31);
        }
        return;
    }
    const int32_t n_words   = ceil_div(query_size, kWord);
    const int32_t pmax      = (max_bw + 1) / 2;
    const int64_t max_elems = (int64_t)ceil_div(min(1 + 2 * pmax, query_size), kWord) * ((int64_t)target_size + 1);
    const int64_t region    = a.ws_offsets[region_index];
    if (region + 64 * (3 * me_max + pw_max) > a.ws_capacity_words) // workspace was sized for a different order
    {
        if (leader)
        {
            a.run_counts[idx] = -1;
            a.metadata[idx]   = (uint32_t)(idx + a.index_base);
        }
        return;
    }
    // slot s owns words [3 me_max s, 3 me_max (s + 1)) of its region: records {pv, mv, score}, element (column, word) after element
    uint32_t* base = a.ws + region + (size_t)s * 3 * (size_t)me_max;
    Band b;
    b.pv     = base;
    b.mv     = base + 1;
    b.score  = reinterpret_cast<int32_t*>(base + 2);
    b.stride = 3;
    b.n_rows = 0;
    // LDS: per pair the pattern table (odd stride: the lanes of a group read neighbouring words), 17 words of target
    // characters, and the backtrace's column window (interleaved over the 64 pairs like the workspace)
    const int32_t pstride = a.lds_pattern_words + 1;
    uint32_t* pat_lds     = myers_lds + (size_t)sb * pstride;
    uint32_t* tbuf        = myers_lds + (size_t)PAIRS * pstride + (size_t)sb * 17;
    uint32_t* tile_lds    = myers_lds + (size_t)PAIRS * pstride + PAIRS * 17; // interleaved over 64 slots like the workspace (a block uses PAIRS of them)
    const PairTable patterns{pat_lds};
    if (!(a.debug_skip & 4))
        for (int32_t w = gl; w < n_words; w += G)
        {
            uint32_t pa = 0, pc = 0, pt = 0, pg = 0;
            const int32_t nchar = min(query_size - w * kWord, kWord);
            const char* qw      = query + w * kWord;
            for (int32_t i = 0; i < nchar; ++i)
            {
                const char ch = qw[i];
                pa |= (uint32_t)(ch == 'A') << i;
                pc |= (uint32_t)(ch == 'C') << i;
                pt |= (uint32_t)(ch == 'T') << i;
                pg |= (uint32_t)(ch == 'G') << i;
            }
            pat_lds[w * 4 + 0] = pa; pat_lds[w * 4 + 1] = pc; pat_lds[w * 4 + 2] = pt; pat_lds[w * 4 + 3] = pg;
        }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    GroupCtx<G> ctx;
    ctx.gl          = gl;
    ctx.top_mask    = 0;
#pragma unroll
    for (int j = 0; j < kPairsPerWave; j++) ctx.top_mask |= 1ull << (j * G + G - 1);
    ctx.tbuf        = tbuf;
    ctx.target      = target;
    ctx.target_size = target_size;
    ctx.lo          = -(1 << 30);

    int32_t estimate = max(1, dlen + min(target_size, query_size) / 20);
    int32_t diagonal_begin = -1, diagonal_end = -1, band_width = 0;
    for (;;)
    {
        int32_t p      = min(min(target_size, query_size), (estimate - dlen) / 2);
        int32_t bw_new = min(1 + 2 * p + dlen, query_size);
        if (bw_new % kWord == 1 && bw_new != query_size) // at least two bits in the last word
        {
            p += 1;
            bw_new = min(1 + 2 * p + dlen, query_size);
        }
        if (bw_new > max_bw)
        {
            bw_new = max_bw;
            p      = (bw_new - 1 - dlen) / 2;
        }
        const int32_t n_words_band = ceil_div(bw_new, kWord);
        if ((int64_t)n_words_band * (int64_t)(target_size + 1) > max_elems)
        {
            band_width = -band_width;
            break;
        }
        band_width = bw_new;
        b.n_rows   = n_words_band;
        cells += (uint64_t)n_words_band * kWord * (uint64_t)target_size;
        int32_t dist;
        if (n_words_band <= G)
        {
            // column 0, then the three stripes (myers_compute_scores_edit_dist_banded, myers_gpu.cu:753-846)
            uint32_t pv = ~0u, mv = 0u;
            int32_t sc  = min((gl + 1) * kWord, band_width);
            if (gl < n_words_band)
            {
                b.pv[b.at(gl, 0)]    = pv;
                b.mv[b.at(gl, 0)]    = mv;
                b.score[b.at(gl, 0)] = sc;
            }
            ctx.lo = -(1 << 30);
            if (!(a.debug_skip & 2))
            {
                if (band_width >= query_size)
                {
                    diagonal_begin = target_size + 1;
                    diagonal_end   = target_size + 1;
                    group_horizontal_band<G>(b, ctx, patterns, n_words, 1, target_size + 1, query_size, n_words_band, 0, pv, mv, sc);
                }
                else
                {
                    const int32_t symmetric = (band_width - min(1 + 2 * p + dlen, query_size) == 0) ? 1 : 0;
                    diagonal_begin = query_size < target_size ? target_size - query_size + p + 2 : p + 2 + (1 - symmetric);
                    diagonal_end   = query_size < target_size ? query_size - p + symmetric : query_size - (query_size - target_size) - p + 1;
                    group_horizontal_band<G>(b, ctx, patterns, n_words, 1, diagonal_begin, band_width, n_words_band, 0, pv, mv, sc);
                    group_diagonal_band<G>(b, ctx, patterns, n_words, diagonal_begin, diagonal_end, band_width, n_words_band, 0, pv, mv, sc);
                    group_horizontal_band<G>(b, ctx, patterns, n_words, diagonal_end, target_size + 1, band_width, n_words_band,
                                             query_size - band_width, pv, mv, sc);
                }
            }
            // the last word's score in the last column, to every lane of the group
            dist = __shfl(sc, gbase + n_words_band - 1);
            if (a.debug_skip & 2) dist = b.score[b.at(n_words_band - 1, target_size)];
        }
        else if (!(a.debug_skip & (2 | 8)) &&
                 multi_attempt_for_group<G>(b, ctx, patterns, n_words, query_size, target_size, p, dlen, n_words_band, band_width, diagonal_begin,
                                            diagonal_end, dist))
        {
            // a band wider than the wavefront that owns the pair: several words per lane (multi_attempt above; GWHIP_MYERS_SKIP
            // bit 3 sends it down the one-lane stripes below instead, for A/B runs)
        }
        else
        {
            // a band wider than the group: the one-lane stripes on the group's first lane (state in the HBM workspace)
            if (leader)
            {
                for (int32_t w = 0; w < n_words_band; ++w)
                {
                    b.pv[b.at(w, 0)]    = ~0u;
                    b.mv[b.at(w, 0)]    = 0u;
                    b.score[b.at(w, 0)] = min((w + 1) * kWord, band_width);
                }
                ColumnState none{};
                if (!(a.debug_skip & 2))
                    banded_stripes<false>(b, none, patterns, n_words, target, query_size, target_size, p, n_words_band, band_width, diagonal_begin,
                                          diagonal_end);
            }
            diagonal_begin = __shfl(diagonal_begin, gbase);
            diagonal_end   = __shfl(diagonal_end, gbase);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            dist = leader ? b.score[b.at(n_words_band - 1, target_size)] : 0;
            dist = __shfl(dist, gbase);
        }
        if (dist <= estimate || band_width == query_size) break;
        if (band_width == max_bw)
        {
            band_width = -band_width;
            break;
        }
        estimate *= 2;
    }
    if (band_width != 0 && (a.debug_skip & 1))
    {
        if (leader)
        {
            a.run_counts[idx] = 0;
            a.metadata[idx]   = (uint32_t)(idx + a.index_base);
        }
    }
    else if (band_width != 0)
    {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // the group's column stores, before the walk reads them
        // the whole group walks (in lockstep, on the same values) and shares the refills of the column window
        const int32_t runs = backtrace_banded<G>(path, counts, b, diagonal_begin, diagonal_end, abs(band_width), target_size,
                                                 LaneArray{tile_lds + s}, a.lds_band_words, leader);
        if (leader)
        {
            a.run_counts[idx] = runs;
            a.metadata[idx]   = (uint32_t)(idx + a.index_base) | (band_width > 0 ? (1u << 31) : 0u);
        }
    }
    else if (leader)
    {
        a.run_counts[idx] = -1;
        a.metadata[idx]   = (uint32_t)(idx + a.index_base);
    }
    if (leader && a.band_cells) a.band_cells[idx] = cells;
}

// Exclusive scan of max(run_counts, 0) into result_starts[n + 1] in three small launches (block totals, scan of the
// totals by one workgroup, per-block rescan with its offset): a million pairs are 489 blocks instead of 977 rounds of one.
constexpr int kScanBlock = 256, kScanPerThread = 8, kScanChunk = kScanBlock * kScanPerThread;

template <typename T>
__device__ __forceinline__ T block_exclusive_scan(T v, T* total)
{
    // wave scan (DPP-free shuffles: these kernels are not on any critical path), then the wave totals through LDS
    __shared__ T wave_total[kScanBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T incl = v;
    for (int off = 1; off < 64; off <<= 1)
    {
        const T t = (T)__shfl_up((long long)incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    T before = 0, all = 0;
    for (int w = 0; w < kScanBlock / 64; ++w)
    {
        if (w < wave) before += wave_total[w];
        all += wave_total[w];
    }
    __syncthreads();
    *total = all;
    return before + incl - v;
}

__global__ __launch_bounds__(kScanBlock) void scan_block_totals_kernel(const int32_t* run_counts, int32_t* block_totals, int32_t n)
{
    const int32_t base = blockIdx.x * kScanChunk + threadIdx.x * kScanPerThread;
    int32_t sum        = 0;
    for (int k = 0; k < kScanPerThread; ++k)
        if (base + k < n) sum += max(run_counts[base + k], 0);
    int32_t total;
    (void)block_exclusive_scan<int32_t>(sum, &total);
    if (threadIdx.x == 0) block_totals[blockIdx.x] = total;
}

template <typename T>
__global__ __launch_bounds__(1024) void scan_totals_kernel(T* block_totals, int32_t n_blocks, T* grand_total, const T* base = nullptr)
{
    __shared__ T part[1024];
    __shared__ T carry;
    if (threadIdx.x == 0) carry = base ? *base : 0; // chunked batches: the runs of the chunks before this one
    __syncthreads();
    for (int32_t base = 0; base < n_blocks; base += 1024)
    {
        const int32_t i = base + threadIdx.x;
        const T v       = i < n_blocks ? block_totals[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1)
        {
            T t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_blocks) block_totals[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *grand_total = carry;
}

__global__ __launch_bounds__(kScanBlock) void scan_apply_kernel(const int32_t* run_counts, const int32_t* block_offsets, int32_t* result_starts, int32_t n)
{
    const int32_t base = blockIdx.x * kScanChunk + threadIdx.x * kScanPerThread;
    int32_t v[kScanPerThread];
    int32_t sum = 0;
    for (int k = 0; k < kScanPerThread; ++k)
    {
        v[k] = base + k < n ? max(run_counts[base + k], 0) : 0;
        sum += v[k];
    }
    int32_t total;
    int32_t at = block_offsets[blockIdx.x] + block_exclusive_scan<int32_t>(sum, &total);
    for (int k = 0; k < kScanPerThread; ++k)
        if (base + k < n)
        {
            result_starts[base + k] = at;
            at += v[k];
        }
}

// one lane per pair: its runs move from the pair's own slot to the packed position (a few runs for short reads, a few
// dozen at 1 kbp; four copies in flight per lane)
__global__ __launch_bounds__(256) void compact_kernel(KernelArgs a, int8_t* results, int32_t* result_counts, const int32_t* result_starts)
{
    const int32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.n) return;
    const int32_t nr  = max(a.run_counts[idx], 0);
    const int64_t src = a.starts[2 * idx] - a.slot_base;
    const int32_t dst = result_starts[idx];
    int32_t k         = 0;
    for (; k + 4 <= nr; k += 4)
    {
        const int8_t o0 = a.slot_ops[src + k], o1 = a.slot_ops[src + k + 1], o2 = a.slot_ops[src + k + 2], o3 = a.slot_ops[src + k + 3];
        const int32_t c0 = a.slot_counts[src + k], c1 = a.slot_counts[src + k + 1], c2 = a.slot_counts[src + k + 2], c3 = a.slot_counts[src + k + 3];
        results[dst + k] = o0; results[dst + k + 1] = o1; results[dst + k + 2] = o2; results[dst + k + 3] = o3;
        result_counts[dst + k] = c0; result_counts[dst + k + 1] = c1; result_counts[dst + k + 2] = c2; result_counts[dst + k + 3] = c3;
    }
    for (; k < nr; ++k)
    {
        results[dst + k]       = a.slot_ops[src + k];
        result_counts[dst + k] = a.slot_counts[src + k];
    }
}

// the call's packed runs [result_starts[0], result_starts[n]) also go to the caller's pinned host mirrors (entries below the
// capacity): wavefront-wide consecutive stores, so the link sees whole lines
__global__ __launch_bounds__(256) void mirror_runs_kernel(const int8_t* results, const int32_t* result_counts, const int32_t* result_starts, int32_t n,
                                                          int8_t* results_host, int32_t* result_counts_host, int64_t capacity,
                                                          const uint32_t* metadata, int32_t* result_starts_host, uint32_t* metadata_host)
{
    if (result_starts_host != nullptr)
        for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x)
        {
            result_starts_host[i] = result_starts[i];
            if (i < n && metadata_host != nullptr) metadata_host[i] = metadata[i];
        }
    if (results_host == nullptr || capacity <= 0) return;
    const int64_t first = result_starts[0], last = min((int64_t)result_starts[n], capacity);
    const int64_t step  = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < last; j += step) result_counts_host[j] = result_counts[j];
    // the operations four per lane where the words are whole
    const int64_t w0 = (first + 3) & ~int64_t(3), w1 = last & ~int64_t(3);
    if (w0 < w1)
    {
        for (int64_t j = w0 + 4 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x); j < w1; j += 4 * step)
            *reinterpret_cast<uint32_t*>(results_host + j) = *reinterpret_cast<const uint32_t*>(results + j);
        if (blockIdx.x == 0 && threadIdx.x < 8)
        {
            const int64_t j = threadIdx.x < 4 ? first + threadIdx.x : w1 + (threadIdx.x - 4);
            if ((threadIdx.x < 4 && j < w0) || (threadIdx.x >= 4 && j < last)) results_host[j] = results[j];
        }
    }
    else
        for (int64_t j = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < last; j += step) results_host[j] = results[j];
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// fixed part of the workspace (depends on n and the total sequence length only), then the per-pair matrices
struct WsPlan
{
    size_t off_ws_offsets, off_run_counts, off_slot_ops, off_slot_counts, off_cells, off_identity, off_scan, off_ws;
};

static WsPlan plan_fixed(int32_t n, int64_t total_len)
{
    WsPlan p{};
    size_t off = 0;
    auto take  = [&](size_t b) { size_t o = off; off = align_up(off + b, 256); return o; };
    p.off_ws_offsets  = take(((size_t)n + 1) * 8);
    p.off_run_counts  = take((size_t)n * 4);
    p.off_slot_ops    = take((size_t)total_len + 16);
    p.off_slot_counts = take(((size_t)total_len + 16) * 4);
    p.off_cells       = take((size_t)n * 8);
    p.off_identity    = take((size_t)n * 4);
    p.off_scan        = take(((size_t)n / kScanChunk + 2) * 8);
    p.off_ws          = off;
    return p;
}

// per-wave workspace sizes (64 slots of the processing order each): one lane per slot and a maximum over the wavefront, then the
// same three-launch exclusive scan as the run counts (64-bit: a million pairs need more than 2^31 workspace words)
__global__ __launch_bounds__(256) void ws_sizes_kernel(const int64_t* starts, const int32_t* max_bws, const int32_t* order,
                                                       int64_t* sizes, int32_t* identity, int32_t n)
{
    const int32_t s = blockIdx.x * 256 + threadIdx.x; // slot; the 64 slots of a workspace region are one wavefront here
    int64_t me      = 0;
    int32_t pw      = 0;
    if (s < n)
    {
        identity[s]     = s;
        const int32_t i = order == identity ? s : order[s];
        pair_ws_dims((int32_t)(starts[2 * i + 1] - starts[2 * i]), (int32_t)(starts[2 * i + 2] - starts[2 * i + 1]), max_bws[i], me, pw);
    }
    for (int o = 32; o > 0; o >>= 1)
    {
        me = max(me, (int64_t)__shfl_xor((long long)me, o));
        pw = max(pw, __shfl_xor(pw, o));
    }
    if ((threadIdx.x & 63) == 0 && s < n) sizes[s >> 6] = 64 * (3 * me + pw);
}

__global__ __launch_bounds__(kScanBlock) void ws_block_totals_kernel(const int64_t* sizes, int64_t* block_totals, int32_t n_waves)
{
    const int32_t base = blockIdx.x * kScanChunk + threadIdx.x * kScanPerThread;
    int64_t sum        = 0;
    for (int k = 0; k < kScanPerThread; ++k)
        if (base + k < n_waves) sum += sizes[base + k];
    int64_t total;
    (void)block_exclusive_scan<int64_t>(sum, &total);
    if (threadIdx.x == 0) block_totals[blockIdx.x] = total;
}

// in place: sizes[w] becomes the first word of wave w's region; sizes[n_waves] the total
__global__ __launch_bounds__(kScanBlock) void ws_apply_kernel(int64_t* sizes, const int64_t* block_offsets, int32_t n_waves)
{
    const int32_t base = blockIdx.x * kScanChunk + threadIdx.x * kScanPerThread;
    int64_t v[kScanPerThread];
    int64_t sum = 0;
    for (int k = 0; k < kScanPerThread; ++k)
    {
        v[k] = base + k < n_waves ? sizes[base + k] : 0;
        sum += v[k];
    }
    int64_t total;
    int64_t at = block_offsets[blockIdx.x] + block_exclusive_scan<int64_t>(sum, &total);
    for (int k = 0; k < kScanPerThread; ++k)
        if (base + k < n_waves)
        {
            sizes[base + k] = at;
            at += v[k];
        }
}

// ------------------------------------------------------------------------------------------------
// Default aligner: Hirschberg's divide and conquer on Myers' bit-vector edit distance
// (replaces hirschberg_myers_gpu, cudaaligner/src/hirschberg_myers_gpu.cu:575-701; constants of
// aligner_global_hirschberg_myers.cpp:32-33). One lane per pair, like the banded kernel: the reference's warp
// only carries the words of one query part (<= a few words for most sub-problems) and serialises everything else
// on lane 0. What decides the output is kept exactly: the LIFO range stack (64 entries, left half pushed first),
// the leaves (empty side, single query character, full Myers matrix + backtrace for queries shorter than 63 when
// the matrix fits max_n_words * 64 elements), the query midpoint len / 2 and the target midpoint as the reference's
// 32 lanes pick it among equal sums (strided first minimum per lane, strict-less shuffle-down tree).
// Workspace per pair (interleaved across the 64 lanes of a wave, element k at word k * 64 + lane):
//   stack[256] | fwd[T + 1] | rev[T + 1] | forward patterns[4 Qw] | reverse patterns[4 Qw] | pv[Qw] | mv[Qw] |
//   leaf pv / mv / score [3 x leaf words]
// ------------------------------------------------------------------------------------------------
constexpr int32_t kHbStackEntries = 64;
constexpr int32_t kHbSwitchToMyers = 63;

struct HirschbergArgs
{
    int32_t n;
    const char* sequences;
    const int64_t* starts;
    int32_t max_query_length;
    int8_t* results;          // per pair: slot at starts[2 i], capacity q + t, states back to front
    int32_t* result_lengths;  // [n]
    uint32_t* ws;
    const int64_t* wave_offsets; // [n_waves + 1] words
    int64_t ws_capacity_words;
    int32_t lds_state_words;     // LDS_STATE kernels: words per lane of one column-state array
    int32_t levels_first;        // hirschberg_levels_kernel has run: the depth-first wave kernel only takes what it left
    int32_t span_levels;         // > 0: the span path takes the pairs whose query exceeds kSpanPartQuery (the other kernels skip them)
};

__host__ __device__ inline int64_t hb_leaf_words(int32_t t, int64_t max_elems)
{
    const int64_t two_cols = 2 * ((int64_t)t + 1);
    return two_cols < max_elems ? two_cols : max_elems;
}
// words of one lane's workspace for a wave whose longest query has qw words and longest target t characters
__host__ __device__ inline int64_t hb_lane_words(int32_t qw, int32_t t, int64_t max_elems)
{
    return 4 * kHbStackEntries + 2 * ((int64_t)t + 1) + 10 * (int64_t)qw + 3 * hb_leaf_words(t, max_elems);
}

__global__ __launch_bounds__(1024) void hb_offsets_kernel(const int64_t* starts, int64_t* offsets, int32_t n, int64_t max_elems)
{
    __shared__ int64_t part[1024];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int32_t n_waves = (n + 63) / 64;
    for (int32_t base = 0; base < n_waves; base += 1024)
    {
        const int32_t wv = base + threadIdx.x;
        int64_t v        = 0;
        if (wv < n_waves)
        {
            int32_t qw = 0, tm = 0;
            for (int32_t i = wv * 64; i < min(n, wv * 64 + 64); i++)
            {
                qw = max(qw, ((int32_t)(starts[2 * i + 1] - starts[2 * i]) + kWord - 1) / kWord);
                tm = max(tm, (int32_t)(starts[2 * i + 2] - starts[2 * i + 1]));
            }
            v = 64 * hb_lane_words(qw, tm, max_elems);
        }
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1)
        {
            int64_t t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (wv < n_waves) offsets[wv] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[n_waves] = carry;
}

// bit i of the result: reversed_query[offset + i] == x, reversed_query = the query read back to front
__device__ __forceinline__ uint32_t make_pattern_reverse(char x, const char* query, int32_t query_size, int32_t offset)
{
    const int32_t n = min(query_size - offset, kWord);
    uint32_t r      = 0;
    for (int32_t i = 0; i < n; ++i) r |= (uint32_t)(query[query_size - 1 - (offset + i)] == x) << i;
    return r;
}

// LDS_STATE: the pv / mv words of the running column (2 x lds_state_words per lane) live in dynamic LDS.
template <bool LDS_STATE>
__global__ __launch_bounds__(64) void hirschberg_myers_kernel(HirschbergArgs a)
{
    extern __shared__ uint32_t hb_lds[];
    const int32_t lane = threadIdx.x & 63;
    const int32_t idx  = blockIdx.x * 64 + lane;
    const int64_t max_elems = (int64_t)ceil_div(max(a.max_query_length, 1), kWord) * (kHbSwitchToMyers + 1);
    // wave-level workspace geometry (all lanes take part)
    int32_t qw_max = 0, t_max = 0;
    if (idx < a.n)
    {
        qw_max = ceil_div((int32_t)(a.starts[2 * idx + 1] - a.starts[2 * idx]), kWord);
        t_max  = (int32_t)(a.starts[2 * idx + 2] - a.starts[2 * idx + 1]);
    }
    for (int off = 32; off > 0; off >>= 1)
    {
        qw_max = max(qw_max, __shfl_xor(qw_max, off));
        t_max  = max(t_max, __shfl_xor(t_max, off));
    }
    if (idx >= a.n) return;
    const char* query       = a.sequences + a.starts[2 * idx];
    const char* target      = a.sequences + a.starts[2 * idx + 1];
    const int32_t query_size  = (int32_t)(a.starts[2 * idx + 1] - a.starts[2 * idx]);
    const int32_t target_size = (int32_t)(a.starts[2 * idx + 2] - a.starts[2 * idx + 1]);
    int8_t* path            = a.results + a.starts[2 * idx];
    const int64_t region    = a.wave_offsets[blockIdx.x];
    if (region + 64 * hb_lane_words(qw_max, t_max, max_elems) > a.ws_capacity_words)
    {
        a.result_lengths[idx] = 0;
        return;
    }
    uint32_t* base = a.ws + region + lane;
    const LaneArray stack{base};
    const LaneArray fwd{stack.base + (size_t)4 * kHbStackEntries * 64};
    const LaneArray rev{fwd.base + ((size_t)t_max + 1) * 64};
    const LaneArray pat_f{rev.base + ((size_t)t_max + 1) * 64};
    const LaneArray pat_r{pat_f.base + (size_t)4 * qw_max * 64};
    const LaneArray hbm_pv{pat_r.base + (size_t)4 * qw_max * 64};
    const LaneArray hbm_mv{hbm_pv.base + (size_t)qw_max * 64};
    const LaneArray st_pv = LDS_STATE ? LaneArray{hb_lds + lane} : hbm_pv;
    const LaneArray st_mv = LDS_STATE ? LaneArray{hb_lds + (size_t)a.lds_state_words * 64 + lane} : hbm_mv;
    const LaneArray part_pat{hb_lds + (size_t)a.lds_state_words * 128 + lane}; // LDS_STATE only: 4 words per part word
    const int64_t leaf_words = hb_leaf_words(t_max, max_elems);
    Band leaf; // full Myers matrices of a leaf (column-major, interleaved)
    leaf.pv    = hbm_mv.base + (size_t)qw_max * 64;
    leaf.mv    = leaf.pv + (size_t)leaf_words * 64;
    leaf.score = reinterpret_cast<int32_t*>(leaf.mv + (size_t)leaf_words * 64);
    leaf.n_rows = 0;

    // pattern tables of the whole query, forward and back to front (myers_preprocess, :227-242)
    const int32_t n_words_query = ceil_div(query_size, kWord);
    for (int32_t w = 0; w < n_words_query; ++w)
    {
        uint32_t f[4], r[4];
        pattern_words(query, query_size, w, f, r);
        pat_f[w * 4 + 0] = f[0]; pat_f[w * 4 + 1] = f[1]; pat_f[w * 4 + 2] = f[2]; pat_f[w * 4 + 3] = f[3];
        pat_r[w * 4 + 0] = r[0]; pat_r[w * 4 + 1] = r[1]; pat_r[w * 4 + 2] = r[2]; pat_r[w * 4 + 3] = r[3];
    }

    // last row of the edit-distance matrix of query[qb, qe) against target[tb, te): out[t], t = 0 .. te - tb
    // (myers_compute_scores with full_score_matrix == false, :278-381); reverse: both read back to front
    auto last_row = [&](int32_t qb, int32_t qe, int32_t tb, int32_t te, bool reverse, const LaneArray& out) {
        const int32_t qn = qe - qb, tn = te - tb;
        const int32_t nw = ceil_div(qn, kWord);
        for (int32_t w = 0; w < nw; ++w) { st_pv[w] = ~0u; st_mv[w] = 0u; }
        const int32_t pattern_offset = reverse ? query_size - qe : qb; // position of the part in the (reversed) query
        if (LDS_STATE) // the part's (shifted) pattern words once, into LDS: the column loop then touches no HBM table
            for (int32_t w = 0; w < nw; ++w)
            {
                const char acgt[4] = {'A', 'C', 'T', 'G'};
                for (int32_t ci = 0; ci < 4; ++ci)
                    part_pat[w * 4 + ci] = reverse ? get_pattern(pat_r, n_words_query, w, pattern_offset, acgt[ci])
                                                   : get_pattern(pat_f, n_words_query, w, pattern_offset, acgt[ci]);
            }
        int32_t sc = qn;
        out[0]     = (uint32_t)sc;
        const uint32_t last_hbit = 1u << (qn - (nw - 1) * kWord - 1);
        char tc_next = reverse ? target[te - 1] : target[tb];
        for (int32_t t = 1; t <= tn; ++t)
        {
            const char tc = tc_next;
            if (t < tn) tc_next = reverse ? target[te - t - 1] : target[tb + t]; // next column's character, overlapped
            int32_t h     = 1; // the implicit first row is 0, 1, 2, ...
            for (int32_t w = 0; w < nw; ++w)
            {
                uint32_t pv = st_pv[w], mv = st_mv[w];
                const uint32_t hbit = w == nw - 1 ? last_hbit : (1u << (kWord - 1));
                const uint32_t eq   = LDS_STATE ? part_pat[w * 4 + (((unsigned char)tc >> 1) & 3)]
                                      : reverse  ? get_pattern(pat_r, n_words_query, w, pattern_offset, tc)
                                                 : get_pattern(pat_f, n_words_query, w, pattern_offset, tc);
                h        = advance_word(hbit, eq, pv, mv, h, nullptr);
                st_pv[w] = pv;
                st_mv[w] = mv;
            }
            sc += h;
            out[t] = (uint32_t)sc;
        }
    };
    // Note: get_pattern masks nothing beyond the part's end; bits above qn in the last word never reach `hbit`.

    // the explicit stack: entries (qb, qe, tb, te) as offsets
    int32_t sp = 0;
    auto push = [&](int32_t qb, int32_t qe, int32_t tb, int32_t te) -> bool {
        if (sp >= kHbStackEntries) return false;
        stack[4 * sp + 0] = (uint32_t)qb; stack[4 * sp + 1] = (uint32_t)qe; stack[4 * sp + 2] = (uint32_t)tb; stack[4 * sp + 3] = (uint32_t)te;
        ++sp;
        return true;
    };
    push(0, query_size, 0, target_size);
    bool ok     = true;
    int32_t len = 0;
    while (ok && sp > 0)
    {
        --sp;
        const int32_t qb = (int32_t)stack[4 * sp + 0], qe = (int32_t)stack[4 * sp + 1];
        const int32_t tb = (int32_t)stack[4 * sp + 2], te = (int32_t)stack[4 * sp + 3];
        const int32_t qn = qe - qb, tn = te - tb;
        if (tn == 0)
        {
            for (int32_t k = 0; k < qn; ++k) path[len + k] = kDeletion;
            len += qn;
        }
        else if (qn == 0)
        {
            for (int32_t k = 0; k < tn; ++k) path[len + k] = kInsertion;
            len += tn;
        }
        else if (qn == 1)
        {
            // hirschberg_myers_single_char_warp (:483-515): right-to-left scan for the first equal character
            const char qc = query[qb];
            int32_t p     = len;
            int32_t t     = te - 1;
            while (t >= tb)
            {
                if (target[t] == qc) { path[p++] = kMatch; --t; break; }
                path[p++] = kInsertion;
                --t;
            }
            if (path[p - 1] != kMatch) path[p - 1] = kMismatch;
            while (t >= tb) { path[p++] = kInsertion; --t; }
            len += tn;
        }
        else
        {
            const int32_t nw = ceil_div(qn, kWord);
            if (qn < kHbSwitchToMyers && (int64_t)(tn + 1) * nw <= max_elems)
            {
                // leaf: full Myers matrix (hirschberg_myers_compute_path, :383-410) + append_myers_backtrace (:124-181)
                leaf.n_rows = nw;
                for (int32_t w = 0; w < nw; ++w)
                {
                    leaf.pv[leaf.at(w, 0)]    = ~0u;
                    leaf.mv[leaf.at(w, 0)]    = 0u;
                    leaf.score[leaf.at(w, 0)] = min((w + 1) * kWord, qn);
                }
                const uint32_t last_hbit = 1u << (qn - (nw - 1) * kWord - 1);
                for (int32_t t = 1; t <= tn; ++t)
                {
                    const char tc = target[tb + t - 1];
                    int32_t h     = 1;
                    for (int32_t w = 0; w < nw; ++w)
                    {
                        uint32_t pv = leaf.pv[leaf.at(w, t - 1)], mv = leaf.mv[leaf.at(w, t - 1)];
                        const uint32_t hbit = w == nw - 1 ? last_hbit : (1u << (kWord - 1));
                        const uint32_t eq   = get_pattern(pat_f, n_words_query, w, qb, tc);
                        h                   = advance_word(hbit, eq, pv, mv, h, nullptr);
                        leaf.score[leaf.at(w, t)] = leaf.score[leaf.at(w, t - 1)] + h;
                        leaf.pv[leaf.at(w, t)]    = pv;
                        leaf.mv[leaf.at(w, t)]    = mv;
                    }
                }
                const uint32_t last_mask = qn % kWord != 0 ? ((1u << (qn % kWord)) - 1) : ~0u;
                int32_t i = qn, j = tn;
                int32_t myscore = leaf.score[leaf.at((i - 1) / kWord, j)];
                while (i > 0 && j > 0)
                {
                    const int32_t above = i == 1 ? j : cell_score(leaf, i - 1, j, last_mask);
                    const int32_t diag  = i == 1 ? j - 1 : cell_score(leaf, i - 1, j - 1, last_mask);
                    const int32_t left  = cell_score(leaf, i, j - 1, last_mask);
                    int8_t r;
                    if (left + 1 == myscore) { r = kInsertion; myscore = left; --j; }
                    else if (above + 1 == myscore) { r = kDeletion; myscore = above; --i; }
                    else { r = diag == myscore ? kMatch : kMismatch; myscore = diag; --i; --j; }
                    path[len++] = r;
                }
                while (i > 0) { path[len++] = kDeletion; --i; }
                while (j > 0) { path[len++] = kInsertion; --j; }
                continue;
            }
            const int32_t qmid = qb + qn / 2;
            last_row(qb, qmid, tb, te, false, fwd);
            last_row(qmid, qe, tb, te, true, rev);
            // hirschberg_myers_compute_target_mid_warp (:461-481): the 32-lane argmin, emulated by this one lane.
            // Lane L of the reference sees t = L, L + 32, ...; the tree keeps the lower lane on equal sums.
            int32_t best_min[32], best_t[32];
            for (int32_t L = 0; L < 32; ++L)
            {
                int32_t cm = INT32_MAX, mp = 0;
                for (int32_t t = L; t <= tn; t += 32)
                {
                    const int32_t sum = (int32_t)fwd[t] + (int32_t)rev[tn - t];
                    if (sum < cm) { cm = sum; mp = t; }
                }
                best_min[L] = cm;
                best_t[L]   = mp;
            }
            for (int32_t step = 16; step > 0; step >>= 1)
                for (int32_t L = 0; L + step < 32; ++L) // ascending L reads partners that this step has not updated yet
                    if (best_min[L + step] < best_min[L]) { best_min[L] = best_min[L + step]; best_t[L] = best_t[L + step]; }
            const int32_t tmid = tb + best_t[0];
            ok = ok && push(qb, qmid, tb, tmid);
            ok = ok && push(qmid, qe, tmid, te);
        }
    }
    a.result_lengths[idx] = ok ? len : 0;
}

// ------------------------------------------------------------------------------------------------
// The default aligner with ONE WAVEFRONT PER PAIR (round 3). The one-lane kernel above walks a query part's words one
// after the other for every target character -- 32 dependent word steps per column at 1 kbp, 64 pairs per wavefront,
// so 2 000 pairs are 32 wavefronts on 1 024 SIMDs. Here a pair owns a wavefront (the reference gives it a warp): lane l
// holds band word l of the running column in registers, the multi-word addition of the column step is a carry lookahead
// over the lanes (two ballots, one 64-bit scalar add: the scheme of the group kernel with a group of 64), the bits that
// cross a word boundary move with one DPP shift, the last word's horizontal delta comes back through v_readlane. Query
// parts of more than 64 words (2 048 bases) run in chunks of 64 words with the three boundary bits (addition carry,
// +1 / -1 delta) handed from chunk to chunk as scalars and their column state and pattern words in LDS. Everything that
// decides the output is the one-lane kernel's (and the reference's): range stack, leaves, query midpoint, the 32-lane
// argmin of the target midpoint (now on 32 real lanes), so the two kernels give identical paths
// (test_default_aligner_kernels_agree, and the oracle tests run on this kernel).
// Workspace: the region of 64 pairs is that of the one-lane kernel (same sizing), a pair's arrays are contiguous in it.
// ------------------------------------------------------------------------------------------------
struct PlainWords
{
    const uint32_t* p;
    __device__ __forceinline__ uint32_t operator[](int32_t e) const { return p[e]; }
};

// limits, LDS layout and eligibility test of hirschberg_levels_kernel (below, after the depth-first wave kernel)
constexpr int32_t kSpanPartQuery = 2048; // span path (below): parts are split across blocks until their query piece is at most this long
constexpr int32_t kSpanMaxPairs  = 64;   // ... for batches of at most this many pairs
constexpr int32_t kLvMaxQuery = 2048;
constexpr int32_t kLvMaxParts = 64;
constexpr int32_t kLvMaxTerm  = 256;
constexpr int32_t kLvFlagRedo = -2;
constexpr int32_t kLvStageBytes = 1536; // path pieces of the terminals of one round
constexpr int32_t kLvRound      = 8;    // terminals worked on side by side, one lane each
struct LvPart { int32_t qb, qe, tb, te; };
struct LvLayout
{
    int32_t pat_f, pat_r, tgt, rows_f, rows_r, parts, term, keys, order, segtab, stage, total; // byte offsets
    int32_t row_entries, leaf_elems;
};
__host__ __device__ inline int32_t lv_target_cap(int32_t max_query) { return (max_query + max_query / 8 + 64 + 3) & ~3; }
__host__ __device__ inline LvLayout lv_layout(int32_t max_query)
{
    LvLayout L{};
    const int32_t qw = (max_query + kWord - 1) / kWord, tcap = lv_target_cap(max_query);
    int32_t off = 0;
    auto take   = [&](int32_t bytes) { const int32_t o = off; off = (off + bytes + 15) & ~15; return o; };
    L.pat_f       = take(qw * 16);
    L.pat_r       = take(qw * 16);
    L.tgt         = take(tcap + 4);
    L.row_entries = tcap + kLvMaxParts + 8;
    L.rows_f      = take(L.row_entries * 2);
    L.rows_r      = take(L.row_entries * 2);
    L.parts       = take(2 * kLvMaxParts * 16);
    L.leaf_elems  = (off - L.rows_f) / 12; // the leaves' matrices reuse the rows and the part lists (dead by then)
    L.term        = take(kLvMaxTerm * 16);
    L.keys        = take(kLvMaxTerm * 4);
    L.order       = take(kLvMaxTerm * 2);
    L.segtab      = take(64 * 4);
    L.stage       = take(kLvStageBytes);
    L.total       = off;
    return L;
}
__host__ __device__ inline bool lv_eligible(int32_t query_size, int32_t target_size, int32_t max_query)
{
    return max_query <= kLvMaxQuery && query_size <= max_query && target_size <= lv_target_cap(max_query) - 4;
}

constexpr int32_t kHwLeafElems = 320; // (word, column) elements of a leaf whose matrices stay in LDS (three arrays of that size)

// What one wavefront needs to align (a part of) one pair depth-first: the pair's sequences, the whole query's pattern tables,
// rows and leaf matrices in HBM, range stack / chunked column state / leaf matrices in LDS. Shared by the one-wavefront-per-pair
// kernel (the part is the whole pair) and the kernels of the span path below (the top of a long pair's tree level by level
// across blocks, then one wavefront per part).
struct HbWaveIn
{
    const char* query;
    const char* target;
    int32_t query_size, target_size;
    uint32_t* fwd;      // [target part + 1] last-row scores, forward half
    uint32_t* rev;      // reversed half
    uint32_t* pat_f;    // [4 x query words] pattern words of the whole query, forward
    uint32_t* pat_r;    // back to front
    Band leaf;          // full Myers matrices of a leaf in HBM (column-major)
    int64_t max_elems;  // (word, column) elements a leaf may have
    uint32_t* stack;    // LDS: range stack, 4 x kHbStackEntries words
    uint32_t* st_pv;    // LDS: chunked column state of a part of more than 64 words
    uint32_t* st_mv;
    uint32_t* st_pt;    // LDS: its pattern words (4 per word)
    uint32_t* leaf_lds; // LDS: the matrices of a leaf that fits kHwLeafElems
    int32_t qb, qe, tb, te; // the part to align
};

__device__ __forceinline__ void hb_build_patterns(const HbWaveIn& in, int32_t lane)
{
    const char* query = in.query;
    const int32_t query_size = in.query_size;
    uint32_t* pat_f = in.pat_f;
    uint32_t* pat_r = in.pat_r;
    // pattern tables of the whole query, forward and back to front, one word per lane at a time
    const int32_t n_words_query = ceil_div(query_size, kWord);
    for (int32_t w = lane; w < n_words_query; w += 64)
    {
        uint32_t f[4], r[4];
        pattern_words(query, query_size, w, f, r);
#pragma unroll
        for (int ci = 0; ci < 4; ci++)
        {
            pat_f[w * 4 + ci] = f[ci];
            pat_r[w * 4 + ci] = r[ci];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

}

// last row of the edit-distance matrix of query[qb, qe) against target[tb, te): out[t], t = 0 .. te - tb
__device__ __forceinline__ void hb_last_row(const HbWaveIn& in, int32_t lane, int32_t qb, int32_t qe, int32_t tb, int32_t te, bool reverse, uint32_t* out)
{
    const char* target = in.target;
    const int32_t query_size = in.query_size;
    const int32_t n_words_query = ceil_div(query_size, kWord);
    const PlainWords tab_f{in.pat_f}, tab_r{in.pat_r};
    uint32_t* st_pv = in.st_pv;
    uint32_t* st_mv = in.st_mv;
    uint32_t* st_pt = in.st_pt;
    auto last_row_impl = [&](auto single_tag, int32_t qb, int32_t qe, int32_t tb, int32_t te, bool reverse, uint32_t* out) {
        constexpr bool SINGLE = decltype(single_tag)::value; // the part fits one chunk of 64 words: state and patterns in registers
        const int32_t qn = qe - qb, tn = te - tb;
        const int32_t nw = ceil_div(qn, kWord), nch = SINGLE ? 1 : ceil_div(nw, 64);
        const int32_t pattern_offset = reverse ? query_size - qe : qb;
        const char acgt[4] = {'A', 'C', 'T', 'G'};
        uint32_t e[4] = {0u, 0u, 0u, 0u}; // single chunk: the lane's pattern word for each base
        uint32_t pv = ~0u, mv = 0u;       // single chunk: the lane's column state
        for (int32_t c = 0; c < nch; ++c)
        {
            const int32_t w = c * 64 + lane;
            uint32_t ew[4];
#pragma unroll
            for (int ci = 0; ci < 4; ci++)
                ew[ci] = w < nw ? (reverse ? get_pattern(tab_r, n_words_query, w, pattern_offset, acgt[ci])
                                           : get_pattern(tab_f, n_words_query, w, pattern_offset, acgt[ci]))
                                : 0u;
            if constexpr (SINGLE)
            {
#pragma unroll
                for (int ci = 0; ci < 4; ci++) e[ci] = ew[ci];
                if (w >= nw) pv = 0u; // lanes past the part: no carry generated or propagated
            }
            else
            {
#pragma unroll
                for (int ci = 0; ci < 4; ci++) st_pt[(c * 4 + ci) * 64 + lane] = ew[ci];
                st_pv[c * 64 + lane] = w < nw ? ~0u : 0u;
                st_mv[c * 64 + lane] = 0u;
            }
        }
        if constexpr (!SINGLE) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const int32_t last_lane  = (nw - 1) & 63;
        const uint32_t last_hbit = 1u << (qn - (nw - 1) * kWord - 1);
        int32_t sc  = qn;
        uint32_t acc = (uint32_t)sc; // out[t] for t == lane (mod 64), flushed 64 columns at a time
        uint32_t tcv = 0;            // target characters of 64 columns, one per lane
        for (int32_t t = 1; t <= tn; ++t)
        {
            if (((t - 1) & 63) == 0)
            {
                const int32_t tt = t + lane; // column of this lane's character
                tcv = tt <= tn ? (uint32_t)(unsigned char)(reverse ? target[te - tt] : target[tb + tt - 1]) : 0u;
            }
            const uint32_t tc = (uint32_t)__builtin_amdgcn_readlane((int32_t)tcv, (t - 1) & 63);
            const uint32_t ci = (tc >> 1) & 3u;
            uint32_t carry = 0u, hin_p = 1u, hin_m = 0u; // the implicit first row is 0, 1, 2, ...
            int32_t h = 0;
            for (int32_t c = 0; c < nch; ++c)
            {
                uint32_t eq;
                if constexpr (SINGLE)
                {
                    // branch-free select on the wave-uniform base index: two scalar masks, three bit-field inserts
                    const uint32_t m1 = 0u - (ci & 1u), m2 = 0u - (ci >> 1);
                    const uint32_t lo = (e[1] & m1) | (e[0] & ~m1), hi = (e[3] & m1) | (e[2] & ~m1);
                    eq = (hi & m2) | (lo & ~m2);
                }
                else
                {
                    pv = st_pv[c * 64 + lane];
                    mv = st_mv[c * 64 + lane];
                    eq = st_pt[(c * 4 + (int32_t)ci) * 64 + lane];
                }
                const uint32_t xv  = eq | mv;
                const uint32_t an  = eq & pv;
                const uint32_t s0  = an + pv;
                const uint64_t gen = __ballot(s0 < an), prp = __ballot(s0 == 0xffffffffu);
                const uint64_t top = 1ull << 63;
                const uint64_t cin = (((gen | prp) & ~top) + (gen & ~top) + carry) ^ (prp & ~top);
                carry              = (uint32_t)(((gen >> 63) | ((prp >> 63) & (cin >> 63))) & 1ull);
                const uint32_t sum = s0 + (uint32_t)((cin >> lane) & 1ull);
                const uint32_t xh  = (sum ^ pv) | eq;
                const uint32_t ph  = mv | ~(xh | pv);
                const uint32_t mh  = pv & xh;
                uint32_t ph_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)(ph >> 31), 0x138, 0xf, 0xf, false); // wave_shr:1
                uint32_t mh_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)(mh >> 31), 0x138, 0xf, 0xf, false);
                if (lane == 0)
                {
                    ph_lo = hin_p;
                    mh_lo = hin_m;
                }
                hin_p = (uint32_t)__builtin_amdgcn_readlane((int32_t)(ph >> 31), 63);
                hin_m = (uint32_t)__builtin_amdgcn_readlane((int32_t)(mh >> 31), 63);
                if (c == nch - 1)
                {
                    const uint32_t php = (uint32_t)__builtin_amdgcn_readlane((int32_t)ph, last_lane);
                    const uint32_t mhp = (uint32_t)__builtin_amdgcn_readlane((int32_t)mh, last_lane);
                    h = ((php & last_hbit) ? 1 : 0) - ((mhp & last_hbit) ? 1 : 0);
                }
                const uint32_t phs = (ph << 1) | ph_lo, mhs = (mh << 1) | mh_lo;
                pv = mhs | ~(xv | phs);
                mv = phs & xv;
                if constexpr (!SINGLE)
                {
                    st_pv[c * 64 + lane] = pv;
                    st_mv[c * 64 + lane] = mv;
                }
            }
            sc += h;
            if ((t & 63) == 0) // columns t - 64 .. t - 1 are complete in `acc`
            {
                out[t - 64 + lane] = acc;
            }
            if (lane == (t & 63)) acc = (uint32_t)sc;
        }
        // the remaining columns: t in [tn & ~63, tn]
        if (lane <= (tn & 63)) out[(tn & ~63) + lane] = acc;
    };
    if (qe - qb <= 64 * kWord) last_row_impl(std::true_type{}, qb, qe, tb, te, reverse, out);
    else last_row_impl(std::false_type{}, qb, qe, tb, te, reverse, out);

}

// hirschberg_myers_compute_target_mid_warp (:461-481) on 32 real lanes: lane L sees t = L, L + 32, ...; the shuffle-down tree
// keeps the lower lane on equal sums. Returns the split column relative to the part's first target column.
__device__ __forceinline__ int32_t hb_target_mid(const uint32_t* fwd, const uint32_t* rev, int32_t tn, int32_t lane)
{
    int32_t cm = INT32_MAX, mp = 0;
    if (lane < 32)
        for (int32_t t = lane; t <= tn; t += 32)
        {
            const int32_t sum = (int32_t)fwd[t] + (int32_t)rev[tn - t];
            if (sum < cm) { cm = sum; mp = t; }
        }
    for (int32_t step = 16; step > 0; step >>= 1)
    {
        const int32_t om = __shfl_down(cm, step, 32), ot = __shfl_down(mp, step, 32);
        if ((lane & 31) + step < 32 && om < cm) { cm = om; mp = ot; }
    }
    return __builtin_amdgcn_readfirstlane(mp);
}

// The depth-first walk of the part's tree; the path is appended to `path` back to front. Returns its length, -1 when the range
// stack overflows.
__device__ __forceinline__ int32_t hb_wave_run(const HbWaveIn& in, int8_t* path, int32_t lane)
{
    const char* query = in.query;
    const char* target = in.target;
    const int32_t query_size = in.query_size;
    const int32_t n_words_query = ceil_div(query_size, kWord);
    const int64_t max_elems = in.max_elems;
    const PlainWords tab_f{in.pat_f};
    uint32_t* stack = in.stack;
    uint32_t* leaf_lds = in.leaf_lds;
    uint32_t* fwd = in.fwd;
    uint32_t* rev = in.rev;
    const Band leaf = in.leaf;
    auto last_row = [&](int32_t qb, int32_t qe, int32_t tb, int32_t te, bool reverse, uint32_t* out) { hb_last_row(in, lane, qb, qe, tb, te, reverse, out); };
    int32_t sp = 0;
    auto push = [&](int32_t qb, int32_t qe, int32_t tb, int32_t te) -> bool {
        if (sp >= kHbStackEntries) return false;
        if (lane == 0)
        {
            stack[4 * sp + 0] = (uint32_t)qb; stack[4 * sp + 1] = (uint32_t)qe; stack[4 * sp + 2] = (uint32_t)tb; stack[4 * sp + 3] = (uint32_t)te;
        }
        ++sp;
        return true;
    };
    push(in.qb, in.qe, in.tb, in.te);
    bool ok     = true;
    int32_t len = 0;
    while (ok && sp > 0)
    {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        --sp;
        const int32_t qb = __builtin_amdgcn_readfirstlane((int32_t)stack[4 * sp + 0]), qe = __builtin_amdgcn_readfirstlane((int32_t)stack[4 * sp + 1]);
        const int32_t tb = __builtin_amdgcn_readfirstlane((int32_t)stack[4 * sp + 2]), te = __builtin_amdgcn_readfirstlane((int32_t)stack[4 * sp + 3]);
        const int32_t qn = qe - qb, tn = te - tb;
        const int32_t nw = ceil_div(max(qn, 1), kWord);
        const bool is_leaf = qn >= 2 && tn > 0 && qn < kHbSwitchToMyers && (int64_t)(tn + 1) * nw <= max_elems;
        if (tn == 0 || qn == 0)
        {
            // one side empty: a run of deletions / insertions, written by all lanes
            const int32_t count = tn == 0 ? qn : tn;
            const int8_t state  = tn == 0 ? kDeletion : kInsertion;
            for (int32_t k = lane; k < count; k += 64) path[len + k] = state;
            len += count;
        }
        else if (is_leaf && qn != 1)
        {
            // leaf: full Myers matrix (hirschberg_myers_compute_path, :383-410) + append_myers_backtrace (:124-181). The
            // matrices (at most two words per column) live in LDS when they fit, the forward pass runs on lane 0 (a chain of
            // word steps), the walk evaluates its three neighbour cells on three lanes at once.
            Band lf     = leaf;
            lf.n_rows   = nw;
            const bool in_lds = (int64_t)(tn + 1) * nw <= kHwLeafElems;
            if (in_lds)
            {
                lf.pv    = leaf_lds;
                lf.mv    = leaf_lds + kHwLeafElems;
                lf.score = reinterpret_cast<int32_t*>(leaf_lds + 2 * kHwLeafElems);
            }
            if (lane == 0)
            {
                for (int32_t w = 0; w < nw; ++w)
                {
                    lf.pv[lf.at(w, 0)]    = ~0u;
                    lf.mv[lf.at(w, 0)]    = 0u;
                    lf.score[lf.at(w, 0)] = min((w + 1) * kWord, qn);
                }
                const uint32_t last_hbit = 1u << (qn - (nw - 1) * kWord - 1);
                // the leaf's pattern words (query part qb .. qe) for the four bases
                uint32_t lp[2][4];
                const char acgt[4] = {'A', 'C', 'T', 'G'};
                for (int32_t w = 0; w < 2; ++w)
                    for (int ci = 0; ci < 4; ci++) lp[w][ci] = w < nw ? get_pattern(tab_f, n_words_query, w, qb, acgt[ci]) : 0u;
                uint32_t pv0 = ~0u, mv0 = 0u, pv1 = ~0u, mv1 = 0u;
                int32_t s0 = min(kWord, qn), s1 = qn;
                for (int32_t t = 1; t <= tn; ++t)
                {
                    const int32_t ci = (((unsigned char)target[tb + t - 1]) >> 1) & 3;
                    int32_t h = advance_word(nw == 1 ? last_hbit : (1u << (kWord - 1)), lp[0][ci], pv0, mv0, 1, nullptr);
                    s0 += h;
                    lf.score[lf.at(0, t)] = s0;
                    lf.pv[lf.at(0, t)]    = pv0;
                    lf.mv[lf.at(0, t)]    = mv0;
                    if (nw == 2)
                    {
                        h = advance_word(last_hbit, lp[1][ci], pv1, mv1, h, nullptr);
                        s1 += h;
                        lf.score[lf.at(1, t)] = s1;
                        lf.pv[lf.at(1, t)]    = pv1;
                        lf.mv[lf.at(1, t)]    = mv1;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            if (!in_lds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t last_mask = qn % kWord != 0 ? ((1u << (qn % kWord)) - 1) : ~0u;
            int32_t i = qn, j = tn, l = len;
            int32_t myscore = __builtin_amdgcn_readfirstlane(lf.score[lf.at((i - 1) / kWord, j)]);
            while (i > 0 && j > 0)
            {
                // lane 0: above (i - 1, j), lane 1: diagonal (i - 1, j - 1), lane 2: left (i, j - 1)
                const int32_t ci_ = lane == 2 ? i : i - 1, cj_ = lane == 0 ? j : j - 1;
                int32_t v = 0;
                if (lane < 3) v = ci_ == 0 ? cj_ : cell_score(lf, max(ci_, 1), cj_, last_mask); // row 0 of the matrix is 0, 1, 2, ...
                const int32_t above = __builtin_amdgcn_readlane(v, 0), diag = __builtin_amdgcn_readlane(v, 1), left = __builtin_amdgcn_readlane(v, 2);
                int8_t r;
                if (left + 1 == myscore) { r = kInsertion; myscore = left; --j; }
                else if (above + 1 == myscore) { r = kDeletion; myscore = above; --i; }
                else { r = diag == myscore ? kMatch : kMismatch; myscore = diag; --i; --j; }
                if (lane == 0) path[l] = r;
                l++;
            }
            for (int32_t k = lane; k < i; k += 64) path[l + k] = kDeletion;
            l += i;
            for (int32_t k = lane; k < j; k += 64) path[l + k] = kInsertion;
            l += j;
            len = l;
        }
        else if (qn == 1)
        {
            int32_t new_len = len;
            if (lane == 0)
            {
                {
                    // hirschberg_myers_single_char_warp (:483-515): right-to-left scan for the first equal character
                    const char qc = query[qb];
                    int32_t p     = len;
                    int32_t t     = te - 1;
                    while (t >= tb)
                    {
                        if (target[t] == qc) { path[p++] = kMatch; --t; break; }
                        path[p++] = kInsertion;
                        --t;
                    }
                    if (path[p - 1] != kMatch) path[p - 1] = kMismatch;
                    while (t >= tb) { path[p++] = kInsertion; --t; }
                    new_len = len + tn;
                }
            }
            len = __builtin_amdgcn_readfirstlane(new_len);
        }
        else
        {
            const int32_t qmid = qb + qn / 2;
            last_row(qb, qmid, tb, te, false, fwd);
            last_row(qmid, qe, tb, te, true, rev);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int32_t tmid = tb + hb_target_mid(fwd, rev, tn, lane);
            ok = ok && push(qb, qmid, tb, tmid);
            ok = ok && push(qmid, qe, tmid, te);
        }
    }
    return ok ? len : -1;
}

// Span path (defined below the depth-first kernel): sizes of its workspace behind the wave region, and whether it is there. A
// caller that sized the workspace with gwhip_hirschberg_myers_workspace_bytes (same n, same max_query_length) always has it;
// with a smaller workspace the span kernels do nothing and the depth-first kernel keeps the pair.
static_assert(kSpanMaxPairs <= 64, "the span path's region geometry reads one pair per lane of one wavefront (region 0)");
__host__ __device__ inline int32_t hb_span_levels(int32_t max_query)
{
    int32_t L = 0;
    while (((int64_t)kSpanPartQuery << L) < (int64_t)max_query) ++L;
    return L;
}
__host__ __device__ inline int64_t hb_span_words(int32_t qw, int32_t t, int32_t max_query)
{
    const int32_t L = hb_span_levels(max_query);
    if (L == 0) return 0;
    const int64_t P = (int64_t)1 << L, rows = (int64_t)t + P + 1;
    return 8 * P + P + 2 * rows + 6 * rows + ((int64_t)32 * qw + t + 64 + 3) / 4 + 16;
}
__device__ __forceinline__ bool hb_span_fits(const HirschbergArgs& a, int32_t qw_max, int32_t t_max, int64_t max_elems)
{
    return a.wave_offsets[0] + 64 * hb_lane_words(qw_max, t_max, max_elems) + (int64_t)a.n * hb_span_words(qw_max, t_max, a.max_query_length) <=
           a.ws_capacity_words;
}
__global__ __launch_bounds__(64) void hirschberg_wave_kernel(HirschbergArgs a)
{
    extern __shared__ uint32_t hw_lds[];
    const int32_t lane = threadIdx.x & 63;
    const int32_t idx  = blockIdx.x;
    if (idx >= a.n) return;
    const int32_t region_index = idx >> 6, s = idx & 63;
    const int64_t max_elems = (int64_t)ceil_div(max(a.max_query_length, 1), kWord) * (kHbSwitchToMyers + 1);
    // geometry of the 64-pair region (sized by its longest query / target)
    int32_t qw_max = 0, t_max = 0;
    {
        const int32_t sj = region_index * 64 + lane;
        if (sj < a.n)
        {
            qw_max = ceil_div((int32_t)(a.starts[2 * sj + 1] - a.starts[2 * sj]), kWord);
            t_max  = (int32_t)(a.starts[2 * sj + 2] - a.starts[2 * sj + 1]);
        }
        for (int off = 32; off > 0; off >>= 1)
        {
            qw_max = max(qw_max, __shfl_xor(qw_max, off));
            t_max  = max(t_max, __shfl_xor(t_max, off));
        }
    }
    const char* query         = a.sequences + a.starts[2 * idx];
    const char* target        = a.sequences + a.starts[2 * idx + 1];
    const int32_t query_size  = (int32_t)(a.starts[2 * idx + 1] - a.starts[2 * idx]);
    const int32_t target_size = (int32_t)(a.starts[2 * idx + 2] - a.starts[2 * idx + 1]);
    int8_t* path              = a.results + a.starts[2 * idx];
    // pairs the level-by-level kernel has already aligned (it ran first on this stream) are not touched again
    if (a.levels_first && lv_eligible(query_size, target_size, a.max_query_length) && a.result_lengths[idx] != kLvFlagRedo) return;
    // the span path (below) aligns this pair -- unless its workspace is not there, in which case it stays here
    if (a.span_levels > 0 && query_size > kSpanPartQuery && hb_span_fits(a, qw_max, t_max, max_elems)) return;
    const int64_t region      = a.wave_offsets[region_index];
    const int64_t per_pair    = hb_lane_words(qw_max, t_max, max_elems);
    if (region + 64 * per_pair > a.ws_capacity_words)
    {
        if (lane == 0) a.result_lengths[idx] = 0;
        return;
    }
    uint32_t* base  = a.ws + region + (size_t)s * (size_t)per_pair;
    uint32_t* fwd   = base + 4 * kHbStackEntries;
    uint32_t* rev   = fwd + ((size_t)t_max + 1);
    uint32_t* pat_f = rev + ((size_t)t_max + 1);
    uint32_t* pat_r = pat_f + (size_t)4 * qw_max;
    const int64_t leaf_words = hb_leaf_words(t_max, max_elems);
    Band leaf; // full Myers matrices of a leaf (column-major), lane 0 only
    leaf.pv     = pat_r + (size_t)4 * qw_max + (size_t)2 * qw_max;
    leaf.mv     = leaf.pv + (size_t)leaf_words;
    leaf.score  = reinterpret_cast<int32_t*>(leaf.mv + (size_t)leaf_words);
    leaf.n_rows = 0;
    leaf.stride = 1;
    // LDS: range stack | chunked column state (pv, mv) | chunked pattern words of the current part (4 per word)
    uint32_t* stack = hw_lds;
    uint32_t* st_pv = hw_lds + 4 * kHbStackEntries;
    uint32_t* st_mv = st_pv + (size_t)a.lds_state_words * 64; // lds_state_words = chunks of the longest part
    uint32_t* st_pt = st_mv + (size_t)a.lds_state_words * 64;
    uint32_t* leaf_lds = st_pt + (size_t)a.lds_state_words * 256; // kHwLeafLdsWords: the matrices of a leaf that fits

    HbWaveIn in{};
    in.query = query; in.target = target; in.query_size = query_size; in.target_size = target_size;
    in.fwd = fwd; in.rev = rev; in.pat_f = pat_f; in.pat_r = pat_r; in.leaf = leaf; in.max_elems = max_elems;
    in.stack = stack; in.st_pv = st_pv; in.st_mv = st_mv; in.st_pt = st_pt; in.leaf_lds = leaf_lds;
    in.qb = 0; in.qe = query_size; in.tb = 0; in.te = target_size;
    hb_build_patterns(in, lane);
    const int32_t len = hb_wave_run(in, path, lane);
    if (lane == 0) a.result_lengths[idx] = len > 0 ? len : 0;
}

// ------------------------------------------------------------------------------------------------
// The SPAN path (round 4): long single pairs (the reference's BM_SingleAlignment shapes, cudaaligner/benchmarks/main.cpp:39-67:
// one pair of up to 100 000 bases). One wavefront walking such a pair's tree depth-first computes every last row one after the
// other -- 2 s for 100 kbp. But the two last rows of a part (forward half, reversed half) are independent, and so are all parts
// of one level of Hirschberg's tree: the top of the tree is therefore grown LEVEL BY LEVEL ACROSS BLOCKS -- one wavefront per
// (part, half) computes a last row (hb_span_rows_kernel), one per part picks the target midpoint the way the reference's 32
// lanes pick it and writes the two children (hb_span_split_kernel) -- until a part's query piece is at most kSpanPartQuery
// bases; every such part is then aligned depth-first by a wavefront of its own (hb_span_parts_kernel: hb_wave_run from that
// part) and the pieces are joined back to front (hb_span_join_kernel). What decides the output -- query midpoint len / 2, the
// 32-lane argmin with its tie rule, the terminal cases and their order along the path -- is the depth-first kernel's; only the
// order in which independent sub-problems are COMPUTED changes, so the paths are identical
// (test_default_aligner_span_path_equals_the_depth_first_kernel). Taken for batches of at most kSpanMaxPairs pairs whose
// max_query_length exceeds kSpanPartQuery, by the pairs whose query does (the others stay with the kernels above).
// Workspace, per pair, behind the region of the 64-pair wave (hb_span_words): part lists of two consecutive levels, piece
// lengths, the rows / leaf matrices of all parts side by side (part j of a level owns [tb_j + j, te_j + j]: parts tile the
// target), and a staging buffer for the pieces (part j owns [qb_j + tb_j, qe_j + te_j)).
// ------------------------------------------------------------------------------------------------
struct SpanGeom
{
    LvPart* parts[2];
    int32_t* piece_len;
    uint32_t* rows_f;
    uint32_t* rows_r;
    uint32_t* leaf;   // 6 words per row entry: pv | mv | score of the parts' leaves, 2 columns' worth per target column
    int8_t* stage;
    int32_t P, levels;
    int64_t rows;
};
// geometry of the region of the (at most 64) pairs of the batch: longest query in words, longest target -- every block
// recomputes it (64 loads, one shuffle tree)
__device__ __forceinline__ void hb_region_geometry(const HirschbergArgs& a, int32_t lane, int32_t& qw_max, int32_t& t_max)
{
    qw_max = 0;
    t_max  = 0;
    if (lane < a.n)
    {
        qw_max = ceil_div((int32_t)(a.starts[2 * lane + 1] - a.starts[2 * lane]), kWord);
        t_max  = (int32_t)(a.starts[2 * lane + 2] - a.starts[2 * lane + 1]);
    }
    for (int off = 32; off > 0; off >>= 1)
    {
        qw_max = max(qw_max, __shfl_xor(qw_max, off));
        t_max  = max(t_max, __shfl_xor(t_max, off));
    }
}
__device__ __forceinline__ SpanGeom hb_span_geom(const HirschbergArgs& a, int32_t idx, int32_t qw_max, int32_t t_max, int64_t max_elems)
{
    SpanGeom g;
    g.levels = hb_span_levels(a.max_query_length);
    g.P      = 1 << g.levels;
    g.rows   = (int64_t)t_max + g.P + 1;
    uint32_t* area = a.ws + a.wave_offsets[0] + 64 * hb_lane_words(qw_max, t_max, max_elems) + (int64_t)idx * hb_span_words(qw_max, t_max, a.max_query_length);
    g.parts[0]  = reinterpret_cast<LvPart*>(area);
    g.parts[1]  = reinterpret_cast<LvPart*>(area + 4 * g.P);
    g.piece_len = reinterpret_cast<int32_t*>(area + 8 * g.P);
    g.rows_f    = area + 9 * g.P;
    g.rows_r    = g.rows_f + g.rows;
    g.leaf      = g.rows_r + g.rows;
    g.stage     = reinterpret_cast<int8_t*>(g.leaf + 6 * g.rows);
    return g;
}
// the pair's slots in the wave region (as the depth-first kernel carves them): only the pattern tables are used by the span path
__device__ __forceinline__ void hb_span_pair_tables(const HirschbergArgs& a, int32_t idx, int32_t qw_max, int32_t t_max, int64_t max_elems, uint32_t*& pat_f,
                                                    uint32_t*& pat_r)
{
    uint32_t* base = a.ws + a.wave_offsets[0] + (size_t)idx * (size_t)hb_lane_words(qw_max, t_max, max_elems);
    pat_f          = base + 4 * kHbStackEntries + 2 * ((size_t)t_max + 1);
    pat_r          = pat_f + (size_t)4 * qw_max;
}
// the pair takes the span path (and its workspace is there: a caller that sized the workspace with
// gwhip_hirschberg_myers_workspace_bytes always has it; otherwise the pair reports no result, hb_span_join_kernel)
__device__ __forceinline__ bool hb_span_pair(const HirschbergArgs& a, int32_t idx)
{
    return a.span_levels > 0 && (int32_t)(a.starts[2 * idx + 1] - a.starts[2 * idx]) > kSpanPartQuery;
}
// a part that is split at the next level: long enough, and both sides non-empty (an empty side is a terminal: a run of gaps)
__device__ __forceinline__ bool hb_span_splits(const LvPart& p) { return p.qe - p.qb > kSpanPartQuery && p.te > p.tb; }

// grid (n): pattern tables of the pair, level-0 part list
__global__ __launch_bounds__(64) void hb_span_init_kernel(HirschbergArgs a)
{
    const int32_t lane = threadIdx.x & 63, idx = blockIdx.x;
    int32_t qw_max, t_max;
    hb_region_geometry(a, lane, qw_max, t_max);
    if (!hb_span_pair(a, idx)) return;
    const int64_t max_elems = (int64_t)ceil_div(max(a.max_query_length, 1), kWord) * (kHbSwitchToMyers + 1);
    if (!hb_span_fits(a, qw_max, t_max, max_elems)) return;
    const SpanGeom g        = hb_span_geom(a, idx, qw_max, t_max, max_elems);
    HbWaveIn in{};
    in.query      = a.sequences + a.starts[2 * idx];
    in.query_size = (int32_t)(a.starts[2 * idx + 1] - a.starts[2 * idx]);
    hb_span_pair_tables(a, idx, qw_max, t_max, max_elems, in.pat_f, in.pat_r);
    hb_build_patterns(in, lane);
    if (lane == 0) g.parts[0][0] = LvPart{0, in.query_size, 0, (int32_t)(a.starts[2 * idx + 2] - a.starts[2 * idx + 1])};
}

// grid (2 x parts of the level, n): block (2 j + half, pair) computes the last row of one half of part j
__global__ __launch_bounds__(64) void hb_span_rows_kernel(HirschbergArgs a, int32_t level)
{
    extern __shared__ uint32_t hw_lds[];
    const int32_t lane = threadIdx.x & 63, idx = blockIdx.y, j = blockIdx.x >> 1;
    const bool reverse = (blockIdx.x & 1) != 0;
    int32_t qw_max, t_max;
    hb_region_geometry(a, lane, qw_max, t_max);
    if (!hb_span_pair(a, idx)) return;
    const int64_t max_elems = (int64_t)ceil_div(max(a.max_query_length, 1), kWord) * (kHbSwitchToMyers + 1);
    if (!hb_span_fits(a, qw_max, t_max, max_elems)) return;
    const SpanGeom g        = hb_span_geom(a, idx, qw_max, t_max, max_elems);
    const LvPart p          = g.parts[level & 1][j];
    if (!hb_span_splits(p)) return;
    HbWaveIn in{};
    in.query       = a.sequences + a.starts[2 * idx];
    in.target      = a.sequences + a.starts[2 * idx + 1];
    in.query_size  = (int32_t)(a.starts[2 * idx + 1] - a.starts[2 * idx]);
    in.target_size = (int32_t)(a.starts[2 * idx + 2] - a.starts[2 * idx + 1]);
    hb_span_pair_tables(a, idx, qw_max, t_max, max_elems, in.pat_f, in.pat_r);
    in.st_pv = hw_lds + 4 * kHbStackEntries;
    in.st_mv = in.st_pv + (size_t)a.lds_state_words * 64;
    in.st_pt = in.st_mv + (size_t)a.lds_state_words * 64;
    const int32_t qmid = p.qb + (p.qe - p.qb) / 2;
    if (!reverse) hb_last_row(in, lane, p.qb, qmid, p.tb, p.te, false, g.rows_f + p.tb + j);
    else hb_last_row(in, lane, qmid, p.qe, p.tb, p.te, true, g.rows_r + p.tb + j);
}

// The same last row by a PIPELINE OF WAVEFRONTS (block = W wavefronts, grid as above): a half of more than 64 words (2 048 bases)
// is a chain of 64-word chunks per target column -- the addition carry and the +1 / -1 delta at a chunk's top bit feed the next
// chunk -- that one wavefront walks chunk after chunk with the column state in LDS (hb_last_row: 25 chunk steps per column
// for a 100 kbp pair). Here chunk c belongs to wavefront c mod W, which keeps the chunk's column state and pattern words in
// REGISTERS and works on batches of 64 columns: unit (c, k) = chunk c, columns 64 k + 1 .. 64 k + 64, needs (c - 1, k) -- whose
// three boundary bits per column arrive as three 64-bit words through an LDS ring -- and (c, k - 1), the wavefront's own
// previous batch. A wavefront runs its units in the order of c + k, so every dependency lies in an earlier stage (no
// deadlock); done[c] counts the batches chunk c has finished, a producer stays at most kMwRing batches ahead of its consumer.
// The chain of a column is as long as before, but the chunks of DIFFERENT columns overlap: a column costs about one chunk step
// (two when a wavefront owns two chunks) instead of one per chunk. Same bits, same scores as hb_last_row.
constexpr int32_t kMwMaxWaves  = 16;
constexpr int32_t kMwMaxChunks = 64;  // halves of up to 131 072 bases
constexpr int32_t kMwSlots     = kMwMaxChunks / kMwMaxWaves;
constexpr int32_t kMwRing      = 8;
struct MwRowsSync
{
    int32_t done[kMwMaxChunks];
    uint64_t hand[kMwMaxChunks][kMwRing][3];
};
__device__ __forceinline__ int32_t mw_poll(const int32_t* p) { return *reinterpret_cast<const volatile int32_t*>(p); }

__global__ __launch_bounds__(kMwMaxWaves * 64) void hb_span_rows_mw_kernel(HirschbergArgs a, int32_t level)
{
    __shared__ MwRowsSync sync;
    const int32_t lane = threadIdx.x & 63, idx = blockIdx.y, j = blockIdx.x >> 1;
    const int32_t wave = __builtin_amdgcn_readfirstlane((int32_t)threadIdx.x >> 6), W = (int32_t)(blockDim.x >> 6);
    const bool reverse = (blockIdx.x & 1) != 0;
    int32_t qw_max, t_max;
    hb_region_geometry(a, lane, qw_max, t_max);
    if (!hb_span_pair(a, idx)) return;
    const int64_t max_elems = (int64_t)ceil_div(max(a.max_query_length, 1), kWord) * (kHbSwitchToMyers + 1);
    if (!hb_span_fits(a, qw_max, t_max, max_elems)) return;
    const SpanGeom g = hb_span_geom(a, idx, qw_max, t_max, max_elems);
    const LvPart p   = g.parts[level & 1][j];
    if (!hb_span_splits(p)) return;
    HbWaveIn in{};
    in.query       = a.sequences + a.starts[2 * idx];
    in.target      = a.sequences + a.starts[2 * idx + 1];
    in.query_size  = (int32_t)(a.starts[2 * idx + 1] - a.starts[2 * idx]);
    in.target_size = (int32_t)(a.starts[2 * idx + 2] - a.starts[2 * idx + 1]);
    hb_span_pair_tables(a, idx, qw_max, t_max, max_elems, in.pat_f, in.pat_r);
    const int32_t qmid = p.qb + (p.qe - p.qb) / 2;
    const int32_t qb = reverse ? qmid : p.qb, qe = reverse ? p.qe : qmid, tb = p.tb, te = p.te;
    uint32_t* out    = (reverse ? g.rows_r : g.rows_f) + p.tb + j;
    const int32_t qn = qe - qb, tn = te - tb;
    const int32_t nw = ceil_div(qn, kWord), nch = ceil_div(nw, 64);
    if (nch == 1 || nch > kMwMaxChunks)
    {
        // one chunk: the single-wavefront routine with everything in registers (more than kMwMaxChunks: its LDS-state flavour
        // needs the launch's dynamic LDS, which this kernel does not ask for -- such halves are left to hb_span_rows_kernel)
        if (wave == 0 && nch == 1) hb_last_row(in, lane, qb, qe, tb, te, reverse, out);
        return;
    }
    if (threadIdx.x < kMwMaxChunks) sync.done[threadIdx.x] = 0;
    __syncthreads();
    const int32_t n_words_query = ceil_div(in.query_size, kWord);
    const PlainWords tab_f{in.pat_f}, tab_r{in.pat_r};
    const int32_t pattern_offset = reverse ? in.query_size - qe : qb;
    const char acgt[4] = {'A', 'C', 'T', 'G'};
    // this wavefront's chunks: slot i holds chunk wave + i W
    uint32_t pv[kMwSlots], mv[kMwSlots], e[kMwSlots][4];
    int32_t sc = qn; // running score of the last row (the wavefront that owns the last chunk)
#pragma unroll
    for (int i = 0; i < kMwSlots; i++)
    {
        const int32_t c = wave + i * W, w = c * 64 + lane;
        pv[i] = (c < nch && w < nw) ? ~0u : 0u; // lanes past the part: no carry generated or propagated
        mv[i] = 0u;
#pragma unroll
        for (int ci = 0; ci < 4; ci++)
            e[i][ci] = (c < nch && w < nw) ? (reverse ? get_pattern(tab_r, n_words_query, w, pattern_offset, acgt[ci])
                                                     : get_pattern(tab_f, n_words_query, w, pattern_offset, acgt[ci]))
                                           : 0u;
    }
    const int32_t last_lane  = (nw - 1) & 63;
    const uint32_t last_hbit = 1u << (qn - (nw - 1) * kWord - 1);
    const int32_t nb         = ceil_div(tn, 64);
    if (wave == ((nch - 1) % W) && lane == 0) out[0] = (uint32_t)qn;
    for (int32_t stage = 0; stage < nb + nch - 1; ++stage)
    {
#pragma unroll
        for (int i = 0; i < kMwSlots; i++)
        {
            const int32_t c = wave + i * W, k = stage - c;
            if (c >= nch || k < 0 || k >= nb) continue; // wave-uniform
            const int32_t t0 = 64 * k, cols = min(64, tn - t0);
            uint64_t in_c = 0ull, in_p = ~0ull, in_m = 0ull; // chunk 0: no carry, the implicit first row 0, 1, 2, ... (+1 per column)
            if (c > 0)
            {
                while (mw_poll(&sync.done[c - 1]) <= k) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const volatile uint64_t* h = sync.hand[c - 1][k % kMwRing];
                in_c = h[0]; in_p = h[1]; in_m = h[2];
                in_c = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(in_c >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)in_c);
                in_p = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(in_p >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)in_p);
                in_m = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(in_m >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)in_m);
            }
            const bool last_chunk = c == nch - 1;
            if (!last_chunk) // the slot this batch's boundary bits go to must have been read by the next chunk
                while (mw_poll(&sync.done[c + 1]) + kMwRing <= k) __builtin_amdgcn_s_sleep(1);
            const int32_t tt = t0 + 1 + lane; // column of this lane's target character
            const uint32_t tcv = tt <= tn ? (uint32_t)(unsigned char)(reverse ? in.target[te - tt] : in.target[tb + tt - 1]) : 0u;
            uint64_t out_c = 0ull, out_p = 0ull, out_m = 0ull;
            uint32_t acc = 0u; // last chunk: score after column t0 + 1 + lane
            uint32_t pvi = pv[i], mvi = mv[i];
            for (int32_t q = 0; q < cols; ++q)
            {
                const uint32_t tc = (uint32_t)__builtin_amdgcn_readlane((int32_t)tcv, q);
                const uint32_t ci = (tc >> 1) & 3u;
                const uint32_t m1 = 0u - (ci & 1u), m2 = 0u - (ci >> 1);
                const uint32_t lo = (e[i][1] & m1) | (e[i][0] & ~m1), hi = (e[i][3] & m1) | (e[i][2] & ~m1);
                const uint32_t eq = (hi & m2) | (lo & ~m2);
                const uint32_t carry = (uint32_t)((in_c >> q) & 1ull), hin_p = (uint32_t)((in_p >> q) & 1ull), hin_m = (uint32_t)((in_m >> q) & 1ull);
                const uint32_t xv  = eq | mvi;
                const uint32_t an  = eq & pvi;
                const uint32_t s0  = an + pvi;
                const uint64_t gen = __ballot(s0 < an), prp = __ballot(s0 == 0xffffffffu);
                const uint64_t top = 1ull << 63;
                const uint64_t cin = (((gen | prp) & ~top) + (gen & ~top) + carry) ^ (prp & ~top);
                const uint32_t carry_out = (uint32_t)(((gen >> 63) | ((prp >> 63) & (cin >> 63))) & 1ull);
                const uint32_t sum = s0 + (uint32_t)((cin >> lane) & 1ull);
                const uint32_t xh  = (sum ^ pvi) | eq;
                const uint32_t ph  = mvi | ~(xh | pvi);
                const uint32_t mh  = pvi & xh;
                uint32_t ph_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)(ph >> 31), 0x138, 0xf, 0xf, false); // wave_shr:1
                uint32_t mh_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)(mh >> 31), 0x138, 0xf, 0xf, false);
                if (lane == 0)
                {
                    ph_lo = hin_p;
                    mh_lo = hin_m;
                }
                const uint32_t hout_p = (uint32_t)__builtin_amdgcn_readlane((int32_t)(ph >> 31), 63);
                const uint32_t hout_m = (uint32_t)__builtin_amdgcn_readlane((int32_t)(mh >> 31), 63);
                out_c |= (uint64_t)carry_out << q;
                out_p |= (uint64_t)hout_p << q;
                out_m |= (uint64_t)hout_m << q;
                if (last_chunk)
                {
                    const uint32_t php = (uint32_t)__builtin_amdgcn_readlane((int32_t)ph, last_lane);
                    const uint32_t mhp = (uint32_t)__builtin_amdgcn_readlane((int32_t)mh, last_lane);
                    sc += ((php & last_hbit) ? 1 : 0) - ((mhp & last_hbit) ? 1 : 0);
                    if (lane == q) acc = (uint32_t)sc;
                }
                const uint32_t phs = (ph << 1) | ph_lo, mhs = (mh << 1) | mh_lo;
                pvi = mhs | ~(xv | phs);
                mvi = phs & xv;
            }
            pv[i] = pvi;
            mv[i] = mvi;
            if (last_chunk)
            {
                if (lane < cols) out[t0 + 1 + lane] = acc;
            }
            else
            {
                if (lane == 0)
                {
                    volatile uint64_t* h = sync.hand[c][k % kMwRing];
                    h[0] = out_c; h[1] = out_p; h[2] = out_m;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            }
            if (lane == 0) *reinterpret_cast<volatile int32_t*>(&sync.done[c]) = k + 1;
        }
    }
}

// grid (parts of the level, n): the two children of part j (or the part itself and an empty part, if it is not split)
__global__ __launch_bounds__(64) void hb_span_split_kernel(HirschbergArgs a, int32_t level)
{
    const int32_t lane = threadIdx.x & 63, idx = blockIdx.y, j = blockIdx.x;
    int32_t qw_max, t_max;
    hb_region_geometry(a, lane, qw_max, t_max);
    if (!hb_span_pair(a, idx)) return;
    const int64_t max_elems = (int64_t)ceil_div(max(a.max_query_length, 1), kWord) * (kHbSwitchToMyers + 1);
    if (!hb_span_fits(a, qw_max, t_max, max_elems)) return;
    const SpanGeom g        = hb_span_geom(a, idx, qw_max, t_max, max_elems);
    const LvPart p          = g.parts[level & 1][j];
    LvPart left = p, right = LvPart{p.qe, p.qe, p.te, p.te};
    if (hb_span_splits(p))
    {
        const int32_t qmid = p.qb + (p.qe - p.qb) / 2;
        const int32_t tmid = p.tb + hb_target_mid(g.rows_f + p.tb + j, g.rows_r + p.tb + j, p.te - p.tb, lane);
        left  = LvPart{p.qb, qmid, p.tb, tmid};
        right = LvPart{qmid, p.qe, tmid, p.te};
    }
    if (lane == 0)
    {
        g.parts[(level + 1) & 1][2 * j]     = left;
        g.parts[(level + 1) & 1][2 * j + 1] = right;
    }
}

// grid (parts of the last level, n): part j of the pair, depth-first, into its own piece of the staging buffer
__global__ __launch_bounds__(64) void hb_span_parts_kernel(HirschbergArgs a)
{
    extern __shared__ uint32_t hw_lds[];
    const int32_t lane = threadIdx.x & 63, idx = blockIdx.y, j = blockIdx.x;
    int32_t qw_max, t_max;
    hb_region_geometry(a, lane, qw_max, t_max);
    if (!hb_span_pair(a, idx)) return;
    const int64_t max_elems = (int64_t)ceil_div(max(a.max_query_length, 1), kWord) * (kHbSwitchToMyers + 1);
    if (!hb_span_fits(a, qw_max, t_max, max_elems)) return;
    const SpanGeom g        = hb_span_geom(a, idx, qw_max, t_max, max_elems);
    const LvPart p          = g.parts[g.levels & 1][j];
    HbWaveIn in{};
    in.query       = a.sequences + a.starts[2 * idx];
    in.target      = a.sequences + a.starts[2 * idx + 1];
    in.query_size  = (int32_t)(a.starts[2 * idx + 1] - a.starts[2 * idx]);
    in.target_size = (int32_t)(a.starts[2 * idx + 2] - a.starts[2 * idx + 1]);
    hb_span_pair_tables(a, idx, qw_max, t_max, max_elems, in.pat_f, in.pat_r);
    in.fwd = g.rows_f + p.tb + j;
    in.rev = g.rows_r + p.tb + j;
    const int64_t leaf_words = hb_leaf_words(p.te - p.tb, max_elems); // <= 2 x (target part + 1)
    in.leaf.pv     = g.leaf + 6 * ((int64_t)p.tb + j);
    in.leaf.mv     = in.leaf.pv + leaf_words;
    in.leaf.score  = reinterpret_cast<int32_t*>(in.leaf.mv + leaf_words);
    in.leaf.n_rows = 0;
    in.leaf.stride = 1;
    in.max_elems   = max_elems;
    in.stack       = hw_lds;
    in.st_pv       = hw_lds + 4 * kHbStackEntries;
    in.st_mv       = in.st_pv + (size_t)a.lds_state_words * 64;
    in.st_pt       = in.st_mv + (size_t)a.lds_state_words * 64;
    in.leaf_lds    = in.st_pt + (size_t)a.lds_state_words * 256;
    in.qb = p.qb; in.qe = p.qe; in.tb = p.tb; in.te = p.te;
    int32_t len = 0;
    if (p.qe > p.qb || p.te > p.tb) len = hb_wave_run(in, g.stage + p.qb + p.tb, lane);
    if (lane == 0) g.piece_len[j] = len;
}

// grid (n): the pieces of the pair's parts, last part first (the path is stored back to front), into the pair's result slot
__global__ __launch_bounds__(256) void hb_span_join_kernel(HirschbergArgs a)
{
    const int32_t lane = threadIdx.x & 63, idx = blockIdx.x;
    int32_t qw_max, t_max;
    hb_region_geometry(a, lane, qw_max, t_max);
    if (!hb_span_pair(a, idx)) return;
    const int64_t max_elems = (int64_t)ceil_div(max(a.max_query_length, 1), kWord) * (kHbSwitchToMyers + 1);
    if (!hb_span_fits(a, qw_max, t_max, max_elems)) return; // the depth-first kernel aligned the pair
    const SpanGeom g        = hb_span_geom(a, idx, qw_max, t_max, max_elems);
    const LvPart* parts     = g.parts[g.levels & 1];
    int8_t* path            = a.results + a.starts[2 * idx];
    int64_t at = 0;
    bool ok    = true;
    for (int32_t j = g.P - 1; j >= 0; --j)
    {
        const int32_t len = g.piece_len[j];
        if (len < 0) { ok = false; break; }
        const int8_t* src = g.stage + parts[j].qb + parts[j].tb;
        for (int32_t k = threadIdx.x; k < len; k += blockDim.x) path[at + k] = src[k];
        at += len;
    }
    if (threadIdx.x == 0) a.result_lengths[idx] = ok ? (int32_t)at : 0;
}

// ------------------------------------------------------------------------------------------------
// Default aligner, one wavefront per pair, LEVEL BY LEVEL (round 3). The depth-first kernel above spends a 1 kbp pair on
// 4 levels x 2 x 1 000 dependent column steps with at most 16 of its 64 lanes at work (a part of 500 bases is 16 words),
// because it handles one (part, direction) at a time. The split of a part depends on nothing but the part, so the tree can
// just as well be grown a whole level at a time: here every lane owns one word of one SEGMENT -- the forward half or the
// reversed second half of one part of the current level -- all segments of a level stand side by side in the wavefront
// (about query words + parts lanes; greedy batches of parts when they exceed 64) and advance one target column per step.
// The multi-word addition is the carry-lookahead of the other kernels with the chain cut at every segment's last lane
// (generate and propagate cleared there); a segment's first lane takes the +1 of the implicit first row instead of its
// neighbour's bit; each lane reads ITS segment's target character (forward or back to front) from an LDS copy of the
// target; each segment's last lane tracks the score of its last row and writes it (uint16) to the part's row in LDS.
// Level k then costs about T / 2^k column steps instead of T -- 2 T for the whole tree instead of 2 T per level.
// What decides the output is unchanged: query midpoint len / 2, the target midpoint as the reference's 32 lanes pick it
// among equal sums (hirschberg_myers_compute_target_mid_warp, :461-481), the terminal cases (empty side, single query
// character, full Myers matrix + backtrace below 63 query characters when the matrix fits). The reference emits the path
// back to front by popping the right child first; here the terminals of all levels are ordered by (query begin, target
// begin) descending and appended in that order by the same three handlers as above.
// Eligible pairs: query <= 2 048 (<= 64 words per level-0 segment pair), target <= the LDS row capacity chosen from
// max_query_length; every other pair, and a pair whose lists overflow (flag -2 in result_lengths), is left to
// hirschberg_wave_kernel, launched right behind on the same stream.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void hirschberg_levels_kernel(HirschbergArgs a)
{
    extern __shared__ uint32_t lv_lds[];
    const int32_t lane = threadIdx.x & 63;
    const int32_t idx  = blockIdx.x;
    if (idx >= a.n) return;
    const int32_t query_size  = (int32_t)(a.starts[2 * idx + 1] - a.starts[2 * idx]);
    const int32_t target_size = (int32_t)(a.starts[2 * idx + 2] - a.starts[2 * idx + 1]);
    if (!lv_eligible(query_size, target_size, a.max_query_length)) return;
    const int32_t region_index = idx >> 6, s = idx & 63;
    const int64_t max_elems = (int64_t)ceil_div(max(a.max_query_length, 1), kWord) * (kHbSwitchToMyers + 1);
    // the pair's HBM workspace (geometry of the depth-first kernel): only a leaf too large for LDS uses it
    int32_t qw_max = 0, t_max = 0;
    {
        const int32_t sj = region_index * 64 + lane;
        if (sj < a.n)
        {
            qw_max = ceil_div((int32_t)(a.starts[2 * sj + 1] - a.starts[2 * sj]), kWord);
            t_max  = (int32_t)(a.starts[2 * sj + 2] - a.starts[2 * sj + 1]);
        }
        for (int off = 32; off > 0; off >>= 1)
        {
            qw_max = max(qw_max, __shfl_xor(qw_max, off));
            t_max  = max(t_max, __shfl_xor(t_max, off));
        }
    }
    const char* query  = a.sequences + a.starts[2 * idx];
    const char* target = a.sequences + a.starts[2 * idx + 1];
    int8_t* path       = a.results + a.starts[2 * idx];
    const int64_t region   = a.wave_offsets[region_index];
    const int64_t per_pair = hb_lane_words(qw_max, t_max, max_elems);
    if (region + 64 * per_pair > a.ws_capacity_words)
    {
        if (lane == 0) a.result_lengths[idx] = 0;
        return;
    }
    uint32_t* ws_base = a.ws + region + (size_t)s * (size_t)per_pair;
    const int64_t leaf_words_hbm = hb_leaf_words(t_max, max_elems);
    uint32_t* leaf_hbm = ws_base + 4 * kHbStackEntries + 2 * ((size_t)t_max + 1) + (size_t)10 * qw_max;

    const LvLayout L = lv_layout(a.max_query_length);
    uint8_t* lds8    = reinterpret_cast<uint8_t*>(lv_lds);
    uint32_t* pat_f  = reinterpret_cast<uint32_t*>(lds8 + L.pat_f);
    uint32_t* pat_r  = reinterpret_cast<uint32_t*>(lds8 + L.pat_r);
    uint8_t* tgt     = lds8 + L.tgt;
    uint16_t* rows_f = reinterpret_cast<uint16_t*>(lds8 + L.rows_f);
    uint16_t* rows_r = reinterpret_cast<uint16_t*>(lds8 + L.rows_r);
    LvPart* parts    = reinterpret_cast<LvPart*>(lds8 + L.parts);
    LvPart* term     = reinterpret_cast<LvPart*>(lds8 + L.term);
    uint32_t* keys   = reinterpret_cast<uint32_t*>(lds8 + L.keys);
    uint16_t* order  = reinterpret_cast<uint16_t*>(lds8 + L.order);
    uint32_t* segtab = reinterpret_cast<uint32_t*>(lds8 + L.segtab);
    uint32_t* leaf_lds = reinterpret_cast<uint32_t*>(lds8 + L.rows_f);

    // pattern tables of the whole query (forward and back to front) and the target, into LDS
    const int32_t n_words_query = ceil_div(query_size, kWord);
    for (int32_t w = lane; w < n_words_query; w += 64)
    {
        uint32_t f[4], r[4];
        pattern_words(query, query_size, w, f, r);
#pragma unroll
        for (int ci = 0; ci < 4; ci++)
        {
            pat_f[w * 4 + ci] = f[ci];
            pat_r[w * 4 + ci] = r[ci];
        }
    }
    for (int32_t i = lane; i < target_size; i += 64) tgt[i] = (uint8_t)target[i];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const PlainWords tab_f{pat_f}, tab_r{pat_r};

    // what becomes of a part (same case order as the depth-first kernels): 0 split, 1 run of deletions / insertions,
    // 2 full Myers matrix + backtrace, 3 single query character
    auto kind_of = [&](int32_t qn, int32_t tn) -> int32_t {
        if (tn == 0 || qn == 0) return 1;
        const int32_t nw = ceil_div(qn, kWord);
        if (qn >= 2 && qn < kHbSwitchToMyers && (int64_t)(tn + 1) * nw <= max_elems) return 2;
        if (qn == 1) return 3;
        return 0;
    };
    int32_t n_cur = 0, n_next = 0, n_term = 0, cur = 0; // wave-uniform; parts[cur * kLvMaxParts + k] is the current level
    bool redo = false;
    auto add_part = [&](int32_t qb, int32_t qe, int32_t tb, int32_t te, bool to_current) {
        const int32_t qn = qe - qb, tn = te - tb;
        if (qn == 0 && tn == 0) return;
        const LvPart pp{qb, qe, tb, te};
        if (kind_of(qn, tn) == 0)
        {
            int32_t& cnt = to_current ? n_cur : n_next;
            if (cnt >= kLvMaxParts) { redo = true; return; }
            if (lane == 0) parts[(to_current ? cur : (cur ^ 1)) * kLvMaxParts + cnt] = pp;
            ++cnt;
        }
        else
        {
            if (n_term >= kLvMaxTerm) { redo = true; return; }
            if (lane == 0) term[n_term] = pp;
            ++n_term;
        }
    };
    add_part(0, query_size, 0, target_size, true);

    const char acgt[4] = {'A', 'C', 'T', 'G'};
    while (n_cur > 0 && !redo)
    {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        // lane p looks at part p of the level: words of its forward and of its reversed segment
        int32_t wf = 0, wr = 0;
        if (lane < n_cur)
        {
            const LvPart pp  = parts[cur * kLvMaxParts + lane];
            const int32_t qn = pp.qe - pp.qb, qh = qn / 2;
            wf = ceil_div(qh, kWord);
            wr = ceil_div(qn - qh, kWord);
        }
        const int32_t wsum = wf + wr;
        int32_t first = 0;
        while (first < n_cur)
        {
            // lanes of the parts from `first` on: exclusive prefix sum of their words; the batch is what fits 64 lanes
            int32_t incl = lane >= first ? wsum : 0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1)
            {
                const int32_t o = __shfl_up(incl, off);
                if (lane >= off) incl += o;
            }
            const int32_t excl = incl - (lane >= first ? wsum : 0);
            const bool in_b    = lane >= first && lane < n_cur && incl <= 64;
            const int32_t nb   = __popcll(__ballot(in_b));
            segtab[lane] = 0xffffffffu;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            if (in_b)
                for (int32_t i = 0; i < wsum; ++i)
                    segtab[excl + i] = (uint32_t)lane | (i >= wf ? 0x100u : 0u) | ((uint32_t)(i >= wf ? i - wf : i) << 16);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            // ---- this lane's word ----
            const uint32_t ent = segtab[lane];
            const bool idle    = ent == 0xffffffffu;
            const int32_t p    = idle ? first : (int32_t)(ent & 0xffu);
            const bool rev     = !idle && (ent & 0x100u) != 0;
            const int32_t w    = idle ? 0 : (int32_t)(ent >> 16);
            const LvPart pp    = parts[cur * kLvMaxParts + p];
            const int32_t qmid = pp.qb + (pp.qe - pp.qb) / 2;
            const int32_t slen = rev ? pp.qe - qmid : qmid - pp.qb;
            const int32_t nws  = ceil_div(slen, kWord);
            const int32_t poff = rev ? query_size - pp.qe : pp.qb;
            uint32_t e[4];
#pragma unroll
            for (int ci = 0; ci < 4; ci++)
                e[ci] = idle ? 0u : (rev ? get_pattern(tab_r, n_words_query, w, poff, acgt[ci]) : get_pattern(tab_f, n_words_query, w, poff, acgt[ci]));
            const bool is_first = idle || w == 0;
            const bool is_last  = !idle && w == nws - 1;
            const uint32_t hbit = 1u << ((slen - (nws - 1) * kWord - 1) & 31);
            const int32_t tn    = idle ? 0 : pp.te - pp.tb;
            const int32_t tstep = rev ? -1 : 1;
            const int32_t taddr = idle ? 0 : (rev ? pp.te : pp.tb - 1); // column t reads tgt[taddr + tstep * t]
            uint16_t* row       = (rev ? rows_r : rows_f) + pp.tb + p;
            const uint64_t cutmask = __ballot(idle || is_last);
            int32_t tmax = tn;
            for (int off = 32; off > 0; off >>= 1) tmax = max(tmax, __shfl_xor(tmax, off));
            tmax = __builtin_amdgcn_readfirstlane(tmax);
            uint32_t pv = idle ? 0u : ~0u, mv = 0u;
            int32_t sc  = slen;
            if (is_last) row[0] = (uint16_t)sc;
            for (int32_t t = 1; t <= tmax; ++t)
            {
                const int32_t tt  = min(t, tn);
                const uint32_t tc = tgt[taddr + tstep * tt];
                const uint32_t ci = (tc >> 1) & 3u;
                const uint32_t lo = (ci & 1u) ? e[1] : e[0], hi = (ci & 1u) ? e[3] : e[2];
                const uint32_t eq = (ci & 2u) ? hi : lo;
                const uint32_t xv = eq | mv;
                const uint32_t an = eq & pv;
                const uint32_t s0 = an + pv;
                const uint64_t gen = __ballot(s0 < an) & ~cutmask, prp = __ballot(s0 == 0xffffffffu) & ~cutmask;
                const uint64_t cin = ((gen | prp) + gen) ^ prp;
                const uint32_t sum = s0 + (uint32_t)((cin >> lane) & 1ull);
                const uint32_t xh  = (sum ^ pv) | eq;
                const uint32_t ph  = mv | ~(xh | pv);
                const uint32_t mh  = pv & xh;
                uint32_t ph_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)(ph >> 31), 0x138, 0xf, 0xf, false); // wave_shr:1
                uint32_t mh_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)(mh >> 31), 0x138, 0xf, 0xf, false);
                if (is_first) // the implicit first row is 0, 1, 2, ...
                {
                    ph_lo = 1u;
                    mh_lo = 0u;
                }
                if (is_last)
                {
                    sc += ((ph & hbit) ? 1 : 0) - ((mh & hbit) ? 1 : 0);
                    if (t <= tn) row[t] = (uint16_t)sc;
                }
                const uint32_t phs = (ph << 1) | ph_lo, mhs = (mh << 1) | mh_lo;
                pv = mhs | ~(xv | phs);
                mv = phs & xv;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            // ---- target midpoints of the batch's parts and their children ----
            for (int32_t k = first; k < first + nb; ++k)
            {
                const LvPart q     = parts[cur * kLvMaxParts + k];
                const int32_t qb = __builtin_amdgcn_readfirstlane(q.qb), qe = __builtin_amdgcn_readfirstlane(q.qe);
                const int32_t tb = __builtin_amdgcn_readfirstlane(q.tb), te = __builtin_amdgcn_readfirstlane(q.te);
                const int32_t tnk = te - tb;
                const uint16_t* rf = rows_f + tb + k;
                const uint16_t* rr = rows_r + tb + k;
                // hirschberg_myers_compute_target_mid_warp (:461-481) on 32 real lanes: lane L sees t = L, L + 32, ...; the
                // shuffle-down tree keeps the lower lane on equal sums
                int32_t cm = INT32_MAX, mp = 0;
                if (lane < 32)
                    for (int32_t t = lane; t <= tnk; t += 32)
                    {
                        const int32_t sum = (int32_t)rf[t] + (int32_t)rr[tnk - t];
                        if (sum < cm) { cm = sum; mp = t; }
                    }
                for (int32_t step = 16; step > 0; step >>= 1)
                {
                    const int32_t om = __shfl_down(cm, step, 32), ot = __shfl_down(mp, step, 32);
                    if ((lane & 31) + step < 32 && om < cm) { cm = om; mp = ot; }
                }
                const int32_t tmid = tb + __builtin_amdgcn_readfirstlane(mp);
                const int32_t qm   = qb + (qe - qb) / 2;
                add_part(qb, qm, tb, tmid, false);
                add_part(qm, qe, tmid, te, false);
            }
            first += nb;
        }
        cur ^= 1;
        n_cur  = n_next;
        n_next = 0;
    }
    if (redo)
    {
        if (lane == 0) a.result_lengths[idx] = kLvFlagRedo;
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    // ---- order of the terminals: the reference pops the right child first, i.e. descending (query begin, target begin) ----
    for (int32_t k = lane; k < n_term; k += 64) keys[k] = ((uint32_t)term[k].qb << 16) | (uint32_t)term[k].tb;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (int32_t k = lane; k < n_term; k += 64)
    {
        const uint32_t mine = keys[k];
        int32_t rank        = 0;
        for (int32_t j = 0; j < n_term; ++j) rank += keys[j] > mine ? 1 : 0;
        order[rank] = (uint16_t)k;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");

    Band leaf; // matrices of a leaf that does not fit LDS (column-major), lane 0 only
    leaf.pv     = leaf_hbm;
    leaf.mv     = leaf.pv + (size_t)leaf_words_hbm;
    leaf.score  = reinterpret_cast<int32_t*>(leaf.mv + (size_t)leaf_words_hbm);
    leaf.n_rows = 0;
    leaf.stride = 1;
    int32_t len = 0;
    int8_t* stage = reinterpret_cast<int8_t*>(lds8 + L.stage);
    for (int32_t r = 0; r < n_term; ++r)
    {
        // ---- a round of terminals side by side: lane l takes terminal r + l, writes its piece of the path to an LDS stage
        // (its length is only known afterwards), and the pieces are appended in order. A round takes the leading terminals
        // (at most kLvRound) whose matrices fit the leaf area together and whose pieces fit the stage; a terminal that fits
        // neither on its own (a long run, a leaf whose matrix needs the HBM workspace) is handled by the whole wavefront below.
        {
            int32_t t_kind = 0, t_qb = 0, t_qe = 0, t_tb = 0, t_te = 0, elems = 0, cap = 0;
            const bool have = lane < kLvRound && r + lane < n_term;
            if (have)
            {
                const LvPart q = term[order[r + lane]];
                t_qb = q.qb; t_qe = q.qe; t_tb = q.tb; t_te = q.te;
                const int32_t qn = t_qe - t_qb, tn = t_te - t_tb;
                t_kind = kind_of(qn, tn);
                elems  = t_kind == 2 ? (tn + 1) * ceil_div(qn, kWord) : 0;
                cap    = qn + tn;
            }
            int32_t ie = elems, ic = cap; // inclusive prefix sums over the lanes
#pragma unroll
            for (int off = 1; off < kLvRound; off <<= 1)
            {
                const int32_t oe = __shfl_up(ie, off), oc = __shfl_up(ic, off);
                if (lane >= off) { ie += oe; ic += oc; }
            }
            const bool fits      = have && ie <= L.leaf_elems && ic <= kLvStageBytes;
            const uint64_t fm    = __ballot(fits) | (~0ull << kLvRound);
            const int32_t m      = ~fm == 0 ? kLvRound : (int32_t)__builtin_ctzll(~fm); // leading lanes that fit
            if (m >= 2)
            {
                const bool mine  = lane < m;
                const int32_t eo = ie - elems, so = ic - cap; // this lane's matrix and stage offsets
                int32_t plen = 0;
                if (mine)
                {
                    const int32_t qn = t_qe - t_qb, tn = t_te - t_tb;
                    int8_t* out = stage + so;
                    if (t_kind == 1)
                    {
                        const int32_t count = tn == 0 ? qn : tn;
                        const int8_t state  = tn == 0 ? kDeletion : kInsertion;
                        for (int32_t c = 0; c < count; ++c) out[c] = state;
                        plen = count;
                    }
                    else if (t_kind == 3)
                    {
                        const uint8_t qc = (uint8_t)query[t_qb];
                        int32_t pz = 0, t = t_te - 1;
                        bool matched = false;
                        while (t >= t_tb)
                        {
                            if (tgt[t] == qc) { out[pz++] = kMatch; --t; matched = true; break; }
                            out[pz++] = kInsertion;
                            --t;
                        }
                        if (!matched) out[pz - 1] = kMismatch;
                        while (t >= t_tb) { out[pz++] = kInsertion; --t; }
                        plen = tn;
                    }
                    else
                    {
                        const int32_t nw = ceil_div(qn, kWord);
                        Band lf;
                        lf.pv     = leaf_lds + eo;
                        lf.mv     = leaf_lds + L.leaf_elems + eo;
                        lf.score  = reinterpret_cast<int32_t*>(leaf_lds + 2 * L.leaf_elems + eo);
                        lf.n_rows = nw;
                        lf.stride = 1;
                        for (int32_t w = 0; w < nw; ++w)
                        {
                            lf.pv[lf.at(w, 0)]    = ~0u;
                            lf.mv[lf.at(w, 0)]    = 0u;
                            lf.score[lf.at(w, 0)] = min((w + 1) * kWord, qn);
                        }
                        const uint32_t last_hbit = 1u << (qn - (nw - 1) * kWord - 1);
                        uint32_t lp[2][4];
                        for (int32_t w = 0; w < 2; ++w)
                            for (int ci = 0; ci < 4; ci++) lp[w][ci] = w < nw ? get_pattern(tab_f, n_words_query, w, t_qb, acgt[ci]) : 0u;
                        uint32_t pv0 = ~0u, mv0 = 0u, pv1 = ~0u, mv1 = 0u;
                        int32_t s0 = min(kWord, qn), s1 = qn;
                        for (int32_t t = 1; t <= tn; ++t)
                        {
                            const uint32_t ci = (tgt[t_tb + t - 1] >> 1) & 3u;
                            int32_t h = advance_word(nw == 1 ? last_hbit : (1u << (kWord - 1)), select4(ci, lp[0][0], lp[0][1], lp[0][2], lp[0][3]), pv0, mv0, 1, nullptr);
                            s0 += h;
                            lf.score[lf.at(0, t)] = s0;
                            lf.pv[lf.at(0, t)]    = pv0;
                            lf.mv[lf.at(0, t)]    = mv0;
                            if (nw == 2)
                            {
                                h = advance_word(last_hbit, select4(ci, lp[1][0], lp[1][1], lp[1][2], lp[1][3]), pv1, mv1, h, nullptr);
                                s1 += h;
                                lf.score[lf.at(1, t)] = s1;
                                lf.pv[lf.at(1, t)]    = pv1;
                                lf.mv[lf.at(1, t)]    = mv1;
                            }
                        }
                        const uint32_t last_mask = qn % kWord != 0 ? ((1u << (qn % kWord)) - 1) : ~0u;
                        int32_t i = qn, j = tn, l = 0;
                        int32_t myscore = lf.score[lf.at((i - 1) / kWord, j)];
                        while (i > 0 && j > 0)
                        {
                            // left as the reference reads it; the cell above it (diag) and the cell above the current one from
                            // the vertical-delta bits of row i (see backtrace_banded's fetch3); row 0 of the matrix is 0, 1, 2, ...
                            const int32_t wi = (i - 1) / kWord, bi = (i - 1) % kWord;
                            const uint32_t pl = lf.pv[lf.at(wi, j - 1)], nl = lf.mv[lf.at(wi, j - 1)];
                            const uint32_t pa = lf.pv[lf.at(wi, j)], na = lf.mv[lf.at(wi, j)];
                            uint32_t mask     = bi == 31 ? 0u : ((~1u) << bi);
                            if (wi == nw - 1) mask &= last_mask;
                            const int32_t left  = lf.score[lf.at(wi, j - 1)] - __popc(mask & pl) + __popc(mask & nl);
                            const int32_t diag  = i == 1 ? j - 1 : left - ((int32_t)((pl >> bi) & 1u) - (int32_t)((nl >> bi) & 1u));
                            const int32_t above = i == 1 ? j : myscore - ((int32_t)((pa >> bi) & 1u) - (int32_t)((na >> bi) & 1u));
                            int8_t st;
                            if (left + 1 == myscore) { st = kInsertion; myscore = left; --j; }
                            else if (above + 1 == myscore) { st = kDeletion; myscore = above; --i; }
                            else { st = diag == myscore ? kMatch : kMismatch; myscore = diag; --i; --j; }
                            out[l++] = st;
                        }
                        for (int32_t c = 0; c < i; ++c) out[l + c] = kDeletion;
                        l += i;
                        for (int32_t c = 0; c < j; ++c) out[l + c] = kInsertion;
                        l += j;
                        plen = l;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                // append the pieces in order
                for (int32_t q = 0; q < m; ++q)
                {
                    const int32_t pl = __builtin_amdgcn_readlane(plen, q), po = __builtin_amdgcn_readlane(so, q);
                    for (int32_t c = lane; c < pl; c += 64) path[len + c] = stage[po + c];
                    len += pl;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // the next round reuses the stage and the leaf area
                r += m - 1;
                continue;
            }
        }
        const int32_t k  = __builtin_amdgcn_readfirstlane((int32_t)order[r]);
        const LvPart q   = term[k];
        const int32_t qb = __builtin_amdgcn_readfirstlane(q.qb), qe = __builtin_amdgcn_readfirstlane(q.qe);
        const int32_t tb = __builtin_amdgcn_readfirstlane(q.tb), te = __builtin_amdgcn_readfirstlane(q.te);
        const int32_t qn = qe - qb, tn = te - tb;
        const int32_t nw = ceil_div(max(qn, 1), kWord);
        const int32_t kind = kind_of(qn, tn);
        if (kind == 1)
        {
            // one side empty: a run of deletions / insertions, written by all lanes
            const int32_t count = tn == 0 ? qn : tn;
            const int8_t state  = tn == 0 ? kDeletion : kInsertion;
            for (int32_t c = lane; c < count; c += 64) path[len + c] = state;
            len += count;
        }
        else if (kind == 2)
        {
            // leaf: full Myers matrix (hirschberg_myers_compute_path, :383-410) + append_myers_backtrace (:124-181): the
            // forward pass runs on lane 0 (a chain of word steps), the walk evaluates its three neighbour cells on three lanes
            Band lf   = leaf;
            lf.n_rows = nw;
            const bool in_lds = (tn + 1) * nw <= L.leaf_elems;
            if (in_lds)
            {
                lf.pv    = leaf_lds;
                lf.mv    = leaf_lds + L.leaf_elems;
                lf.score = reinterpret_cast<int32_t*>(leaf_lds + 2 * L.leaf_elems);
            }
            if (lane == 0)
            {
                for (int32_t w = 0; w < nw; ++w)
                {
                    lf.pv[lf.at(w, 0)]    = ~0u;
                    lf.mv[lf.at(w, 0)]    = 0u;
                    lf.score[lf.at(w, 0)] = min((w + 1) * kWord, qn);
                }
                const uint32_t last_hbit = 1u << (qn - (nw - 1) * kWord - 1);
                uint32_t lp[2][4];
                for (int32_t w = 0; w < 2; ++w)
                    for (int ci = 0; ci < 4; ci++) lp[w][ci] = w < nw ? get_pattern(tab_f, n_words_query, w, qb, acgt[ci]) : 0u;
                uint32_t pv0 = ~0u, mv0 = 0u, pv1 = ~0u, mv1 = 0u;
                int32_t s0 = min(kWord, qn), s1 = qn;
                for (int32_t t = 1; t <= tn; ++t)
                {
                    const int32_t ci = (tgt[tb + t - 1] >> 1) & 3;
                    int32_t h = advance_word(nw == 1 ? last_hbit : (1u << (kWord - 1)), lp[0][ci], pv0, mv0, 1, nullptr);
                    s0 += h;
                    lf.score[lf.at(0, t)] = s0;
                    lf.pv[lf.at(0, t)]    = pv0;
                    lf.mv[lf.at(0, t)]    = mv0;
                    if (nw == 2)
                    {
                        h = advance_word(last_hbit, lp[1][ci], pv1, mv1, h, nullptr);
                        s1 += h;
                        lf.score[lf.at(1, t)] = s1;
                        lf.pv[lf.at(1, t)]    = pv1;
                        lf.mv[lf.at(1, t)]    = mv1;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            if (!in_lds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t last_mask = qn % kWord != 0 ? ((1u << (qn % kWord)) - 1) : ~0u;
            int32_t i = qn, j = tn, l = len;
            int32_t myscore = __builtin_amdgcn_readfirstlane(lf.score[lf.at((i - 1) / kWord, j)]);
            while (i > 0 && j > 0)
            {
                // lane 0: above (i - 1, j), lane 1: diagonal (i - 1, j - 1), lane 2: left (i, j - 1)
                const int32_t ci_ = lane == 2 ? i : i - 1, cj_ = lane == 0 ? j : j - 1;
                int32_t v = 0;
                if (lane < 3) v = ci_ == 0 ? cj_ : cell_score(lf, max(ci_, 1), cj_, last_mask); // row 0 of the matrix is 0, 1, 2, ...
                const int32_t above = __builtin_amdgcn_readlane(v, 0), diag = __builtin_amdgcn_readlane(v, 1), left = __builtin_amdgcn_readlane(v, 2);
                int8_t st;
                if (left + 1 == myscore) { st = kInsertion; myscore = left; --j; }
                else if (above + 1 == myscore) { st = kDeletion; myscore = above; --i; }
                else { st = diag == myscore ? kMatch : kMismatch; myscore = diag; --i; --j; }
                if (lane == 0) path[l] = st;
                l++;
            }
            for (int32_t c = lane; c < i; c += 64) path[l + c] = kDeletion;
            l += i;
            for (int32_t c = lane; c < j; c += 64) path[l + c] = kInsertion;
            l += j;
            len = l;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // the next leaf reuses the matrices
        }
        else
        {
            // hirschberg_myers_single_char_warp (:483-515): right-to-left scan for the first equal character
            int32_t new_len = len;
            if (lane == 0)
            {
                const uint8_t qc = (uint8_t)query[qb];
                int32_t pz       = len;
                int32_t t        = te - 1;
                while (t >= tb)
                {
                    if (tgt[t] == qc) { path[pz++] = kMatch; --t; break; }
                    path[pz++] = kInsertion;
                    --t;
                }
                if (path[pz - 1] != kMatch) path[pz - 1] = kMismatch;
                while (t >= tb) { path[pz++] = kInsertion; --t; }
                new_len = len + tn;
            }
            len = __builtin_amdgcn_readfirstlane(new_len);
        }
    }
    if (lane == 0) a.result_lengths[idx] = len;
}

// ------------------------------------------------------------------------------------------------
// Unit hooks (the reference's test kernels: Test_HirschbergMyers.cu:37-53, Test_MyersAlgorithm.cu:42-97): they run the
// production device functions above on one pair with one active lane.
// ------------------------------------------------------------------------------------------------
struct PlainTable
{
    const uint32_t* p;
    __device__ __forceinline__ uint32_t operator[](int32_t e) const { return p[e]; }
};

// out[w * 8 + c]: c = 0..3 forward A, C, T, G; 4..7 back to front (the layout of the reference's n_words x 8 matrix)
__global__ void myers_patterns_hook_kernel(const char* query, int32_t query_size, uint32_t* out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int32_t n_words = ceil_div(query_size, kWord);
    for (int32_t w = 0; w < n_words; ++w)
    {
        uint32_t f[4], r[4];
        pattern_words(query, query_size, w, f, r);
        for (int c = 0; c < 4; ++c)
        {
            out[w * 8 + c]     = f[c];
            out[w * 8 + 4 + c] = r[c];
        }
    }
}

// out[i] = get_query_pattern(patterns, idx, i, x, reverse) for the shifts i = 0..31 (hirschberg_myers_gpu.cu:244-276);
// table: scratch of 4 * n_words words
__global__ void myers_get_pattern_hook_kernel(const char* query, int32_t query_size, int32_t idx, char x, int32_t reverse,
                                              uint32_t* table, uint32_t* out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int32_t n_words = ceil_div(query_size, kWord);
    for (int32_t w = 0; w < n_words; ++w)
    {
        uint32_t f[4], r[4];
        pattern_words(query, query_size, w, f, r);
        for (int c = 0; c < 4; ++c) table[w * 4 + c] = reverse ? r[c] : f[c];
    }
    const PlainTable t{table};
    for (int32_t i = 0; i < 32; ++i) out[i] = get_pattern(t, n_words, idx, i, x);
}

// one band attempt with a given band (myers_compute_scores_edit_dist_banded_test_kernel, Test_MyersAlgorithm.cu:42-97):
// lane 0 of one wave; ws = 64 * (3 * n_words_band * (target + 1) + 4 * n_words) words, interleaved like the production kernel
__global__ __launch_bounds__(64) void myers_banded_matrices_hook_kernel(const char* query, const char* target, int32_t query_size,
                                                                        int32_t target_size, int32_t band_width, int32_t p, uint32_t* ws,
                                                                        int32_t* diag_out)
{
    if (threadIdx.x != 0) return;
    const int32_t n_words      = ceil_div(query_size, kWord);
    const int32_t n_words_band = ceil_div(band_width, kWord);
    const int64_t me           = (int64_t)n_words_band * ((int64_t)target_size + 1);
    Band b;
    b.pv     = ws;
    b.mv     = ws + 64 * me;
    b.score  = reinterpret_cast<int32_t*>(ws + 128 * me);
    b.n_rows = n_words_band;
    const LaneArray patterns{ws + 192 * me};
    for (int32_t w = 0; w < n_words; ++w)
    {
        uint32_t f[4], r[4];
        pattern_words(query, query_size, w, f, r);
        for (int c = 0; c < 4; ++c) patterns[w * 4 + c] = f[c];
    }
    if (band_width - (n_words_band - 1) * kWord < 2) // invalid band: everything zero (the reference's test kernel does the same)
    {
        for (int32_t t = 0; t <= target_size; ++t)
            for (int32_t w = 0; w < n_words_band; ++w) { b.pv[b.at(w, t)] = 0; b.mv[b.at(w, t)] = 0; b.score[b.at(w, t)] = 0; }
        return;
    }
    for (int32_t w = 0; w < n_words_band; ++w)
    {
        b.pv[b.at(w, 0)]    = ~0u;
        b.mv[b.at(w, 0)]    = 0u;
        b.score[b.at(w, 0)] = min((w + 1) * kWord, band_width);
    }
    ColumnState cs{};
    int32_t diagonal_begin = -1, diagonal_end = -1;
    banded_stripes<false>(b, cs, patterns, n_words, target, query_size, target_size, p, n_words_band, band_width, diagonal_begin, diagonal_end);
    diag_out[0] = diagonal_begin;
    diag_out[1] = diagonal_end;
}

static int fail(hipError_t e, const char* what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return (int)e;
}

} // namespace myers
} // namespace gwhip

using namespace gwhip;
using namespace gwhip::myers;

extern "C" {

int64_t gwhip_myers_banded_workspace_words(int32_t first_slot, int32_t n_slots, const int64_t* sequence_starts_host,
                                           const int32_t* max_bandwidths_host, const int32_t* scheduling_index_host)
{
    int64_t words = 0;
    for (int32_t w0 = first_slot; w0 < first_slot + n_slots; w0 += 64) // one interleaved region per wave of 64 slots
    {
        int64_t me_max = 0;
        int32_t pw_max = 0;
        for (int32_t s = w0; s < std::min(first_slot + n_slots, w0 + 64); s++)
        {
            const int32_t i = scheduling_index_host ? scheduling_index_host[s] : s;
            int64_t me;
            int32_t pw;
            pair_ws_dims((int32_t)(sequence_starts_host[2 * i + 1] - sequence_starts_host[2 * i]),
                         (int32_t)(sequence_starts_host[2 * i + 2] - sequence_starts_host[2 * i + 1]), max_bandwidths_host[i], me, pw);
            me_max = std::max(me_max, me);
            pw_max = std::max(pw_max, pw);
        }
        words += 64 * (3 * me_max + pw_max);
    }
    return words;
}

size_t gwhip_myers_banded_workspace_bytes_of_words(int32_t n_alignments, int64_t total_sequence_length, int64_t words)
{
    if (n_alignments <= 0) return 256;
    // (the result slots are indexed relative to the first pair's offset: a chunk of a larger batch passes its own slice)
    const WsPlan p = plan_fixed(n_alignments, total_sequence_length);
    return p.off_ws + (size_t)words * 4 + 256;
}

size_t gwhip_myers_banded_workspace_bytes_ordered(int32_t n_alignments, const int64_t* sequence_starts_host,
                                                  const int32_t* max_bandwidths_host, const int32_t* scheduling_index_host)
{
    if (n_alignments <= 0) return 256;
    return gwhip_myers_banded_workspace_bytes_of_words(
        n_alignments, sequence_starts_host[2 * (size_t)n_alignments] - sequence_starts_host[0],
        gwhip_myers_banded_workspace_words(0, n_alignments, sequence_starts_host, max_bandwidths_host, scheduling_index_host));
}

size_t gwhip_myers_banded_workspace_bytes(int32_t n_alignments, const int64_t* sequence_starts_host,
                                          const int32_t* max_bandwidths_host)
{
    return gwhip_myers_banded_workspace_bytes_ordered(n_alignments, sequence_starts_host, max_bandwidths_host, nullptr);
}

// two bases per byte -> one character per base (include/gwhip.h: gwhip_unpack_bases); a thread expands 8 packed bytes
__global__ __launch_bounds__(256) void unpack_bases_kernel(const uint8_t* packed, char* out, int64_t first, int64_t last)
{
    // bytes 'A' 'C' 'T' 'G' 'N' 'N' 'N' 'N' of the codes 0..7 (5..7 never occur)
    constexpr uint64_t kLut = 0x4e4e4e4e47544341ull;
    const int64_t i0 = (first & ~int64_t(15)) + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; // 16 bases, 16-aligned
    if (i0 >= last) return;
    const uint64_t bits = *reinterpret_cast<const uint64_t*>(packed + (i0 >> 1)); // (the staging buffer is 8-byte padded)
    char c[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) c[k] = (char)(kLut >> (8 * ((bits >> (4 * k)) & 7u)));
    if (i0 >= first && i0 + 16 <= last)
        *reinterpret_cast<uint4*>(out + i0) = *reinterpret_cast<const uint4*>(c);
    else
        for (int k = 0; k < 16; ++k)
            if (i0 + k >= first && i0 + k < last) out[i0 + k] = c[k];
}

int gwhip_unpack_bases(const uint8_t* packed, char* sequences, int64_t first, int64_t last, gwhip_stream_t stream_)
{
    if (!packed || !sequences || first < 0 || last < first)
    {
        g_last_error = "gwhip_unpack_bases: invalid arguments";
        return (int)hipErrorInvalidValue;
    }
    if (last == first) return 0;
    const int64_t groups = (last - (first & ~int64_t(15)) + 15) / 16;
    hipLaunchKernelGGL(unpack_bases_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, packed, sequences, first, last);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(e, "unpack_bases_kernel launch");
    return 0;
}

int gwhip_myers_occupancy(int device, int* blocks_per_cu)
{
    if (!blocks_per_cu)
    {
        g_last_error = "gwhip_myers_occupancy: null output";
        return (int)hipErrorInvalidValue;
    }
    int prev = 0;
    hipError_t e = hipGetDevice(&prev);
    if (e == hipSuccess && prev != device) e = hipSetDevice(device);
    if (e != hipSuccess) return fail(e, "gwhip_myers_occupancy: device");
    int blocks = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, reinterpret_cast<const void*>(&myers_banded_kernel<false>), 64, 0);
    if (prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return fail(e, "gwhip_myers_occupancy: occupancy query");
    *blocks_per_cu = blocks;
    return 0;
}

int gwhip_myers_banded(const gwhip_myers_args* args, gwhip_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!args || args->n_alignments < 0)
    {
        g_last_error = "gwhip_myers_banded: invalid arguments";
        return (int)hipErrorInvalidValue;
    }
    const int32_t n = args->n_alignments;
    if (n == 0) return 0;
    if (!args->workspace || ((uintptr_t)args->workspace & 255) != 0)
    {
        g_last_error = "gwhip_myers_banded: workspace must be 256-byte aligned";
        return (int)hipErrorInvalidValue;
    }
    const WsPlan p = plan_fixed(n, args->total_sequence_length);
    uint8_t* ws    = (uint8_t*)args->workspace;
    KernelArgs ka{};
    ka.n              = n;
    ka.sequences      = args->sequences;
    ka.starts         = args->sequence_starts;
    ka.max_bandwidths = args->max_bandwidths;
    ka.ws_offsets     = reinterpret_cast<int64_t*>(ws + p.off_ws_offsets);
    ka.run_counts     = reinterpret_cast<int32_t*>(ws + p.off_run_counts);
    ka.slot_ops       = reinterpret_cast<int8_t*>(ws + p.off_slot_ops);
    ka.slot_counts    = reinterpret_cast<int32_t*>(ws + p.off_slot_counts);
    ka.band_cells     = reinterpret_cast<uint64_t*>(ws + p.off_cells);
    int32_t* identity = reinterpret_cast<int32_t*>(ws + p.off_identity);
    ka.order          = args->scheduling_index ? args->scheduling_index : identity;
    ka.ws             = reinterpret_cast<uint32_t*>(ws + p.off_ws);
    ka.metadata       = args->result_metadata;
    ka.index_base     = args->index_base;
    ka.slot_base      = args->first_sequence_offset;

    ka.ws_capacity_words = ((int64_t)args->workspace_bytes - (int64_t)p.off_ws) / 4;
    // sizing ahead of the alignment kernel and scan / compaction behind it go to the caller's side stream when it gave one
    hipStream_t side = args->side_stream ? (hipStream_t)args->side_stream : stream;
    auto hand_over   = [&](hipStream_t from, hipStream_t to) -> hipError_t {
        if (from == to) return hipSuccess;
        // events of this host thread and device, made once and reused round robin (a wait refers to the record that preceded
        // it, so recording the event again later does not disturb it); two creations + destructions per chunk and launch
        // were on the timed path of every align_all() before
        struct EventRing
        {
            int device = -1;
            hipEvent_t ev[8] = {};
            unsigned next = 0; // (never destroyed: a thread's exit may come after the runtime's own teardown)
        };
        thread_local EventRing rings[4];
        int device = 0;
        if (hipError_t e = hipGetDevice(&device); e != hipSuccess) return e;
        EventRing* ring = nullptr;
        for (EventRing& r : rings)
            if (r.device == device || r.device < 0)
            {
                ring = &r;
                break;
            }
        hipEvent_t ev = nullptr;
        bool own      = false;
        if (ring != nullptr)
        {
            ring->device   = device;
            hipEvent_t& slot = ring->ev[ring->next++ & 7];
            if (slot == nullptr)
                if (hipError_t e = hipEventCreateWithFlags(&slot, hipEventDisableTiming); e != hipSuccess) return e;
            ev = slot;
        }
        else // (a thread that has visited more than four devices)
        {
            if (hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming); e != hipSuccess) return e;
            own = true;
        }
        hipError_t e = hipEventRecord(ev, from);
        if (e == hipSuccess) e = hipStreamWaitEvent(to, ev, 0);
        if (own) (void)hipEventDestroy(ev); // (released once the recorded work has completed)
        return e;
    };
    const bool do_sizing = args->phases == 0 || (args->phases & GWHIP_MYERS_SIZING) != 0;
    const bool do_align  = args->phases == 0 || (args->phases & GWHIP_MYERS_ALIGN) != 0;
    if (do_sizing)
    {
        hipStream_t stream = side;
        const int32_t n_waves  = (n + 63) / 64;
        int64_t* offsets       = const_cast<int64_t*>(ka.ws_offsets);
        int64_t* block_totals  = reinterpret_cast<int64_t*>(ws + p.off_scan);
        const int32_t n_blocks = (n_waves + kScanChunk - 1) / kScanChunk;
        hipLaunchKernelGGL(ws_sizes_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, args->sequence_starts, args->max_bandwidths, ka.order, offsets,
                           identity, n);
        hipLaunchKernelGGL(ws_block_totals_kernel, dim3(n_blocks), dim3(kScanBlock), 0, stream, offsets, block_totals, n_waves);
        hipLaunchKernelGGL(scan_totals_kernel<int64_t>, dim3(1), dim3(1024), 0, stream, block_totals, n_blocks, offsets + n_waves,
                           (const int64_t*)nullptr);
        hipLaunchKernelGGL(ws_apply_kernel, dim3(n_blocks), dim3(kScanBlock), 0, stream, offsets, block_totals, n_waves);
    }
    if (!do_align)
    {
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : fail(e, "myers workspace sizing launch");
    }
    if (do_sizing)
        if (hipError_t e = hand_over(side, stream); e != hipSuccess) return fail(e, "gwhip_myers_banded: side stream -> stream");
    // LDS flavour when every pair's pattern table and column state fit one wave's share (<= 1 KiB per lane)
    bool use_lds = false;
    if (args->max_query_length > 0 && args->max_bandwidth_hint > 0)
    {
        const int32_t qwords = (args->max_query_length + kWord - 1) / kWord;
        const int32_t bwords = (std::min(args->max_bandwidth_hint + 2, args->max_query_length) + kWord - 1) / kWord + 1;
        const int32_t per_lane_words = 4 * qwords + 3 * bwords + 16; // + the target window (TargetStream)
        if (per_lane_words <= 252)
        {
            use_lds              = true;
            ka.lds_pattern_words = 4 * qwords;
            ka.lds_band_words    = bwords;
        }
    }
    {
        const char* sk = std::getenv("GWHIP_MYERS_SKIP");
        ka.debug_skip  = sk ? std::atoi(sk) : 0;
    }
    const char* myers_dbg = std::getenv("GWHIP_MYERS_HBM_STATE"); // debugging: force the HBM-state kernel
    if (myers_dbg && myers_dbg[0] == '1') use_lds = false;
    // Small batches of long pairs: G lanes per pair (myers_banded_group_kernel). Worth it while the one-lane kernel
    // would leave SIMDs empty (fewer wavefronts than SIMDs) and a pair is a long chain (queries of 256 bases and more).
    // G = 6 / 8 for batches that then fill the SIMDs (BASELINE configs[1]); G = 16 / 32 / 64 (round 6) while the batch still
    // fits one wavefront per SIMD AND a band attempt may need more words than lanes -- an attempt wider than the group falls
    // back to the one-lane stripes on the group's first lane, which is what made 1024 pairs of 2 kbp at max_bandwidth 1024
    // (band attempts of 5, 9 and 13+ words; the reference benchmark's shape, cudaaligner/benchmarks/main.cpp:81-95) take
    // 15.7 ms: every column of a 13-word attempt was 13 dependent word steps of one lane.
    int group_lanes = 0; // 0: not the group kernel
    {
        const char* gdbg = std::getenv("GWHIP_MYERS_GROUP"); // debugging: 0 = never, 1 = whenever the LDS tables fit
        int cus = 0, dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        const int32_t simds  = 4 * cus;
        const int32_t qwords = args->max_query_length > 0 ? (args->max_query_length + kWord - 1) / kWord : 0;
        // LDS of a block of four wavefronts with G lanes per pair: per pair the pattern table (+ 1: odd stride) and 17 words of
        // target characters, plus the backtrace's column window (224 words for each of the 64 slots of a workspace region)
        // (the wide groups may go without the column window: a band of more than nine words leaves it under the eight columns
        // below which the walk ignores it anyway, and queries of 64 kbp only fit that way -- four pairs per block, 131 KB of tables)
        auto lds_bytes = [&](int G, int tile_words) {
            const size_t pairs = G == 6 ? 40 : 256 / (size_t)G;
            return (pairs * (size_t)(4 * qwords + 1) + pairs * 17 + 64 * (size_t)tile_words) * sizeof(uint32_t);
        };
        auto tile_of = [&](int G) -> int { // words of the backtrace's column window per pair, -1: the tables do not fit
            const size_t limit = G <= 8 ? (size_t)80 * 1024 : (size_t)156 * 1024; // two blocks / one block per CU
            if (qwords <= 0) return -1;
            if (lds_bytes(G, 224) <= limit) return 224;
            if (G >= 16 && lds_bytes(G, 0) <= limit) return 0;
            return -1;
        };
        auto fits = [&](int G) { return tile_of(G) >= 0; };
        bool use_group = (n + 63) / 64 < simds && args->max_query_length >= 256;
        if (gdbg && gdbg[0] == '0') use_group = false;
        if (gdbg && gdbg[0] == '1') use_group = true;
        if (use_group)
        {
            // (four lanes per pair with two words per lane -- half the wavefronts for 10 000 pairs -- was measured and is no
            // faster: 1.71 vs 1.68 ms on configs[1], profiles/r03_h_aligner_group_lanes.txt.) Six lanes per pair (ten pairs per
            // wavefront) when that is what puts the batch on one wavefront per SIMD.
            int G = (n > 8 * simds && n <= 10 * simds) ? 6 : 8;
            // the widest band an attempt can reach, in words
            const int32_t bw_cap      = args->max_bandwidth_hint > 0 ? std::min(args->max_bandwidth_hint, args->max_query_length) : args->max_query_length;
            const int32_t worst_words = (bw_cap + kWord - 1) / kWord;
            while (G >= 8 && G < 64 && G < worst_words && (int64_t)n * (2 * G) / 64 <= simds) G *= 2;
            if (const char* gl = std::getenv("GWHIP_MYERS_GROUP_LANES")) // forces the choice (6, 8, 16, 32, 64)
            {
                const int v = std::atoi(gl);
                if (v == 6 || v == 8 || v == 16 || v == 32 || v == 64) G = v;
            }
            // tables that do not fit: fewer pairs per block (more lanes per pair) until they do, if the batch allows it
            for (int g = G; g <= 64 && group_lanes == 0; g = g == 6 ? 8 : 2 * g)
                if (fits(g) && (g == G || (int64_t)n * g / 64 <= 2 * simds)) group_lanes = g;
        }
        if (group_lanes != 0)
        {
            ka.lds_pattern_words = 4 * qwords;
            ka.lds_band_words    = tile_of(group_lanes); // words of the backtrace's column window per pair
            const size_t lds     = lds_bytes(group_lanes, ka.lds_band_words);
            auto launch = [&](auto kernel, int pairs) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(lds, 80 * 1024));
                hipLaunchKernelGGL(kernel, dim3((n + pairs - 1) / pairs), dim3(256), lds, stream, ka);
            };
            switch (group_lanes)
            {
            case 6: launch(&myers_banded_group_kernel<6, 4>, 40); break;
            case 8: launch(&myers_banded_group_kernel<8, 4>, 32); break;
            case 16: launch(&myers_banded_group_kernel<16, 4>, 16); break;
            case 32: launch(&myers_banded_group_kernel<32, 4>, 8); break;
            default: launch(&myers_banded_group_kernel<64, 4>, 4); break;
            }
        }
    }
    if (group_lanes != 0) {}
    else if (use_lds)
        hipLaunchKernelGGL(myers_banded_kernel<true>, dim3((n + 63) / 64), dim3(64),
                           (size_t)(ka.lds_pattern_words + 3 * ka.lds_band_words + 16) * 64 * sizeof(uint32_t), stream, ka);
    else
        hipLaunchKernelGGL(myers_banded_kernel<false>, dim3((n + 63) / 64), dim3(64), 0, stream, ka);
    if (hipError_t e = hand_over(stream, side); e != hipSuccess) return fail(e, "gwhip_myers_banded: stream -> side stream");
    stream = side; // (everything below)
    {
        int32_t* block_totals  = reinterpret_cast<int32_t*>(ws + p.off_scan);
        const int32_t n_blocks = (n + kScanChunk - 1) / kScanChunk;
        hipLaunchKernelGGL(scan_block_totals_kernel, dim3(n_blocks), dim3(kScanBlock), 0, stream, ka.run_counts, block_totals, n);
        hipLaunchKernelGGL(scan_totals_kernel<int32_t>, dim3(1), dim3(1024), 0, stream, block_totals, n_blocks, args->result_starts + n,
                           args->result_starts_base);
        hipLaunchKernelGGL(scan_apply_kernel, dim3(n_blocks), dim3(kScanBlock), 0, stream, ka.run_counts, block_totals, args->result_starts, n);
    }
    hipLaunchKernelGGL(compact_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, ka, args->results, args->result_counts,
                       args->result_starts);
    {
        const bool runs = args->results_host && args->result_counts_host && args->results_host_capacity > 0;
        // (a small grid: the link needs few stores in flight, and a wide one slows the alignment kernel it runs beside)
        static const int mirror_blocks = [] { const char* e = std::getenv("GWHIP_MIRROR_BLOCKS"); return e ? std::max(1, std::atoi(e)) : 16; }();
        if (runs || args->result_starts_host)
            hipLaunchKernelGGL(mirror_runs_kernel, dim3(std::max(1, std::min((n + 63) / 64, mirror_blocks))), dim3(256), 0, stream, args->results, args->result_counts,
                               args->result_starts, n, runs ? args->results_host : nullptr, args->result_counts_host, args->results_host_capacity,
                               args->result_metadata, args->result_starts_host, args->result_metadata_host);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(e, "myers kernels launch");
    if (args->band_cells)
    {
        e = hipMemcpyAsync(args->band_cells, ka.band_cells, (size_t)n * 8, hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return fail(e, "band_cells copy");
    }
    return 0;
}


int gwhip_myers_test_patterns(const char* query_d, int32_t query_length, uint32_t* patterns_d, gwhip_stream_t stream_)
{
    if (!query_d || !patterns_d || query_length < 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(myers_patterns_hook_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, query_d, query_length, patterns_d);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(e, "myers_patterns_hook_kernel launch");
}

int gwhip_myers_test_get_pattern(const char* query_d, int32_t query_length, int32_t word_index, char x, int32_t reverse,
                                 uint32_t* scratch_d, uint32_t* out32_d, gwhip_stream_t stream_)
{
    if (!query_d || !scratch_d || !out32_d || query_length < 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(myers_get_pattern_hook_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, query_d, query_length, word_index, x,
                       reverse, scratch_d, out32_d);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(e, "myers_get_pattern_hook_kernel launch");
}

size_t gwhip_myers_test_banded_matrices_words(int32_t query_length, int32_t target_length, int32_t band_width)
{
    const int64_t nwb = (band_width + kWord - 1) / kWord, nw = (query_length + kWord - 1) / kWord;
    return (size_t)(64 * (3 * nwb * ((int64_t)target_length + 1) + 4 * nw));
}

int gwhip_myers_test_banded_matrices(const char* query_d, const char* target_d, int32_t query_length, int32_t target_length,
                                     int32_t band_width, int32_t p, uint32_t* workspace_d, int32_t* diagonals_d, gwhip_stream_t stream_)
{
    if (!query_d || !target_d || !workspace_d || !diagonals_d || query_length <= 0 || target_length <= 0 || band_width <= 0)
        return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(myers_banded_matrices_hook_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, query_d, target_d, query_length,
                       target_length, band_width, p, workspace_d, diagonals_d);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(e, "myers_banded_matrices_hook_kernel launch");
}

size_t gwhip_hirschberg_myers_workspace_bytes(int32_t n_alignments, const int64_t* sequence_starts_host, int32_t max_query_length)
{
    if (n_alignments <= 0) return 256;
    const int64_t max_elems = (int64_t)((std::max(max_query_length, 1) + kWord - 1) / kWord) * (kHbSwitchToMyers + 1);
    const int32_t n_waves   = (n_alignments + 63) / 64;
    int64_t words           = 0;
    for (int32_t wv = 0; wv < n_waves; wv++)
    {
        int32_t qw = 0, tm = 0;
        for (int32_t i = wv * 64; i < std::min(n_alignments, wv * 64 + 64); i++)
        {
            qw = std::max(qw, ((int32_t)(sequence_starts_host[2 * i + 1] - sequence_starts_host[2 * i]) + kWord - 1) / kWord);
            tm = std::max(tm, (int32_t)(sequence_starts_host[2 * i + 2] - sequence_starts_host[2 * i + 1]));
        }
        words += 64 * hb_lane_words(qw, tm, max_elems);
        // the span path of long single pairs keeps its part lists, rows and pieces behind the (only) wave region
        if (n_alignments <= kSpanMaxPairs) words += (int64_t)n_alignments * hb_span_words(qw, tm, max_query_length);
    }
    return 256 + ((size_t)n_waves + 1) * 8 + 256 + (size_t)words * 4 + 256;
}

int gwhip_hirschberg_myers(const gwhip_hirschberg_args* args, gwhip_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!args || args->n_alignments < 0 || args->max_query_length < 0)
    {
        g_last_error = "gwhip_hirschberg_myers: invalid arguments";
        return (int)hipErrorInvalidValue;
    }
    const int32_t n = args->n_alignments;
    if (n == 0) return 0;
    const int32_t n_waves = (n + 63) / 64;
    uint8_t* ws           = (uint8_t*)args->workspace;
    const size_t off_ws   = ((((size_t)n_waves + 1) * 8) + 255) / 256 * 256;
    if (args->workspace_bytes < off_ws + 256)
    {
        g_last_error = "gwhip_hirschberg_myers: workspace too small";
        return (int)hipErrorInvalidValue;
    }
    HirschbergArgs ka{};
    ka.n                 = n;
    ka.sequences         = args->sequences;
    ka.starts            = args->sequence_starts;
    ka.max_query_length  = args->max_query_length;
    ka.results           = args->results;
    ka.result_lengths    = args->result_lengths;
    ka.wave_offsets      = reinterpret_cast<int64_t*>(ws);
    ka.ws                = reinterpret_cast<uint32_t*>(ws + off_ws);
    ka.ws_capacity_words = ((int64_t)args->workspace_bytes - (int64_t)off_ws) / 4;
    const int64_t max_elems = (int64_t)((std::max(args->max_query_length, 1) + kWord - 1) / kWord) * (kHbSwitchToMyers + 1);
    hipLaunchKernelGGL(hb_offsets_kernel, dim3(1), dim3(1024), 0, stream, args->sequence_starts,
                       const_cast<int64_t*>(ka.wave_offsets), n, max_elems);
    const int32_t qwords = (std::max(args->max_query_length, 1) + kWord - 1) / kWord;
    // One wavefront per pair unless the batch alone fills the device with one lane per pair many times over
    // (GWHIP_HIRSCHBERG_WAVE = 0 / 1 forces the choice). Its LDS: range stack + (pv, mv, 4 pattern words) per word of the
    // longest query part (half the longest query), in chunks of 64 words.
    const int32_t part_chunks = std::max(1, ((qwords + 1) / 2 + 1 + 63) / 64);
    const size_t wave_lds     = (size_t)(4 * kHbStackEntries + 6 * 64 * part_chunks + 3 * kHwLeafElems) * sizeof(uint32_t);
    // LDS a block may ask for on this device (queried once per device: 160 KB on gfx950) minus a margin for the kernels' static
    // LDS; a launch that would not fit falls through to the kernels below instead of failing
    static std::atomic<int> lds_limit_of_device[64]; // zero-initialised; filled by whichever host thread asks first (same value)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && lds_limit_of_device[dev].load(std::memory_order_relaxed) == 0)
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) v = 64 * 1024;
        lds_limit_of_device[dev].store(v, std::memory_order_relaxed);
    }
    const size_t lds_limit    = (size_t)((dev >= 0 && dev < 64) ? lds_limit_of_device[dev].load(std::memory_order_relaxed) : 64 * 1024) - 10 * 1024;
    bool use_wave             = n <= 262144 && wave_lds <= lds_limit;
    {
        const char* hw = std::getenv("GWHIP_HIRSCHBERG_WAVE");
        if (hw && hw[0] == '0') use_wave = false;
        if (hw && hw[0] == '1') use_wave = wave_lds <= lds_limit;
    }
    if (use_wave && wave_lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&hirschberg_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wave_lds) != hipSuccess)
    {
        (void)hipGetLastError();
        use_wave = false; // the one-lane kernels below need no more than 60 KB
    }
    if (use_wave)
    {
        ka.lds_state_words = part_chunks;
        const size_t lds_request = wave_lds;
        // queries of up to 2 048 bases: the level-by-level kernel first (GWHIP_HIRSCHBERG_LEVELS=0: depth-first only), the
        // depth-first kernel behind it for what that one leaves (targets beyond its LDS rows, list overflows)
        {
            const char* lv = std::getenv("GWHIP_HIRSCHBERG_LEVELS");
            if (args->max_query_length <= kLvMaxQuery && !(lv && lv[0] == '0'))
            {
                const size_t lv_lds = (size_t)lv_layout(std::max(args->max_query_length, 1)).total;
                bool lv_ok          = lv_lds <= lds_limit;
                if (lv_ok && lv_lds > 48 * 1024 &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(&hirschberg_levels_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lv_lds) != hipSuccess)
                {
                    (void)hipGetLastError();
                    lv_ok = false; // the depth-first kernel behind it takes every pair
                }
                if (lv_ok)
                {
                    ka.levels_first = 1;
                    hipLaunchKernelGGL(hirschberg_levels_kernel, dim3(n), dim3(64), lv_lds, stream, ka);
                }
            }
        }
        // long single pairs (queries beyond kSpanPartQuery in a batch of at most kSpanMaxPairs pairs): the top of the tree level by
        // level across blocks, then one wavefront per part (GWHIP_HIRSCHBERG_SPAN=0: the depth-first kernel alone)
        {
            const char* sp = std::getenv("GWHIP_HIRSCHBERG_SPAN");
            ka.span_levels = (n <= kSpanMaxPairs && !(sp && sp[0] == '0')) ? hb_span_levels(args->max_query_length) : 0;
        }
        if (lds_request > 48 * 1024)
        {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hirschberg_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_request);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hb_span_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_request);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hb_span_parts_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_request);
        }
        hipLaunchKernelGGL(hirschberg_wave_kernel, dim3(n), dim3(64), lds_request, stream, ka);
        if (ka.span_levels > 0)
        {
            hipLaunchKernelGGL(hb_span_init_kernel, dim3(n), dim3(64), 0, stream, ka);
            for (int32_t level = 0; level < ka.span_levels; ++level)
            {
                // the longest half of this level in 64-word chunks: one wavefront per chunk (at most kMwMaxWaves, which then own
                // several) -- GWHIP_HIRSCHBERG_SPAN=1: one wavefront per half, chunk after chunk (A/B)
                const int32_t half   = ((args->max_query_length >> level) + 1) / 2 + 1;
                const int32_t chunks = (((half + kWord - 1) / kWord) + 63) / 64;
                const char* sp       = std::getenv("GWHIP_HIRSCHBERG_SPAN");
                if (chunks > kMwMaxChunks || (sp && sp[0] == '1'))
                    hipLaunchKernelGGL(hb_span_rows_kernel, dim3(2u << level, n), dim3(64), lds_request, stream, ka, level);
                else
                    hipLaunchKernelGGL(hb_span_rows_mw_kernel, dim3(2u << level, n), dim3(64 * std::max(1, std::min(chunks, kMwMaxWaves))), 0, stream, ka, level);
                hipLaunchKernelGGL(hb_span_split_kernel, dim3(1u << level, n), dim3(64), 0, stream, ka, level);
            }
            hipLaunchKernelGGL(hb_span_parts_kernel, dim3(1u << ka.span_levels, n), dim3(64), lds_request, stream, ka);
            hipLaunchKernelGGL(hb_span_join_kernel, dim3(n), dim3(256), 0, stream, ka);
        }
    }
    else if ((size_t)qwords * 6 * 64 * sizeof(uint32_t) <= 60 * 1024)
    {
        ka.lds_state_words = qwords;
        hipLaunchKernelGGL(hirschberg_myers_kernel<true>, dim3(n_waves), dim3(64), (size_t)qwords * 6 * 64 * sizeof(uint32_t), stream, ka);
    }
    else
        hipLaunchKernelGGL(hirschberg_myers_kernel<false>, dim3(n_waves), dim3(64), 0, stream, ka);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(e, "hirschberg kernels launch");
    return 0;
}

} // extern "C"
