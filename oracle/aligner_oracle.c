/*
 * aligner_oracle.c -- CPU restatement of the reference's banded Myers aligner (cudaaligner/src/myers_gpu.cu).
 *
 * TEST INFRASTRUCTURE ONLY (see aligner_oracle.h). The restatement keeps the reference's own decomposition:
 * 32-bit words, "warp iterations" of 32 words with the cross-lane add / shift helpers emulated lane by lane,
 * so that band geometry, the implicit worst-case row 0 and the three-phase backtrace are literal.
 * Pinned by Test_AlignerGlobal.cpp:79-148, Test_ApproximateBandedMyers.cpp:72-170 and, for edit distances of
 * optimal results, by the reference's own CPU code built into oracle/_ref (Makefile.ref); and by the reference's own myers_gpu.cu
 * run on the CPU (oracle/simt -> oracle/_ref/libref_cudaaligner_simt.so): tests/golden/reference_simt_alignments.json.gz, and every
 * pair of the configs[1] and configs[4] goldens (tests/golden/reference_simt_config_check.json), tests/test_reference_simt.py.
 */
#include "aligner_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

typedef uint32_t WordType;
#define WORD_SIZE 32
#define WARP_SIZE 32

static int32_t ceiling_divide(int32_t a, int32_t b) { return (a + b - 1) / b; }
static int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
static int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }
static int32_t iabs(int32_t a) { return a < 0 ? -a : a; }

/* column-major band matrices: element (word idx, column t) */
typedef struct
{
    WordType* pv;
    WordType* mv;
    int32_t* score;
    int32_t n_rows; /* n_words_band */
    int32_t n_cols; /* target_size + 1 */
} band_t;
#define PV(b, i, t) ((b)->pv[(size_t)(t) * (b)->n_rows + (i)])
#define MV(b, i, t) ((b)->mv[(size_t)(t) * (b)->n_rows + (i)])
#define SC(b, i, t) ((b)->score[(size_t)(t) * (b)->n_rows + (i)])

/* myers_generate_query_pattern, myers_gpu.cu:196-208 */
static WordType generate_query_pattern(char x, const char* query, int32_t query_size, int32_t offset)
{
    const int32_t max_i = imin(query_size - offset, WORD_SIZE);
    WordType r          = 0;
    for (int32_t i = 0; i < max_i; ++i)
        if (x == query[i + offset]) r |= ((WordType)1 << i);
    return r;
}

/* get_query_pattern, myers_gpu.cu:210-241; patterns[word * 4 + char_idx] */
static WordType get_query_pattern(const WordType* patterns, int32_t n_words, int32_t idx, int32_t query_begin_offset, char x)
{
    const int32_t char_idx   = ((unsigned char)x >> 1) & 0x3;
    const int32_t idx_offset = query_begin_offset / WORD_SIZE;
    const int32_t shift      = query_begin_offset % WORD_SIZE;
    WordType r               = (idx + idx_offset < n_words) ? patterns[(idx + idx_offset) * 4 + char_idx] : 0;
    if (shift != 0)
    {
        r >>= shift;
        if (idx + idx_offset + 1 < n_words) r |= patterns[(idx + idx_offset + 1) * 4 + char_idx] << (WORD_SIZE - shift);
    }
    return r;
}

/* get_myers_score, myers_gpu.cu:243-255 */
static int32_t get_myers_score(int32_t i, int32_t j, const band_t* b, WordType last_entry_mask)
{
    const int32_t word_idx = (i - 1) / WORD_SIZE;
    const int32_t bit_idx  = (i - 1) % WORD_SIZE;
    int32_t s              = SC(b, word_idx, j);
    WordType mask          = bit_idx == 31 ? 0 : ((~(WordType)1) << bit_idx);
    if (word_idx == b->n_rows - 1) mask &= last_entry_mask;
    s -= __builtin_popcount(mask & PV(b, word_idx, j));
    s += __builtin_popcount(mask & MV(b, word_idx, j));
    return s;
}

/* One warp iteration (<= 32 lanes): warp_add_sync :104-130, warp_leftshift_sync :78-89 over active lanes. */
static void chunk_add(const WordType* a, const WordType* b, WordType* r, int n)
{
    uint64_t carry = 0;
    for (int l = 0; l < n; l++)
    {
        uint64_t s = (uint64_t)a[l] + (uint64_t)b[l] + carry;
        r[l]       = (WordType)s;
        carry      = s >> 32;
    }
}
static void chunk_shl1(WordType* v, int n)
{
    WordType in = 0;
    for (int l = 0; l < n; l++)
    {
        WordType out = v[l] >> (WORD_SIZE - 1);
        v[l]         = (v[l] << 1) | in;
        in           = out;
    }
}

/* myers_advance_block / myers_advance_block2 (:132-194) for all lanes of one warp iteration.
   carry_in applies to lane 0 only (other lanes carry 0). out_x[l] = delta at hbit[l]; out_y[l] = delta at hbit[l]<<1. */
static void advance_chunk(int n, const WordType* hbit, const WordType* eq_in, WordType* pv, WordType* mv, int32_t carry_in0,
                          int32_t* out_x, int32_t* out_y)
{
    WordType eq[WARP_SIZE], xv[WARP_SIZE], a[WARP_SIZE], xh[WARP_SIZE], ph[WARP_SIZE], mh[WARP_SIZE];
    for (int l = 0; l < n; l++)
    {
        eq[l] = eq_in[l];
        xv[l] = eq[l] | mv[l];
        if (l == 0 && carry_in0 < 0) eq[l] |= 1u;
        a[l] = eq[l] & pv[l];
    }
    chunk_add(a, pv, xh, n);
    for (int l = 0; l < n; l++)
    {
        xh[l] = (xh[l] ^ pv[l]) | eq[l];
        ph[l] = mv[l] | (~(xh[l] | pv[l]));
        mh[l] = pv[l] & xh[l];
        out_x[l] = ((ph[l] & hbit[l]) == 0 ? 0 : 1) - ((mh[l] & hbit[l]) == 0 ? 0 : 1);
        if (out_y)
        {
            WordType h2 = hbit[l] << 1;
            out_y[l]    = ((ph[l] & h2) == 0 ? 0 : 1) - ((mh[l] & h2) == 0 ? 0 : 1);
        }
    }
    chunk_shl1(ph, n);
    chunk_shl1(mh, n);
    if (carry_in0 < 0) mh[0] |= 1u;
    if (carry_in0 > 0) ph[0] |= 1u;
    for (int l = 0; l < n; l++)
    {
        pv[l] = mh[l] | (~(xv[l] | ph[l]));
        mv[l] = ph[l] & xv[l];
    }
}

/* myers_compute_scores_horizontal_band_impl :629-674 */
static void horizontal_band(band_t* b, const WordType* patterns, int32_t n_words_query, const char* target, int32_t t_begin,
                            int32_t t_end, int32_t width, int32_t n_words, int32_t pattern_idx_offset)
{
    for (int32_t t = t_begin; t < t_end; ++t)
    {
        int32_t warp_carry = 1; /* worst case for the top border of the band */
        for (int32_t base = 0; base < n_words; base += WARP_SIZE)
        {
            const int n = imin(WARP_SIZE, n_words - base);
            WordType pv[WARP_SIZE], mv[WARP_SIZE], hb[WARP_SIZE], eq[WARP_SIZE];
            int32_t ox[WARP_SIZE];
            for (int l = 0; l < n; l++)
            {
                const int32_t idx = base + l;
                pv[l] = PV(b, idx, t - 1);
                mv[l] = MV(b, idx, t - 1);
                hb[l] = (WordType)1 << (idx == (n_words - 1) ? width - (n_words - 1) * WORD_SIZE - 1 : WORD_SIZE - 1);
                eq[l] = get_query_pattern(patterns, n_words_query, idx, pattern_idx_offset, target[t - 1]);
            }
            advance_chunk(n, hb, eq, pv, mv, warp_carry, ox, NULL);
            for (int l = 0; l < n; l++)
            {
                const int32_t idx = base + l;
                SC(b, idx, t)     = SC(b, idx, t - 1) + ox[l];
                PV(b, idx, t)     = pv[l];
                MV(b, idx, t)     = mv[l];
            }
            /* carry into the next warp iteration: lane 31's carry_out, only when this iteration was a full warp */
            warp_carry = (n == WARP_SIZE) ? ox[WARP_SIZE - 1] : 0;
        }
    }
}

/* myers_compute_scores_diagonal_band_impl :676-751 */
static void diagonal_band(band_t* b, const WordType* patterns, int32_t n_words_query, const char* target, int32_t t_begin,
                          int32_t t_end, int32_t band_width, int32_t n_words_band, int32_t pattern_idx_offset)
{
    for (int32_t t = t_begin; t < t_end; ++t)
    {
        int32_t carry = 1;
        for (int32_t base = 0; base < n_words_band; base += WARP_SIZE)
        {
            const int n = imin(WARP_SIZE, n_words_band - base);
            WordType pv[WARP_SIZE], mv[WARP_SIZE], hb[WARP_SIZE], eq[WARP_SIZE];
            int32_t ox[WARP_SIZE], oy[WARP_SIZE];
            /* warp_rightshift_sync over the active lanes of this iteration, plus the bit fetched from the next
               iteration's first word by lane 31 of a full warp (:703-714) */
            for (int l = 0; l < n; l++)
            {
                const int32_t idx = base + l;
                WordType p = PV(b, idx, t - 1) >> 1, m = MV(b, idx, t - 1) >> 1;
                if (l + 1 < n)
                {
                    p |= PV(b, idx + 1, t - 1) << (WORD_SIZE - 1);
                    m |= MV(b, idx + 1, t - 1) << (WORD_SIZE - 1);
                }
                else if (l == WARP_SIZE - 1 && n == WARP_SIZE && idx < n_words_band - 1)
                {
                    p |= PV(b, idx + 1, t - 1) << (WORD_SIZE - 1);
                    m |= MV(b, idx + 1, t - 1) << (WORD_SIZE - 1);
                }
                pv[l] = p;
                mv[l] = m;
                eq[l] = get_query_pattern(patterns, n_words_query, idx, pattern_idx_offset + t - t_begin + 1, target[t - 1]);
                const WordType delta_right_bit =
                    (WordType)1 << (idx == (n_words_band - 1) ? band_width - (n_words_band - 1) * WORD_SIZE - 2 : WORD_SIZE - 2);
                hb[l] = delta_right_bit;
                if (idx == n_words_band - 1)
                {
                    pv[l] |= (delta_right_bit << 1);
                    mv[l] &= ~(delta_right_bit << 1);
                }
            }
            advance_chunk(n, hb, eq, pv, mv, carry, ox, oy);
            for (int l = 0; l < n; l++)
            {
                const int32_t idx      = base + l;
                const WordType ddb     = hb[l] << 1;
                const int32_t delta_dn = ((pv[l] & ddb) == 0 ? 0 : 1) - ((mv[l] & ddb) == 0 ? 0 : 1);
                SC(b, idx, t)          = SC(b, idx, t - 1) + ox[l] + delta_dn;
                PV(b, idx, t)          = pv[l];
                MV(b, idx, t)          = mv[l];
            }
            carry = (n == WARP_SIZE) ? oy[WARP_SIZE - 1] : 0;
        }
    }
}

/* myers_compute_scores_edit_dist_banded :753-846 */
static void compute_scores_banded(int32_t* diagonal_begin, int32_t* diagonal_end, band_t* b, const WordType* patterns,
                                  int32_t n_words_query, const char* target, int32_t target_size, int32_t query_size,
                                  int32_t band_width, int32_t n_words_band, int32_t p)
{
    for (int32_t idx = 0; idx < n_words_band; idx++)
    {
        PV(b, idx, 0) = ~(WordType)0;
        MV(b, idx, 0) = 0;
        SC(b, idx, 0) = imin((idx + 1) * WORD_SIZE, band_width);
    }
    if (band_width >= query_size)
    {
        *diagonal_begin = target_size + 1;
        *diagonal_end   = target_size + 1;
        horizontal_band(b, patterns, n_words_query, target, 1, target_size + 1, query_size, n_words_band, 0);
    }
    else
    {
        const int32_t symmetric_band = (band_width - imin(1 + 2 * p + iabs(target_size - query_size), query_size) == 0) ? 1 : 0;
        *diagonal_begin = query_size < target_size ? target_size - query_size + p + 2 : p + 2 + (1 - symmetric_band);
        *diagonal_end   = query_size < target_size ? query_size - p + symmetric_band : query_size - (query_size - target_size) - p + 1;
        horizontal_band(b, patterns, n_words_query, target, 1, *diagonal_begin, band_width, n_words_band, 0);
        diagonal_band(b, patterns, n_words_query, target, *diagonal_begin, *diagonal_end, band_width, n_words_band, 0);
        horizontal_band(b, patterns, n_words_query, target, *diagonal_end, target_size + 1, band_width, n_words_band, query_size - band_width);
    }
}

#define EMIT(R)                                                                                                        \
    do                                                                                                                 \
    {                                                                                                                  \
        int8_t _r = (R);                                                                                               \
        if (prev_r != _r)                                                                                              \
        {                                                                                                              \
            if (prev_r != -1)                                                                                          \
            {                                                                                                          \
                path[pos]       = prev_r;                                                                              \
                path_count[pos] = r_count;                                                                             \
                ++pos;                                                                                                 \
            }                                                                                                          \
            prev_r  = _r;                                                                                              \
            r_count = 0;                                                                                               \
        }                                                                                                              \
        ++r_count;                                                                                                     \
    } while (0)

/* Scalar model of the kernels' backtrace step (gwhip_myers.hip, backtrace_banded::fetch3): not three cell scores as the
 * reference reads them, but `left` = cell (i2, j2) alone, `diag` = the cell above it = left minus the vertical delta of row
 * i2 (a bit of pv / mv of the same word), `above` = the walk's own score minus the vertical delta of the current row in the
 * current column. Evaluated next to the reference's three reads in every step of every backtrace of this oracle, wherever
 * the kernel uses the value (the reference's formula cases override it elsewhere); a difference counts as a mismatch
 * (aligner_oracle_delta_identity_mismatches, asserted 0 by the tests). */
static int64_t g_delta_identity_mismatches = 0;
int64_t aligner_oracle_delta_identity_mismatches(void) { return g_delta_identity_mismatches; }

static int32_t vertical_delta(const band_t* b, int32_t i, int32_t j)
{
    const int32_t w = (i - 1) / WORD_SIZE, bit = (i - 1) % WORD_SIZE;
    return (int32_t)((PV(b, w, j) >> bit) & 1u) - (int32_t)((MV(b, w, j) >> bit) & 1u);
}
static void model_step(const band_t* b, WordType mask, int32_t rows, int32_t i, int32_t j, int32_t i2, int32_t j2, int32_t myscore,
                       int32_t* above, int32_t* diag)
{
    const int32_t ia  = i < 1 ? 1 : (i > rows ? rows : i);
    const int in2     = i2 >= 1 && i2 <= rows;
    int32_t il        = in2 ? i2 : i2 - 1;
    il                = il < 1 ? 1 : (il > rows ? rows : il);
    const int32_t jl  = j2 < 0 ? 0 : j2, ja = j < 0 ? 0 : j;
    const int32_t s   = get_myers_score(il, jl, b, mask);
    *diag             = in2 ? s - vertical_delta(b, il, jl) : s;
    *above            = myscore - vertical_delta(b, ia, ja);
}

/* myers_backtrace_banded :444-627 */
static int32_t backtrace_banded(int8_t* path, int32_t* path_count, const band_t* b, int32_t diagonal_begin, int32_t diagonal_end,
                                int32_t band_width, int32_t target_size, int32_t query_size)
{
    const int32_t out_of_band = INT32_MAX - 1;
    int32_t i = band_width, j = target_size;
    const WordType last_entry_mask = band_width % WORD_SIZE != 0 ? (((WordType)1 << (band_width % WORD_SIZE)) - 1) : ~(WordType)0;
    const int32_t last_diagonal_score = diagonal_end < 2 ? out_of_band : get_myers_score(1, diagonal_end - 2, b, last_entry_mask) + 2;
    int32_t myscore = i > 0 ? SC(b, (i - 1) / WORD_SIZE, j) : 0;
    int32_t pos = 0, r_count = 0;
    int8_t prev_r = -1;
    (void)query_size;
    while (j >= diagonal_end)
    {
        const int32_t above = i <= 1 ? (last_diagonal_score + j - diagonal_end) : get_myers_score(i - 1, j, b, last_entry_mask);
        const int32_t diag  = i <= 1 ? (last_diagonal_score + j - 1 - diagonal_end) : get_myers_score(i - 1, j - 1, b, last_entry_mask);
        const int32_t left  = i < 1 ? (last_diagonal_score + j - 1 - diagonal_end) : get_myers_score(i, j - 1, b, last_entry_mask);
        if (i >= 2)
        {
            int32_t ka, kd;
            model_step(b, last_entry_mask, band_width, i, j, i, j - 1, myscore, &ka, &kd);
            g_delta_identity_mismatches += (ka != above) + (kd != diag);
        }
        int8_t r;
        if (left + 1 == myscore) { r = ALN_INSERTION; myscore = left; --j; }
        else if (above + 1 == myscore) { r = ALN_DELETION; myscore = above; --i; }
        else { r = (diag == myscore ? ALN_MATCH : ALN_MISMATCH); myscore = diag; --i; --j; }
        EMIT(r);
    }
    while (j >= diagonal_begin)
    {
        const int32_t above = i <= 1 ? out_of_band : get_myers_score(i - 1, j, b, last_entry_mask);
        const int32_t diag  = i <= 0 ? j - 1 : get_myers_score(i, j - 1, b, last_entry_mask);
        const int32_t left  = i >= band_width ? out_of_band : get_myers_score(i + 1, j - 1, b, last_entry_mask);
        if (i >= 1)
        {
            int32_t ka, kd;
            model_step(b, last_entry_mask, band_width, i, j, i + 1, j - 1, myscore, &ka, &kd);
            g_delta_identity_mismatches += (i >= 2 && ka != above) + (kd != diag);
        }
        int8_t r;
        if (left + 1 == myscore) { r = ALN_INSERTION; myscore = left; ++i; --j; }
        else if (above + 1 == myscore) { r = ALN_DELETION; myscore = above; --i; }
        else { r = (diag == myscore ? ALN_MATCH : ALN_MISMATCH); myscore = diag; --j; }
        EMIT(r);
    }
    while (i > 0 && j > 0)
    {
        const int32_t above = i == 1 ? j : get_myers_score(i - 1, j, b, last_entry_mask);
        const int32_t diag  = i == 1 ? j - 1 : get_myers_score(i - 1, j - 1, b, last_entry_mask);
        const int32_t left  = i > band_width ? out_of_band : get_myers_score(i, j - 1, b, last_entry_mask);
        if (i >= 2)
        {
            int32_t ka, kd;
            model_step(b, last_entry_mask, band_width, i, j, i, j - 1, myscore, &ka, &kd);
            g_delta_identity_mismatches += (ka != above) + (kd != diag);
        }
        int8_t r;
        if (left + 1 == myscore) { r = ALN_INSERTION; myscore = left; --j; }
        else if (above + 1 == myscore) { r = ALN_DELETION; myscore = above; --i; }
        else { r = (diag == myscore ? ALN_MATCH : ALN_MISMATCH); myscore = diag; --i; --j; }
        EMIT(r);
    }
    if (i > 0)
    {
        if (prev_r != ALN_DELETION)
        {
            if (prev_r != -1) { path[pos] = prev_r; path_count[pos] = r_count; ++pos; }
            prev_r  = ALN_DELETION;
            r_count = 0;
        }
        r_count += i;
    }
    if (j > 0)
    {
        if (prev_r != ALN_INSERTION)
        {
            if (prev_r != -1) { path[pos] = prev_r; path_count[pos] = r_count; ++pos; }
            prev_r  = ALN_INSERTION;
            r_count = 0;
        }
        r_count += j;
    }
    if (r_count != 0) { path[pos] = prev_r; path_count[pos] = r_count; ++pos; }
    return pos;
}

/* AlignerGlobalMyersBanded::add_alignment clamp, aligner_global_myers_banded.cpp:174-178 */
int32_t aligner_oracle_host_max_bandwidth(int32_t max_bandwidth, int32_t query_length)
{
    if (max_bandwidth > query_length) max_bandwidth = (query_length % WORD_SIZE == 1 ? query_length + 1 : query_length);
    return max_bandwidth;
}

/* per-alignment body of myers_banded_kernel :897-1021 */
int32_t aligner_oracle_myers_banded(const char* query, int32_t query_size, const char* target, int32_t target_size,
                                    int32_t max_bandwidth, int8_t* ops, int32_t* counts, int32_t* n_runs, int32_t* is_optimal,
                                    int64_t* band_cells)
{
    *n_runs     = 0;
    *is_optimal = 0;
    if (max_bandwidth - 1 < iabs(target_size - query_size) && query_size != 0 && target_size != 0) return 1; /* no result */
    if (target_size == 0 || query_size == 0)
    {
        *is_optimal = 1;
        if (query_size == 0 && target_size == 0) return 0; /* success, empty alignment */
        ops[0]    = query_size == 0 ? ALN_INSERTION : ALN_DELETION;
        counts[0] = query_size + target_size;
        *n_runs   = 1;
        return 0;
    }
    const int32_t n_words = ceiling_divide(query_size, WORD_SIZE);
    WordType* patterns    = (WordType*)calloc((size_t)n_words * 4 + 4, sizeof(WordType));
    for (int32_t idx = 0; idx < n_words; idx++)
    {
        patterns[idx * 4 + 0] = generate_query_pattern('A', query, query_size, idx * WORD_SIZE);
        patterns[idx * 4 + 1] = generate_query_pattern('C', query, query_size, idx * WORD_SIZE);
        patterns[idx * 4 + 2] = generate_query_pattern('T', query, query_size, idx * WORD_SIZE);
        patterns[idx * 4 + 3] = generate_query_pattern('G', query, query_size, idx * WORD_SIZE);
    }
    /* workspace as the host sizes it: compute_matrix_size_for_alignment, aligner_global_myers_banded.cpp:47-55 */
    const int32_t pmax     = (max_bandwidth + 1) / 2;
    const int64_t max_elem = (int64_t)ceiling_divide(imin(1 + 2 * pmax, query_size), WORD_SIZE) * ((int64_t)target_size + 1);
    band_t b;
    b.pv    = (WordType*)calloc((size_t)max_elem + 1, sizeof(WordType));
    b.mv    = (WordType*)calloc((size_t)max_elem + 1, sizeof(WordType));
    b.score = (int32_t*)calloc((size_t)max_elem + 1, sizeof(int32_t));
    b.n_rows = 0;
    b.n_cols = target_size + 1;

    int32_t max_distance_estimate = imax(1, iabs(target_size - query_size) + imin(target_size, query_size) / 20);
    int32_t diagonal_begin = -1, diagonal_end = -1, band_width = 0;
    for (;;)
    {
        int32_t p = imin(imin(target_size, query_size), (max_distance_estimate - iabs(target_size - query_size)) / 2);
        int32_t band_width_new = imin(1 + 2 * p + iabs(target_size - query_size), query_size);
        if (band_width_new % WORD_SIZE == 1 && band_width_new != query_size)
        {
            p += 1;
            band_width_new = imin(1 + 2 * p + iabs(target_size - query_size), query_size);
        }
        if (band_width_new > max_bandwidth)
        {
            band_width_new = max_bandwidth;
            p              = (band_width_new - 1 - iabs(target_size - query_size)) / 2;
        }
        const int32_t n_words_band = ceiling_divide(band_width_new, WORD_SIZE);
        if ((int64_t)n_words_band * (int64_t)(target_size + 1) > max_elem)
        {
            band_width = -band_width;
            break;
        }
        band_width = band_width_new;
        b.n_rows   = n_words_band;
        if (band_cells) *band_cells += (int64_t)n_words_band * WORD_SIZE * target_size;
        compute_scores_banded(&diagonal_begin, &diagonal_end, &b, patterns, n_words, target, target_size, query_size, band_width,
                              n_words_band, p);
        const int32_t cur_edit_distance = n_words_band > 0 ? SC(&b, n_words_band - 1, target_size) : target_size;
        if (cur_edit_distance <= max_distance_estimate || band_width == query_size) break;
        if (band_width == max_bandwidth)
        {
            band_width = -band_width;
            break;
        }
        max_distance_estimate *= 2;
    }
    int32_t rc = 0;
    if (band_width != 0)
    {
        *n_runs     = backtrace_banded(ops, counts, &b, diagonal_begin, diagonal_end, iabs(band_width), target_size, query_size);
        *is_optimal = band_width > 0 ? 1 : 0;
    }
    else
        rc = 1;
    free(patterns);
    free(b.pv);
    free(b.mv);
    free(b.score);
    return rc;
}
