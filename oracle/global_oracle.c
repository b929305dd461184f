/*
 * global_oracle.c -- CPU restatement of AlignerGlobalUkkonen and AlignerGlobalMyers. TEST INFRASTRUCTURE ONLY:
 * used by tests/ and __graft_entry__.smoke() as the checker; never by the product path.
 *
 * Pinning: tests/test_oracle_aligner.py checks both functions against the reference's own golden CIGARs
 * (cudaaligner/tests/Test_AlignerGlobal.cpp:79-153: the table is run for the Ukkonen and Myers classes too; the
 * empty-sequence cases only for Myers) and, when oracle/_ref is built, against the reference's own CPU code
 * compiled in place: ukkonen_cpu() (ukkonen_cpu.cpp, the function the reference's own tests compare the GPU
 * Ukkonen path with) and the naive NW edit distance for optimality of both; and against the reference's own ukkonen_gpu.cu /
 * myers_gpu.cu run on the CPU (oracle/simt): tests/golden/reference_simt_alignments.json.gz, tests/test_reference_simt.py.
 *
 * Ukkonen: the reference stores the band in anti-diagonal coordinates, slot (k, l) = ((j - i + p) / 2, i + j) with C
 * integer division, and its backtrace indexes that storage directly -- including the slot aliasing of the
 * diagonal just left of the band (j - i + p == -1 truncates to k == 0). The restatement therefore keeps the same
 * storage and index arithmetic instead of an (i, j) matrix.
 */
#include "global_oracle.h"

#include <stdlib.h>
#include <string.h>

#define GO_MAX ((int16_t)(INT16_MAX - 1)) /* numeric_limits<nw_score_t>::max() - 1, ukkonen_gpu.cu:81,146 */

static int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
static int32_t iabs(int32_t a) { return a < 0 ? -a : a; }
static int16_t min3(int16_t a, int16_t b, int16_t c) { return a < b ? (a < c ? a : c) : (b < c ? b : c); }

/* ukkonen_gpu.cu:49-61 */
static void to_band_indices(int32_t i, int32_t j, int32_t p, int32_t* k, int32_t* l)
{
    *k = (j - i + p) / 2; /* truncates towards zero */
    *l = j + i;
}

int32_t ukkonen_oracle_align(const char* query, int32_t query_size, const char* target, int32_t target_size, int32_t p,
                             int8_t* path, int32_t* path_length)
{
    /* ukkonen_compute_score_matrix, ukkonen_gpu.cu:214-258: the shorter sequence runs along i */
    int32_t m         = query_size + 1;
    int32_t n         = target_size + 1;
    const char* a     = query;
    const char* b     = target;
    int8_t insertion  = GO_INSERTION;
    int8_t deletion   = GO_DELETION;
    if (m > n)
    {
        int32_t t = n; n = m; m = t;
        const char* s = a; a = b; b = s;
        insertion = GO_DELETION; /* ukkonen_backtrace_kernel :87-91 */
        deletion  = GO_INSERTION;
    }
    const int32_t bw        = (1 + n - m + 2 * p + 1) / 2;
    const int32_t kmax_odd  = (n - m + 2 * p - 1) / 2 + 1;
    const int32_t kmax_even = (n - m + 2 * p) / 2 + 1;
    const int32_t rows = bw, cols = n + m;
    int16_t* S = (int16_t*)malloc((size_t)rows * (size_t)cols * sizeof(int16_t));
    if (!S) return -1;
#define SC(k, l) S[(size_t)(k) * (size_t)cols + (size_t)(l)]

    /* ukkonen_init_score_matrix :189-208 */
    for (int32_t k = 0; k < rows; k++)
        for (int32_t l = 0; l < cols; l++)
        {
            const int32_t j = k - (p + l) / 2 + l; /* to_matrix_indices :42-47 */
            const int32_t i = l - j;
            int16_t v       = GO_MAX;
            if (i == 0) v = (int16_t)j;
            else if (j == 0) v = (int16_t)i;
            SC(k, l) = v;
        }

    /* anti-diagonals in increasing l; the diagonal parity handled at l is (p + l) % 2 (:236-257). The reference
       runs l far past the matrix; the lmax test makes those steps no-ops, so stopping at cols is equivalent. */
    for (int32_t l = 0; l < cols; l++)
    {
        if ((p + l) % 2 == 0)
        {
            for (int32_t k = 0; k < kmax_even; k++) /* ukkonen_compute_score_matrix_even :167-187 */
            {
                const int32_t lmin = iabs(2 * k - p);
                const int32_t lmax = 2 * k <= p ? 2 * (m - p + 2 * k) + lmin : (2 * imin(m, n - 2 * k + p) + lmin);
                if (lmin + 1 <= l && l < lmax)
                {
                    const int32_t j     = k - (p + l) / 2 + l;
                    const int32_t i     = l - j;
                    const int16_t left  = (k - 1 < 0 || l - 1 < 0) ? GO_MAX : (int16_t)(SC(k - 1, l - 1) + 1);
                    const int16_t diag  = l - 2 < 0 ? GO_MAX : (int16_t)(SC(k, l - 2) + (a[i - 1] == b[j - 1] ? 0 : 1));
                    const int16_t above = l - 1 < 0 ? GO_MAX : (int16_t)(SC(k, l - 1) + 1);
                    SC(k, l)            = min3(left, diag, above);
                }
            }
        }
        else
        {
            for (int32_t k = 0; k < kmax_odd; k++) /* ukkonen_compute_score_matrix_odd :145-165 */
            {
                const int32_t lmin = iabs(2 * k + 1 - p);
                const int32_t lmax = 2 * k + 1 <= p ? 2 * (m - p + 2 * k + 1) + lmin : (2 * imin(m, n - (2 * k + 1) + p) + lmin);
                if (lmin + 1 <= l && l < lmax)
                {
                    const int32_t j     = k - (p + l) / 2 + l;
                    const int32_t i     = l - j;
                    const int16_t diag  = l - 2 < 0 ? GO_MAX : (int16_t)(SC(k, l - 2) + (a[i - 1] == b[j - 1] ? 0 : 1));
                    const int16_t left  = l - 1 < 0 ? GO_MAX : (int16_t)(SC(k, l - 1) + 1);
                    const int16_t above = (l - 1 < 0 || k + 1 >= rows) ? GO_MAX : (int16_t)(SC(k + 1, l - 1) + 1);
                    SC(k, l)            = min3(diag, left, above);
                }
            }
        }
    }

    /* ukkonen_backtrace_kernel :66-143 */
    int32_t i = m - 1, j = n - 1, k, l;
    to_band_indices(i, j, p, &k, &l);
    int16_t myscore = SC(k, l);
    int32_t pos     = 0;
#define FETCH(ii, jj, out)                                                       \
    do {                                                                         \
        to_band_indices((ii), (jj), p, &k, &l);                                  \
        (out) = (k < 0 || k >= rows || l < 0 || l >= cols) ? GO_MAX : SC(k, l);  \
    } while (0)
    while (i > 0 && j > 0)
    {
        int16_t above, diag, left;
        int8_t r;
        FETCH(i - 1, j, above);
        FETCH(i - 1, j - 1, diag);
        FETCH(i, j - 1, left);
        if (left + 1 == myscore)
        {
            r       = insertion;
            myscore = left;
            --j;
        }
        else if (above + 1 == myscore)
        {
            r       = deletion;
            myscore = above;
            --i;
        }
        else
        {
            r       = diag == myscore ? GO_MATCH : GO_MISMATCH;
            myscore = diag;
            --i;
            --j;
        }
        path[pos++] = r;
    }
    while (i > 0) { path[pos++] = deletion; --i; }
    while (j > 0) { path[pos++] = insertion; --j; }
    *path_length = pos;
#undef FETCH
#undef SC
    free(S);
    return 0;
}

int32_t myers_full_oracle_align(const char* query, int32_t query_size, const char* target, int32_t target_size,
                                int8_t* path, int32_t* path_length)
{
    /* The bit-vector matrices of myers_compute_score_matrix_kernel (myers_gpu.cu:326-390) encode the exact unit-cost
       edit-distance matrix D (get_myers_score :226-238 decodes D(i, j) for i >= 1; row 0 is implicit, D(0, j) = j),
       so the value-level restatement keeps D itself. */
    const int32_t rows = query_size + 1, cols = target_size + 1;
    int32_t* D = (int32_t*)malloc((size_t)rows * (size_t)cols * sizeof(int32_t));
    if (!D) return -1;
#define DM(i, j) D[(size_t)(i) * (size_t)cols + (size_t)(j)]
    for (int32_t j = 0; j < cols; j++) DM(0, j) = j;
    for (int32_t i = 1; i < rows; i++)
    {
        DM(i, 0) = i;
        for (int32_t j = 1; j < cols; j++)
        {
            const int32_t d = DM(i - 1, j - 1) + (query[i - 1] == target[j - 1] ? 0 : 1);
            const int32_t u = DM(i - 1, j) + 1;
            const int32_t h = DM(i, j - 1) + 1;
            DM(i, j)        = d < u ? (d < h ? d : h) : (u < h ? u : h);
        }
    }
    /* myers_backtrace :240-315 */
    int32_t i = query_size, j = target_size;
    int32_t myscore = i > 0 ? DM(i, j) : 0;
    int32_t pos     = 0;
    while (i > 0 && j > 0)
    {
        const int32_t above = DM(i - 1, j), diag = DM(i - 1, j - 1), left = DM(i, j - 1);
        int8_t r;
        if (left + 1 == myscore)
        {
            r       = GO_INSERTION;
            myscore = left;
            --j;
        }
        else if (above + 1 == myscore)
        {
            r       = GO_DELETION;
            myscore = above;
            --i;
        }
        else
        {
            r       = diag == myscore ? GO_MATCH : GO_MISMATCH;
            myscore = diag;
            --i;
            --j;
        }
        path[pos++] = r;
    }
    while (i > 0) { path[pos++] = GO_DELETION; --i; }
    while (j > 0) { path[pos++] = GO_INSERTION; --j; }
    *path_length = pos;
#undef DM
    free(D);
    return 0;
}
