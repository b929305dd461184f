/*
 * hirschberg_oracle.c -- CPU restatement of the reference's default global aligner, Hirschberg's divide and
 * conquer with Myers' bit-vector edit distance (cudaaligner/src/hirschberg_myers_gpu.cu:110-181, 278-644 and the
 * host constants of aligner_global_hirschberg_myers.cpp:32-33,57).
 * TEST INFRASTRUCTURE ONLY: used by tests/ and __graft_entry__.smoke(); never by the product.
 *
 * Value-level: the bit-vector machinery of the reference only ever produces unit-cost edit distances, so this file
 * computes them with the plain recurrence and keeps everything that decides the OUTPUT:
 *   - the explicit LIFO stack of (query, target) ranges, capacity 64, left half pushed before the right half
 *     (:575-641), so the path comes out back to front; a failed push empties the result;
 *   - the leaves: empty target -> deletions, empty query -> insertions (:598-605), one query character -> the
 *     right-to-left scan of :483-515, short queries (< 63 characters, matrix fits) -> full matrix + the backtrace
 *     of :124-181 (insertion before deletion before diagonal);
 *   - the split: query midpoint len/2, target midpoint = argmin_t F[t] + R[T - t] (:412-481) found the way the
 *     32 lanes of the reference's warp find it (strided first-minimum per lane, then a strict-less shuffle-down
 *     tree), which fixes the choice among equal sums.
 * PINNING: the restatement is pinned by the reference's known-answer CIGARs (cudaaligner/tests/Test_AlignerGlobal.cpp:73-155,
 * tests/golden), by optimality of the edit distance against the reference's CPU NW (oracle/_ref), and -- tie-breaking
 * included -- by the reference's own hirschberg_myers_gpu.cu run on the CPU (oracle/simt, oracle/_ref/libref_cudaaligner_simt.so):
 * tests/golden/reference_simt_alignments.json.gz (pairs up to 5 kbp) and fresh random pairs, tests/test_reference_simt.py.
 */
#include "hirschberg_oracle.h"

#include <stdlib.h>
#include <string.h>

enum { kStackSize = 64, kSwitchToMyers = 63, kWord = 32, kWarp = 32 };

static int32_t ceil_div(int32_t a, int32_t b) { return (a + b - 1) / b; }

/* last row of the edit-distance matrix: out[t] = ED(q[0..qn), target prefix of length t), t = 0..tn;
   reverse != 0 reads both sequences back to front (myers_compute_scores with full_score_matrix == false, :278-381) */
static void last_row(const char* q, int32_t qn, const char* tg, int32_t tn, int reverse, int32_t* out, int32_t* tmp)
{
    int32_t* prev = out;
    int32_t* cur  = tmp;
    for (int32_t t = 0; t <= tn; t++) prev[t] = t;
    for (int32_t i = 1; i <= qn; i++)
    {
        const char qc = reverse ? q[qn - i] : q[i - 1];
        cur[0]        = i;
        for (int32_t t = 1; t <= tn; t++)
        {
            const char tc = reverse ? tg[tn - t] : tg[t - 1];
            int32_t v     = prev[t - 1] + (qc != tc);
            if (prev[t] + 1 < v) v = prev[t] + 1;
            if (cur[t - 1] + 1 < v) v = cur[t - 1] + 1;
            cur[t] = v;
        }
        int32_t* s = prev; prev = cur; cur = s;
    }
    if (prev != out) memcpy(out, prev, sizeof(int32_t) * (size_t)(tn + 1));
}

/* hirschberg_myers_compute_target_mid_warp (:412-481): the argmin as the warp computes it */
static int32_t target_mid(const int32_t* fwd, const int32_t* rev, int32_t tn)
{
    int32_t cur_min[kWarp], midpoint[kWarp];
    for (int lane = 0; lane < kWarp; lane++)
    {
        cur_min[lane]  = INT32_MAX;
        midpoint[lane] = 0;
        for (int32_t t = lane; t <= tn; t += kWarp)
        {
            const int32_t sum = fwd[t] + rev[tn - t];
            if (sum < cur_min[lane]) { cur_min[lane] = sum; midpoint[lane] = t; }
        }
    }
    for (int i = 16; i > 0; i >>= 1)
    {
        /* all lanes read their partner's value of the previous step, then update (shfl_down: lanes without a partner
           read themselves and never win the strict comparison) */
        int32_t nm[kWarp], np[kWarp];
        for (int lane = 0; lane < kWarp; lane++)
        {
            const int src = lane + i < kWarp ? lane + i : lane;
            nm[lane]      = cur_min[src];
            np[lane]      = midpoint[src];
        }
        for (int lane = 0; lane < kWarp; lane++)
            if (nm[lane] < cur_min[lane]) { cur_min[lane] = nm[lane]; midpoint[lane] = np[lane]; }
    }
    return midpoint[0];
}

/* full matrix + append_myers_backtrace (:124-181); returns the number of states appended */
static int32_t full_backtrace(const char* q, int32_t qn, const char* tg, int32_t tn, int8_t* path)
{
    const int32_t w = tn + 1;
    int32_t* d      = (int32_t*)malloc(sizeof(int32_t) * (size_t)(qn + 1) * (size_t)w);
    for (int32_t t = 0; t <= tn; t++) d[t] = t;
    for (int32_t i = 1; i <= qn; i++)
    {
        d[(size_t)i * w] = i;
        for (int32_t t = 1; t <= tn; t++)
        {
            int32_t v = d[(size_t)(i - 1) * w + t - 1] + (q[i - 1] != tg[t - 1]);
            if (d[(size_t)(i - 1) * w + t] + 1 < v) v = d[(size_t)(i - 1) * w + t] + 1;
            if (d[(size_t)i * w + t - 1] + 1 < v) v = d[(size_t)i * w + t - 1] + 1;
            d[(size_t)i * w + t] = v;
        }
    }
    int32_t i = qn, j = tn, pos = 0;
    int32_t myscore = d[(size_t)i * w + j];
    while (i > 0 && j > 0)
    {
        const int32_t above = d[(size_t)(i - 1) * w + j];
        const int32_t diag  = d[(size_t)(i - 1) * w + j - 1];
        const int32_t left  = d[(size_t)i * w + j - 1];
        int8_t r;
        if (left + 1 == myscore) { r = HO_INSERTION; myscore = left; --j; }
        else if (above + 1 == myscore) { r = HO_DELETION; myscore = above; --i; }
        else { r = diag == myscore ? HO_MATCH : HO_MISMATCH; myscore = diag; --i; --j; }
        path[pos++] = r;
    }
    while (i > 0) { path[pos++] = HO_DELETION; --i; }
    while (j > 0) { path[pos++] = HO_INSERTION; --j; }
    free(d);
    return pos;
}

/* hirschberg_myers_single_char_warp (:483-515) */
static int32_t single_char(char qc, const char* tb, const char* te, int8_t* path)
{
    int8_t* p     = path;
    const char* t = te - 1;
    while (t >= tb)
    {
        if (*t == qc) { *p++ = HO_MATCH; --t; break; }
        *p++ = HO_INSERTION;
        --t;
    }
    if (*(p - 1) != HO_MATCH) *(p - 1) = HO_MISMATCH;
    while (t >= tb) { *p++ = HO_INSERTION; --t; }
    return (int32_t)(te - tb);
}

typedef struct { const char *qb, *qe, *tb, *te; } range_t;

/* `query` / `target` are what the bit-vector parts see of the pair (tests/oracle_aligner.py, pattern_view(): the four query
   patterns and the target's pattern index; the identity over ACGT); the single-character leaf compares the characters themselves
   (:483-515), so it gets the caller's own sequences. */
int32_t hirschberg_oracle_align_raw(const char* query, int32_t query_size, const char* target, int32_t target_size,
                                    const char* raw_query, const char* raw_target, int32_t max_query_length, int8_t* path,
                                    int32_t* path_length)
{
    /* per-alignment matrix capacity of the reference's workspace: max_n_words * (switch size + 1) elements (:40-41) */
    const int64_t max_elems = (int64_t)ceil_div(max_query_length, kWord) * (kSwitchToMyers + 1);
    range_t stack[kStackSize];
    int sp      = 0;
    stack[sp++] = (range_t){query, query + query_size, target, target + target_size};
    int32_t* fwd = (int32_t*)malloc(sizeof(int32_t) * (size_t)(target_size + 1) * 3);
    int32_t* rev = fwd + (target_size + 1);
    int32_t* tmp = rev + (target_size + 1);
    int ok       = 1;
    int32_t len  = 0;
    while (ok && sp > 0)
    {
        const range_t e = stack[--sp];
        const int32_t qn = (int32_t)(e.qe - e.qb), tn = (int32_t)(e.te - e.tb);
        if (tn == 0)
        {
            memset(path + len, HO_DELETION, (size_t)qn);
            len += qn;
        }
        else if (qn == 0)
        {
            memset(path + len, HO_INSERTION, (size_t)tn);
            len += tn;
        }
        else if (qn == 1)
            len += single_char(raw_query[e.qb - query], raw_target + (e.tb - target), raw_target + (e.te - target), path + len);
        else
        {
            if (qn < kSwitchToMyers)
            {
                const int32_t n_words = ceil_div(qn, kWord);
                if ((int64_t)(tn + 1) * n_words <= max_elems)
                {
                    len += full_backtrace(e.qb, qn, e.tb, tn, path + len);
                    continue;
                }
            }
            const char* qmid = e.qb + qn / 2;
            last_row(e.qb, (int32_t)(qmid - e.qb), e.tb, tn, 0, fwd, tmp);
            last_row(qmid, (int32_t)(e.qe - qmid), e.tb, tn, 1, rev, tmp);
            const char* tmid = e.tb + target_mid(fwd, rev, tn);
            if (sp < kStackSize) stack[sp++] = (range_t){e.qb, qmid, e.tb, tmid}; else ok = 0;
            if (ok && sp < kStackSize) stack[sp++] = (range_t){qmid, e.qe, tmid, e.te}; else ok = 0;
        }
    }
    free(fwd);
    if (!ok) len = 0;
    *path_length = len;
    return ok ? 0 : 1;
}

int32_t hirschberg_oracle_align(const char* query, int32_t query_size, const char* target, int32_t target_size,
                                int32_t max_query_length, int8_t* path, int32_t* path_length)
{
    return hirschberg_oracle_align_raw(query, query_size, target, target_size, query, target, max_query_length, path, path_length);
}
