/* traceback_run_stats.c -- analysis aid (CPU, oracle side): observes the banded traceback of the oracle and histograms
 * what a wave-parallel "run skipping" walk could take at once: runs of consecutive steps that each move one row up and
 * one column left (di == 1 && dj == 1), the steps that break a run, and the row distance of the moves.
 * Built by tools/traceback_run_stats.py into tools/bin/; TEST/ANALYSIS INFRASTRUCTURE ONLY. */
#include <stdint.h>
#include <string.h>

extern void (*poa_oracle_step_hook)(int32_t, int32_t, int32_t, int32_t);

#define MAXRUN 129
static int64_t run_hist[2][MAXRUN]; /* [early/late read][run length], run = consecutive (1,1) steps */
static int64_t kind[2][4];          /* (1,1) | other diagonal | vertical | horizontal */
static int64_t dist_hist[2][9];     /* row distance of a move (0..7, 8 = more) */
static int64_t steps[2], walks[2];
static int32_t cur_run, last_pi = -1, last_pj = -1, walk_no, late_from = 16;

static void flush_run(int late)
{
    if (cur_run > 0) run_hist[late][cur_run < MAXRUN ? cur_run : MAXRUN - 1]++;
    cur_run = 0;
}
static void hook(int32_t i, int32_t j, int32_t pi, int32_t pj)
{
    if (!(i == last_pi && j == last_pj)) /* first step of a new walk */
    {
        flush_run(walk_no >= late_from);
        walk_no++;
        walks[walk_no >= late_from]++;
    }
    const int late = walk_no >= late_from;
    const int di = i - pi, dj = j - pj;
    steps[late]++;
    dist_hist[late][di > 8 ? 8 : di]++;
    if (di == 1 && dj == 1) { kind[late][0]++; cur_run++; }
    else
    {
        /* a run of n (1,1) steps followed by one other step = one iteration of the run-skipping walk */
        run_hist[late][cur_run < MAXRUN ? cur_run : MAXRUN - 1]++;
        cur_run = 0;
        kind[late][dj == 1 && di > 0 ? 1 : (dj == 0 ? 2 : 3)]++;
    }
    last_pi = pi; last_pj = pj;
}
void trs_install(int32_t late_from_read) { poa_oracle_step_hook = hook; late_from = late_from_read; }
void trs_new_window(void) { flush_run(walk_no >= late_from); walk_no = 0; last_pi = last_pj = -1; }
void trs_get(int64_t* out_run, int64_t* out_kind, int64_t* out_dist, int64_t* out_steps)
{
    memcpy(out_run, run_hist, sizeof(run_hist));
    memcpy(out_kind, kind, sizeof(kind));
    memcpy(out_dist, dist_hist, sizeof(dist_hist));
    out_steps[0] = steps[0]; out_steps[1] = steps[1]; out_steps[2] = walks[0]; out_steps[3] = walks[1];
}
