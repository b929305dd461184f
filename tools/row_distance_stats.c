/* row_distance_stats.c -- analysis aid (CPU, oracle side): observes the rows of the oracle's banded forward pass and
 * histograms the distance to a row's farthest predecessor and its predecessor count: what fraction of the rows a packed
 * forward pass with an LDS ring of R rows could serve from the ring (bands 384 / 512 with a four-row ring).
 * Built by tools/row_distance_stats.py into tools/bin/; TEST/ANALYSIS INFRASTRUCTURE ONLY. */
#include <stdint.h>
#include <string.h>

extern void (*poa_oracle_row_hook)(int32_t, int32_t, int32_t, int32_t);

static int64_t far_hist[17];  /* farthest predecessor 0 (none) .. 15 rows up, 16 = more */
static int64_t cnt_hist[8];   /* predecessor count 0 .. 6, 7 = more */
static int64_t rows_total, rows_prev_only;

static void hook(int32_t row, int32_t pred_count, int32_t far, int32_t band_start)
{
    (void)row; (void)band_start;
    rows_total++;
    far_hist[far > 16 ? 16 : far]++;
    cnt_hist[pred_count > 7 ? 7 : pred_count]++;
    if (pred_count == 1 && far == 1) rows_prev_only++;
}
void rds_install(void) { poa_oracle_row_hook = hook; }
void rds_get(int64_t* far, int64_t* cnt, int64_t* totals)
{
    memcpy(far, far_hist, sizeof(far_hist));
    memcpy(cnt, cnt_hist, sizeof(cnt_hist));
    totals[0] = rows_total;
    totals[1] = rows_prev_only;
}
