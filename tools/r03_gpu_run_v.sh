#!/bin/bash
# round 3, GPU call V: SQ instruction / wait counters of the sub-records' kernels (long-read pipeline, level-by-level Hirschberg,
# six-lane group kernel): what bounds them
set -u
TAG=${1:-r03v}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
SUBS=aligner,default_aligner,long_reads PASSES="insts waits vmem" bash tools/pmc_passes.sh $OUT/pmc > $OUT/pmc.log 2>&1
python tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.csv 2>/dev/null
rm -rf $OUT/pmc
grep -E "poa_window_kernel<int, int|hirschberg_levels|group_kernel" $OUT/pmc_summary.csv | cut -c1-150
