#!/bin/bash
# rocprofv3 --kernel-trace --stats of the driver's bench command at the final commit (per-kernel durations for profiles/)
set -u
TAG=${1:-r02stats}
REPO=$(pwd)
OUT=gpurun_out/$TAG
mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 420 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/stats -- python $REPO/bench.py --no-cpu-baseline > $REPO/$OUT/bench.json 2> $REPO/$OUT/stats.log)
DB=$(find $OUT/stats -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py "$DB" > $OUT/kernel_stats.csv && head -14 $OUT/kernel_stats.csv
find $OUT -name "*.db" -delete
