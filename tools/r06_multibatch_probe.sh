#!/bin/bash
# Where the reference's multi-batch pattern spends its time: tools/fill_probe over the config-3 windows in
# the benchmark's BatchConfig(1024, 200), host phases on the wall clock and the kernels under rocprofv3 --kernel-trace.
#   gpurun -- 'bash tools/r06_multibatch_probe.sh r06b'
set -u
TAG=${1:-probe}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python tools/dump_config3_windows.py /tmp/config3_windows.txt 1024
for W in 512 1024 2048; do
    tools/bin/fill_probe $W 32 /tmp/config3_windows.txt > $OUT/probe_$W.json
    cat $OUT/probe_$W.json
    (cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_$W -- $REPO/tools/bin/fill_probe $W 32 /tmp/config3_windows.txt > /dev/null 2> $OUT/prof_$W.log)
    DB=$(find $OUT/prof_$W -name "*.db" | head -1)
    [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" > $OUT/kernels_$W.csv && head -6 $OUT/kernels_$W.csv | cut -c1-220
    rm -rf $OUT/prof_$W
done
