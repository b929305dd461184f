#!/bin/bash
# profiling counters of the pipelined long-read forward pass and the code-lookup traceback (GWHIP_DEBUG selectors)
set -u
TAG=${1:-r02skp}
FIRST=${2:-0}
COUNT=${3:-64}
mkdir -p gpurun_out/${TAG}
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}/counters.txt
: > $OUT
for sel in ${SKSELS:-0 1 2 3 4 5 6 7 8 9 10 11 12 13}; do
  echo "sksel $sel: $(GWHIP_DEBUG=$((sel << 12)) timeout 300 python tools/profile_long_read.py $FIRST $COUNT 2>&1 | tail -1)" >> $OUT
done
for sel in ${PSELS:-2 3 4 5 6}; do
  echo "psel $sel: $(GWHIP_DEBUG=$((sel << 22)) timeout 300 python tools/profile_long_read.py $FIRST $COUNT 2>&1 | tail -1)" >> $OUT
done
cat $OUT | cut -c1-400
