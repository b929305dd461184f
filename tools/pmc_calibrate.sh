#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of three microkernels of known byte counts (tools/pmc_calibration.hip): the factors that turn the
# counters of the POA kernels into bytes. usage: tools/pmc_calibrate.sh <out_dir> ; run from the repo root on the GPU box
set -u
OUT=${1:-gpurun_out/pmc_cal}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$REPO/$OUT/$c" -o cal -- $REPO/tools/bin/pmc_calibration > "$REPO/$OUT/$c.json" 2> "$REPO/$OUT/$c.log"; echo "$c rc=$?"
done
cd "$REPO"
python tools/pmc_summary.py "$OUT" > "$OUT/summary.csv"; cat "$OUT/summary.csv"; cat "$OUT/FETCH_SIZE.json"
