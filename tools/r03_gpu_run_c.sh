#!/bin/bash
# round 3, GPU call C: row microbenchmark (production forward routine, synthetic row tables, ablations) + store ablation of the kernel
set -u
TAG=${1:-r03c}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 tools/bin/microbench_rows 1024 8 > $OUT/microbench_rows.json 2> $OUT/microbench_rows.err; echo "microbench rc=$?"
cat $OUT/microbench_rows.json | cut -c1-220
timeout 300 python tools/r03_store_ablation.py 1024 > $OUT/store_ablation.json 2> $OUT/store_ablation.err; echo "ablation rc=$?"
cat $OUT/store_ablation.json; tail -3 $OUT/store_ablation.err
