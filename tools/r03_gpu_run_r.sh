#!/bin/bash
# round 3, GPU call R: default aligner level by level: parity against the depth-first kernel and the oracle, benchmark shapes
set -u
TAG=${1:-r03r}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_aligner.py tests/test_gpu_aligner_vectors.py -m gpu -q -x 2>&1 | tail -15 ) > $OUT/pytest.log; cat $OUT/pytest.log
timeout 300 python tools/bench_default_aligner.py quick > $OUT/default_aligner_levels.json 2> $OUT/da1.err; cat $OUT/default_aligner_levels.json | head -c 1500; echo
GWHIP_HIRSCHBERG_LEVELS=0 timeout 300 python tools/bench_default_aligner.py quick > $OUT/default_aligner_depth_first.json 2> $OUT/da0.err; cat $OUT/default_aligner_depth_first.json | head -c 1500; echo
