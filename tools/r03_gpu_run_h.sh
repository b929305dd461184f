#!/bin/bash
# round 3, GPU call H: aligner -- group kernel with register pattern windows: parity tests, bench of configs[1] / [4]
set -u
TAG=${1:-r03h}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_aligner.py tests/test_gpu_aligner_vectors.py tests/test_gpu_config_goldens.py -m gpu -q 2>&1 | tail -15 ) > $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 300 python tools/bench_aligner.py 1000000 > $OUT/aligner_bench.json 2> $OUT/aligner_bench.err; cat $OUT/aligner_bench.json | cut -c1-500
