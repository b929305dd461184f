#!/bin/bash
# round 3, GPU call I: default aligner with one wavefront per pair: parity (oracle + the one-lane kernel), benchmark shapes
set -u
TAG=${1:-r03i}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_aligner.py tests/test_gpu_aligner_vectors.py tests/test_gpu_pygenomeworks_bindings.py -m gpu -q -x 2>&1 | tail -25 ) > $OUT/pytest.log; tail -6 $OUT/pytest.log
timeout 600 python tools/bench_default_aligner.py > $OUT/default_aligner_wave.json 2> $OUT/default_aligner_wave.err; cat $OUT/default_aligner_wave.json
GWHIP_HIRSCHBERG_WAVE=0 timeout 900 python tools/bench_default_aligner.py quick > $OUT/default_aligner_lane.json 2> $OUT/default_aligner_lane.err; cat $OUT/default_aligner_lane.json
