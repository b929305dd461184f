#!/bin/bash
# round 2, GPU call: the new GPU tests (bindings, aligner vectors + hooks, multi-device, spoa_accurate), full suite, bench
set -u
TAG=${1:-r02c}
mkdir -p gpurun_out/${TAG}
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > gpurun_out/${TAG}/pytest.log
( timeout 900 python bench.py > gpurun_out/${TAG}/bench.json 2> gpurun_out/${TAG}/bench.err ) ; echo "bench rc=$?" >> gpurun_out/${TAG}/bench.err
ls -la gpurun_out/${TAG}
