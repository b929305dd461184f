#!/bin/bash
# round 2, second GPU call: the new GPU tests (bindings, aligner vectors + hooks, multi-device, spoa_accurate), full suite, bench
set -u
mkdir -p gpurun_out/r02b
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > gpurun_out/r02b/pytest.log
( timeout 900 python bench.py > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err ) ; echo "bench rc=$?" >> gpurun_out/r02b/bench.err
ls -la gpurun_out/r02b
