#!/bin/bash
# round 2: long-read kernels (multi-wave forward, staged traceback): parity on the 598-window golden, phase breakdown, bench
set -u
TAG=${1:-r02d}
mkdir -p gpurun_out/${TAG}
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_config_goldens.py tests/test_gpu_poa.py -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/${TAG}/pytest.log
( timeout 300 python tools/profile_phases.py 1024 2>&1 | tail -1 ) > gpurun_out/${TAG}/metric_phases.json
( timeout 600 python tools/profile_long_read.py 0 64 2>&1 | tail -1 ) > gpurun_out/${TAG}/long_read_phases.json
( timeout 900 python bench.py --sub-configs long_reads --no-cpu-baseline > gpurun_out/${TAG}/bench.json 2> gpurun_out/${TAG}/bench.err ) ; echo "bench rc=$?" >> gpurun_out/${TAG}/bench.err
ls -la gpurun_out/${TAG}
