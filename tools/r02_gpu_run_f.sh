#!/bin/bash
# round 2: aligner kernels after the target-stream change: parity, phase ablation, bench sub-records
set -u
TAG=${1:-r02f}
mkdir -p gpurun_out/${TAG}
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_aligner.py tests/test_gpu_aligner_vectors.py tests/test_gpu_config_goldens.py tests/test_gpu_multi_device.py tests/test_overlap_alignment.py -m gpu -q 2>&1 | tail -15 ) > gpurun_out/${TAG}/pytest.log
for s in 0 1 3; do
  ( GWHIP_MYERS_SKIP=$s timeout 300 python tools/bench_aligner.py 200000 2>&1 | grep config | sed "s/^/skip=$s /" ) >> gpurun_out/${TAG}/aligner_ablation.txt
done
( timeout 900 python bench.py --sub-configs aligner > gpurun_out/${TAG}/bench.json 2> gpurun_out/${TAG}/bench.err ) ; echo "bench rc=$?" >> gpurun_out/${TAG}/bench.err
ls -la gpurun_out/${TAG}
