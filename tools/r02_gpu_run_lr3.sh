#!/bin/bash
# long reads after a kernel change: the A/B test of the kernel variants, the 598-window golden, the long-read sub-record twice
set -u
TAG=${1:-r02lr}
mkdir -p gpurun_out/${TAG}
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_poa.py -m gpu -q -x -k "long_read" 2>&1 | tail -5 ) > gpurun_out/${TAG}/pytest.log
( timeout 600 python -m pytest tests/test_gpu_config_goldens.py -m gpu -q -x -k "config4 or long" 2>&1 | tail -5 ) >> gpurun_out/${TAG}/pytest.log
for i in 1 2; do
  ( timeout 600 python bench.py --sub-configs long_reads --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/${TAG}/bench_$i.json 2> gpurun_out/${TAG}/bench_$i.err )
  python - gpurun_out/${TAG}/bench_$i.json >> gpurun_out/${TAG}/runs.txt <<'PY'
import json,sys
d=json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][0]); v=d['sub_records']['configs[3]']
print("bench", v['value'], v['unit'], v['ms'], "ms", "golden", v['windows_equal_to_oracle_golden'], "differ", v['windows_differing_from_golden'])
PY
done
cat gpurun_out/${TAG}/pytest.log gpurun_out/${TAG}/runs.txt
