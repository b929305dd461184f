#!/bin/bash
# MSA kernel with the wave-wide racon order: parity (MSA tests, goldens incl. the 598 long-read windows, bindings), A/B against
# the serial routine, long-read sub-record
set -u
TAG=${1:-r02msa}
mkdir -p gpurun_out/${TAG}
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_poa.py tests/test_gpu_config_goldens.py tests/test_gpu_pygenomeworks_bindings.py tests/test_gpu_multi_device.py -m gpu -q -x 2>&1 | tail -5 ) > gpurun_out/${TAG}/pytest.log
for mode in 1 0 1 0; do
  ( GWHIP_MSA_SERIAL=$mode timeout 600 python bench.py --sub-configs long_reads --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/${TAG}/bench_$mode.json 2> /dev/null )
  python - gpurun_out/${TAG}/bench_$mode.json $mode >> gpurun_out/${TAG}/runs.txt <<'PY'
import json,sys
d=json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][0]); v=d['sub_records']['configs[3]']
print("msa_serial=%s" % sys.argv[2], v['value'], v['unit'], v['ms'], "ms", "golden", v['windows_equal_to_oracle_golden'], "differ", v['windows_differing_from_golden'])
PY
done
cat gpurun_out/${TAG}/pytest.log gpurun_out/${TAG}/runs.txt
