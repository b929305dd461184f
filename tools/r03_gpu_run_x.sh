#!/bin/bash
# round 3, GPU call X: MSA rows by one lane per node: MSA tests (A/B against the per-sequence walk), binding tests, 598-window
# golden, long-read sub-record, kernel timeline tail
set -u
TAG=${1:-r03x}
REPO=$(pwd)
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_poa.py tests/test_gpu_config_goldens.py tests/test_gpu_pygenomeworks_bindings.py -m gpu -q -x 2>&1 | tail -6 ) > $OUT/pytest.log; cat $OUT/pytest.log
for i in 1 2; do
timeout 600 python bench.py --sub-configs long_reads --no-cpu-baseline --steps 3 --warmup 1 > $OUT/bench_long_reads_$i.json 2> $OUT/bench.err
python - $OUT/bench_long_reads_$i.json <<'PY'
import json,sys
d=json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][0]); v=d['sub_records']['configs[3]']
print("bench", v['value'], v['unit'], v['ms'], "ms", "golden", v['windows_equal_to_oracle_golden'], "differ", v['windows_differing_from_golden'])
PY
done
