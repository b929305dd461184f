#!/bin/bash
# round 3, GPU call P: long reads on move bytes (sheared tile + run skipping in the traceback): parity, phases, sub-record
set -u
TAG=${1:-r03p}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_poa.py tests/test_gpu_poa_hooks.py -m gpu -q -x 2>&1 | tail -5 ) > $OUT/pytest.log
cat $OUT/pytest.log
echo "window 389: $(timeout 300 python tools/profile_long_read.py 389 1 2>&1 | tail -1)" > $OUT/window_phases.txt
cat $OUT/window_phases.txt
( timeout 900 python -m pytest tests/test_gpu_config_goldens.py -m gpu -q -x 2>&1 | tail -3 ) > $OUT/pytest_golden.log; cat $OUT/pytest_golden.log
timeout 600 python tools/profile_long_read_windows.py 30486 6 > $OUT/class0_windows.json 2> $OUT/class0.err; tail -2 $OUT/class0.err
python - $OUT/class0_windows.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print({k:v for k,v in d.items() if k!='slowest'})
for r in d['slowest']: print(r['window'], r['ticks_M'], r['share'])
PY
timeout 600 python bench.py --sub-configs long_reads --no-cpu-baseline --steps 3 --warmup 1 > $OUT/bench_long_reads.json 2> $OUT/bench.err
python - $OUT/bench_long_reads.json <<'PY'
import json,sys
d=json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][0]); v=d['sub_records']['configs[3]']
print("headline", d['value'], d['roofline']['kernel_ms'], d['equals_oracle_golden'])
print("bench", v['value'], v['unit'], v['ms'], "ms", "golden", v['windows_equal_to_oracle_golden'], "differ", v['windows_differing_from_golden'])
PY
