#!/bin/bash
# round 3, GPU call Q: long reads after a forward-pass change: A/B test, golden, per-window phases, sub-record twice
set -u
TAG=${1:-r03q}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_poa.py -m gpu -q -x -k "long_read" 2>&1 | tail -3 ) > $OUT/pytest.log
( timeout 900 python -m pytest tests/test_gpu_config_goldens.py -m gpu -q -x -k "config4 or long" 2>&1 | tail -3 ) >> $OUT/pytest.log; cat $OUT/pytest.log
timeout 600 python tools/profile_long_read_windows.py 30486 4 > $OUT/class0_windows.json 2> $OUT/class0.err; tail -2 $OUT/class0.err
python - $OUT/class0_windows.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print({k:v for k,v in d.items() if k!='slowest'})
for r in d['slowest']: print(r['window'], r['ticks_M'], r['share'])
PY
for i in 1 2; do
timeout 600 python bench.py --sub-configs long_reads --no-cpu-baseline --steps 3 --warmup 1 > $OUT/bench_long_reads_$i.json 2> $OUT/bench.err
python - $OUT/bench_long_reads_$i.json <<'PY'
import json,sys
d=json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][0]); v=d['sub_records']['configs[3]']
print("headline", d['value'], d['roofline']['kernel_ms'], d['equals_oracle_golden'])
print("bench", v['value'], v['unit'], v['ms'], "ms", "golden", v['windows_equal_to_oracle_golden'], "differ", v['windows_differing_from_golden'])
PY
done
