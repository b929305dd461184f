#!/bin/bash
# round 3, GPU call U: full GPU suite and the full driver-format bench line
set -u
TAG=${1:-r03u}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $OUT/pytest.log; tail -3 $OUT/pytest.log
( timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err ); echo "bench rc=$?"; tail -3 $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][0])
print("headline", d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['equals_oracle_golden'])
s=d['sub_records']
print("configs[1]", s['configs[1]']['value'], s['configs[1]']['ms'], s['configs[1]']['kernel_only'])
print("configs[4]", s['configs[4]']['value'], s['configs[4]']['ms'], s['configs[4]']['kernel_only'])
print("default", s['default_aligner']['value'], s['default_aligner']['ms'], s['default_aligner'].get('kernel_only'))
print("long", s['configs[3]']['value'], s['configs[3]']['ms'], s['configs[3]']['windows_equal_to_oracle_golden'])
PY
