#!/bin/bash
# round 3, last GPU call: full GPU suite, full driver-format bench line, rocprofv3 kernel stats of the headline alone
set -u
TAG=${1:-r03final2}
REPO=$(pwd)
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
for k in ('configs[1]','configs[4]','default_aligner','configs[3]'):
    print(k, s[k]['value'], s[k]['ms'], s[k].get('kernel_only'), s[k]['roofline'].get('traffic'))
PY
bash tools/r03_gpu_run_stats.sh ${TAG}_stats
