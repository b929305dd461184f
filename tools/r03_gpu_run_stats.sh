#!/bin/bash
# rocprofv3 --kernel-trace --stats of the headline alone (bench.py --sub-configs none): the metric kernel's average must agree
# with the bench line's roofline.kernel_ms (the full line also launches that instantiation for other shapes and band widths)
set -u
TAG=${1:-r03stats}
REPO=$(pwd)
OUT=gpurun_out/$TAG
mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 420 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/stats -- python $REPO/bench.py --no-cpu-baseline --sub-configs none --steps 20 --warmup 2 > $REPO/$OUT/bench.json 2> $REPO/$OUT/stats.log)
DB=$(find $OUT/stats -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py "$DB" > $OUT/kernel_stats.csv && head -6 $OUT/kernel_stats.csv | cut -c1-160
rm -rf $OUT/stats
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][0])
print("headline", d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['equals_oracle_golden'])
PY
