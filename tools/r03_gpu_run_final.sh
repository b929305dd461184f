#!/bin/bash
# round 3, final GPU call: full GPU suite, PMC fetch / write passes over the sub-records' kernels (-> the traffic file the
# bench line reads), the full driver-format bench line, rocprofv3 --kernel-trace --stats of the bench command
set -u
TAG=${1:-r03final}
REPO=$(pwd)
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $OUT/pytest.log; tail -3 $OUT/pytest.log
SUBS=aligner,default_aligner,long_reads PASSES="fetch write" bash tools/pmc_passes.sh $OUT/pmc_sub > $OUT/pmc_sub.log 2>&1
python tools/pmc_summary.py $OUT/pmc_sub > $OUT/pmc_sub_summary.csv 2>/dev/null
python tools/pmc_sub_traffic.py $OUT/pmc_sub_summary.csv "tools/pmc_passes.sh with SUBS=aligner,default_aligner,long_reads (tools/r03_gpu_run_final.sh)" > $OUT/r03_pmc_traffic_sub.json && cp $OUT/r03_pmc_traffic_sub.json profiles/r03_pmc_traffic_sub.json
rm -rf $OUT/pmc_sub
( timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err ); echo "bench rc=$?"; tail -3 $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][0])
print("headline", d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['equals_oracle_golden'])
s=d['sub_records']
for k in ('configs[1]','configs[4]','default_aligner','configs[3]'):
    print(k, s[k]['value'], s[k]['ms'], s[k].get('kernel_only'), s[k]['roofline'].get('traffic'))
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/stats -- python $REPO/bench.py --no-cpu-baseline > $REPO/$OUT/bench_under_rocprof.json 2> $REPO/$OUT/stats.log)
DB=$(find $OUT/stats -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py "$DB" > $OUT/kernel_stats.csv && head -12 $OUT/kernel_stats.csv | cut -c1-150
rm -rf $OUT/stats
