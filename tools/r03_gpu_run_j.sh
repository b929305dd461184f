#!/bin/bash
# round 3, GPU call J: full GPU suite, full bench line with every sub-record, PMC fetch / write passes over the sub-records' kernels
set -u
TAG=${1:-r03j}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $OUT/pytest.log; tail -3 $OUT/pytest.log
( timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err ); echo "bench rc=$?"; tail -3 $OUT/bench.err
cut -c1-600 $OUT/bench.json
SUBS=aligner,default_aligner,long_reads PASSES="fetch write" bash tools/pmc_passes.sh $OUT/pmc_sub > $OUT/pmc_sub.log 2>&1
python tools/pmc_summary.py $OUT/pmc_sub > $OUT/pmc_sub_summary.csv 2>/dev/null; cut -c1-160 $OUT/pmc_sub_summary.csv
