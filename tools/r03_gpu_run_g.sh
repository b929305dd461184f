#!/bin/bash
# round 3, GPU call G: full GPU suite after the cleanup (old packed pass and code-table walk retired, row-kind selectors, wrap
# detector, advisor fixes), PMC calibration, FETCH / WRITE / instruction counters of the metric kernel, kernel stats, bench line
set -u
TAG=${1:-r03g}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/pytest.log
for f in tests/test_gpu_config_goldens.py tests/test_gpu_poa.py tests/test_gpu_poa_hooks.py $(ls tests/test_*.py | grep -v "test_gpu_config_goldens\|test_gpu_poa.py\|test_gpu_poa_hooks"); do
  echo "== $f" >> $OUT/pytest.log
  ( timeout 900 python -m pytest $f -m gpu -q 2>&1 | tail -${PYTAIL:-25} ) >> $OUT/pytest.log
done
grep -E "^== |passed|failed|error|Aborted|fault" $OUT/pytest.log | grep -B1 -E "passed|failed|error|Aborted|fault" | grep -v "^--" | tail -40
bash tools/pmc_calibrate.sh $OUT/pmc_cal > $OUT/pmc_cal.log 2>&1; tail -12 $OUT/pmc_cal.log
PASSES="insts waits fetch write" bash tools/pmc_passes.sh $OUT/pmc > $OUT/pmc.log 2>&1
python tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.csv 2>/dev/null; grep "poa_window" $OUT/pmc_summary.csv | cut -c1-200
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/stats -o stats -- python $OLDPWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline --sub-configs none > $OLDPWD/$OUT/stats.log 2>&1 )
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv; head -6 $OUT/kernel_stats.csv | cut -c1-250
( timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err ); echo "bench rc=$?"
cut -c1-1200 $OUT/bench.json
