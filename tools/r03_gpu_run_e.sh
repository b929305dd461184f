#!/bin/bash
# round 3, GPU call E: two wavefronts per window (lead + trail forward pass): row microbenchmark with one and two wavefronts,
# GPU suite file by file, phase breakdown with two wavefronts and with the lead alone (GWHIP_DEBUG bit 24), bench
set -u
TAG=${1:-r03e}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 tools/bin/microbench_rows 1024 8 1 > $OUT/microbench_rows.json 2> $OUT/microbench_rows.err; echo "microbench rc=$?"
cut -c1-200 $OUT/microbench_rows.json
: > $OUT/pytest.log
for f in tests/test_gpu_config_goldens.py tests/test_gpu_poa.py tests/test_gpu_poa_hooks.py $(ls tests/test_*.py | grep -v "test_gpu_config_goldens\|test_gpu_poa.py\|test_gpu_poa_hooks"); do
  echo "== $f" >> $OUT/pytest.log
  ( timeout 900 python -m pytest $f -m gpu -q -x 2>&1 | tail -${PYTAIL:-25} ) >> $OUT/pytest.log
done
grep -E "^== |passed|failed|error|Aborted|fault" $OUT/pytest.log | grep -B1 -E "passed|failed|error|Aborted|fault" | grep -v "^--" | tail -40
timeout 200 python tools/profile_phases.py 1024 2>$OUT/phases_new.err | tail -1 > $OUT/phase_breakdown_two_wavefronts.json
GWHIP_DEBUG=16777216 timeout 200 python tools/profile_phases.py 1024 2>$OUT/phases_solo.err | tail -1 > $OUT/phase_breakdown_lead_alone.json
cut -c1-900 $OUT/phase_breakdown_two_wavefronts.json; echo; cut -c1-900 $OUT/phase_breakdown_lead_alone.json; echo
( timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --sub-configs none > $OUT/bench.json 2> $OUT/bench.err ); echo "bench rc=$?"
cut -c1-1500 $OUT/bench.json
