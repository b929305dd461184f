#!/bin/bash
# round 3, GPU call A: GPU suite (file by file, so that a device fault in one file does not hide the others) on the move-byte
# forward pass + run-skipping traceback, phase breakdown of the new and of the round-2 path (GWHIP_DEBUG bit 25), headline bench.
set -u
TAG=${1:-r03a}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/pytest.log
for f in tests/test_gpu_config_goldens.py tests/test_gpu_poa.py tests/test_gpu_poa_hooks.py $(ls tests/test_*.py | grep -v "test_gpu_config_goldens\|test_gpu_poa.py\|test_gpu_poa_hooks"); do
  echo "== $f" >> $OUT/pytest.log
  ( timeout 900 python -m pytest $f -m gpu -q -x 2>&1 | tail -${PYTAIL:-25} ) >> $OUT/pytest.log
done
grep -E "^== |passed|failed|error|Aborted|fault" $OUT/pytest.log | tail -60
timeout 200 python tools/profile_phases.py 1024 2>$OUT/phases_new.err | tail -1 > $OUT/phase_breakdown_new.json
GWHIP_DEBUG=33554432 timeout 200 python tools/profile_phases.py 1024 2>$OUT/phases_old.err | tail -1 > $OUT/phase_breakdown_round2_path.json
cut -c1-900 $OUT/phase_breakdown_new.json; echo; cut -c1-900 $OUT/phase_breakdown_round2_path.json; echo
( timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --sub-configs none > $OUT/bench.json 2> $OUT/bench.err ); echo "bench rc=$?"
cut -c1-1500 $OUT/bench.json
