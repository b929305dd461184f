#!/bin/bash
# One GPU-box session: parity suite, phase breakdown (incremental vs full re-sort topsort), bench line.
# usage (through gpurun): bash tools/gpu_check.sh <tag>
set -u
TAG=${1:-check}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 200 python tools/profile_phases.py 1024 2>/dev/null | tail -1 > $OUT/phase_breakdown.json
GWHIP_DEBUG=2097152 timeout 200 python tools/profile_phases.py 1024 2>/dev/null | tail -1 > $OUT/phase_breakdown_full_resort.json
cut -c1-700 $OUT/phase_breakdown.json; echo; cut -c1-700 $OUT/phase_breakdown_full_resort.json; echo
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
