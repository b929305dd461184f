#!/bin/bash
# round 3, GPU call A: full GPU suite on the move-byte forward pass + run-skipping traceback, phase breakdown of the new and
# of the round-2 path (GWHIP_DEBUG bit 25), headline bench line.
set -u
TAG=${1:-r03a}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -80 ) > $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 200 python tools/profile_phases.py 1024 2>$OUT/phases_new.err | tail -1 > $OUT/phase_breakdown_new.json
GWHIP_DEBUG=33554432 timeout 200 python tools/profile_phases.py 1024 2>$OUT/phases_old.err | tail -1 > $OUT/phase_breakdown_round2_path.json
cut -c1-900 $OUT/phase_breakdown_new.json; echo; cut -c1-900 $OUT/phase_breakdown_round2_path.json; echo
( timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --sub-configs none > $OUT/bench.json 2> $OUT/bench.err ); echo "bench rc=$?"
cut -c1-1500 $OUT/bench.json
