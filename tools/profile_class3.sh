#!/bin/bash
# Why rows are class 3 (general routine) in the packed forward pass: count per reason, per window (needs a GPU).
OUT=${1:-gpurun_out/class3.txt}
: > $OUT
i=0
for flag in 4096 16384 20480 524288 528384; do
  i=$((i+1))
  v=$(GWHIP_DEBUG=$flag python tools/profile_phases.py 1024 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d['ticks_per_window']['other'])")
  echo "reason=$i other(count x1000 + ~85k): $v" | tee -a $OUT
done
