#!/bin/bash
# Sub-phase profile of the graph-build kernel on the config-3 batch (needs a GPU): one run per selector, the
# selected quantity arrives in the "other" accumulator (minus the ~0.09 M cycles "other" always holds).
# traceback selectors: GWHIP_DEBUG bits 22-24, topsort: bits 25-27, forward: bits 28-29 (class 3 rows; the class 2
# timer, selector 3, only exists in a build with -DGWHIP_PROFILE_CLASS2), merge passes: bits 16-18 (see the kernels).
OUT=${1:-gpurun_out/subphases.txt}
N=${2:-1024}
: > $OUT
for sel in 0 1 2 3 4 5 6 7; do
  v=$(GWHIP_DEBUG=$((sel << 22)) python tools/profile_phases.py $N 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d['ticks_per_window']['other'], d['ticks_per_window']['sink_traceback'], d['kernel_ms'])")
  echo "traceback sel=$sel other/tb/kernel_ms: $v" | tee -a $OUT
done
for sel in 1 2 3 4 5 6; do
  v=$(GWHIP_DEBUG=$((sel << 25)) python tools/profile_phases.py $N 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d['ticks_per_window']['other'], d['ticks_per_window']['topsort'], d['kernel_ms'])")
  echo "topsort sel=$sel other/topsort/kernel_ms: $v" | tee -a $OUT
done
for sel in 1 2 3; do
  v=$(GWHIP_DEBUG=$((sel << 28)) python tools/profile_phases.py $N 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d['ticks_per_window']['other'], d['ticks_per_window']['nw_forward'], d['kernel_ms'])")
  echo "forward sel=$sel other/forward/kernel_ms: $v" | tee -a $OUT
done
for sel in 1 2 3 4; do
  v=$(GWHIP_DEBUG=$((sel << 16)) python tools/profile_phases.py $N 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d['ticks_per_window']['other'], d['ticks_per_window']['graph_merge'], d['kernel_ms'])")
  echo "merge sel=$sel (1 load, 2 classify, 3 create, 4 edges) other/merge/kernel_ms: $v" | tee -a $OUT
done
