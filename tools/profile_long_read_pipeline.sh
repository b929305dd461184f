#!/bin/bash
# Counters of the multi-wave forward pass (generic_forward_skew, GWHIP_DEBUG bits 12-15; summed over the eight wavefronts,
# they arrive in the "other" phase) for one long-read window: tools/profile_long_read_pipeline.sh [window] > out.txt
W=${1:-266}
for sel in 0 1 2 4 5 6 7 8 10 12 13; do
  v=$(GWHIP_DEBUG=$((sel << 12)) python tools/profile_long_read.py $W 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); m=d['mean_ticks_per_window']; print(m['other'], m['nw_forward'], round(d['kernel_ms'],1))")
  echo "sel=$sel other/forward/kernel_ms: $v"
done
