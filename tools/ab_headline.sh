#!/bin/bash
# Same-box A/B of two builds of the native libraries: genomeworks_amd/lib (candidate) against genomeworks_amd/lib_old
# (baseline), alternating, headline only. usage (on the GPU box, repo root): bash tools/ab_headline.sh [rounds] [bench args]
ROUNDS=${1:-3}
shift
ARGS=${@:---sub-configs none --steps 20 --warmup 3 --no-cpu-baseline}
mkdir -p gpurun_out/ab
run() { # $1 = label
    python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$1', 'kernel_ms', d['roofline']['kernel_ms'], 'step_ms', d['ms_per_step'], 'gcups', d['value'], 'golden', d['equals_oracle_golden'])"
}
for i in $(seq $ROUNDS); do
    run candidate
    mv genomeworks_amd/lib genomeworks_amd/lib_new && mv genomeworks_amd/lib_old genomeworks_amd/lib
    run baseline
    mv genomeworks_amd/lib genomeworks_amd/lib_old && mv genomeworks_amd/lib_new genomeworks_amd/lib
done
