#!/bin/bash
# round 3, GPU call B: what bounds the forward pass -- store ablations, cycles per row kind, PMC passes (instruction mix,
# waits, VMEM issue cycles, L2 -> memory write requests and stalls, TLB).
set -u
TAG=${1:-r03b}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/r03_store_ablation.py 1024 > $OUT/store_ablation.json 2> $OUT/store_ablation.err; cat $OUT/store_ablation.json
timeout 400 python tools/r03_row_kind_cycles.py 1024 > $OUT/row_kind_cycles.json 2> $OUT/row_kind_cycles.err; cat $OUT/row_kind_cycles.json
PASSES="${PASSES:-insts waits vmem tcc tcp write fetch}" bash tools/pmc_passes.sh $OUT/pmc > $OUT/pmc.log 2>&1
python tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.csv 2>/dev/null; grep "poa_window" $OUT/pmc_summary.csv | cut -c1-200
