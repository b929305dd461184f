#!/bin/bash
# rocprofv3 PMC passes over one bench invocation (counters only with --kernel-trace, one group per pass).
# usage: tools/pmc_passes.sh <out_dir> ; run from the repo root on the GPU box
set -u
OUT=${1:-gpurun_out/pmc}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --sub-configs none"
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$REPO/$OUT/$name" -o $name -- $CMD > "$REPO/$OUT/$name.log" 2>&1; echo "$name rc=$?"; }
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES
pass waits SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_BRANCH
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL
pass fetch FETCH_SIZE
pass write WRITE_SIZE
cd "$REPO"
find "$OUT" -name "*counter_collection.csv" | head
