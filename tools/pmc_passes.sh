#!/bin/bash
# rocprofv3 PMC passes over one bench invocation (counters only with --kernel-trace, one group per pass).
# usage: tools/pmc_passes.sh <out_dir> ; run from the repo root on the GPU box
set -u
OUT=${1:-gpurun_out/pmc}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# SUBS: the sub-records whose kernels should run under the counters as well (aligner, default_aligner, long_reads)
CMD="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --sub-configs ${SUBS:-none}"
# PMC_CMD: another command to run under the counters instead (e.g. "python $REPO/tools/profile_phases.py 1024 full_band")
[ -n "${PMC_CMD:-}" ] && CMD="$PMC_CMD"
PASSES=${PASSES:-all}
pass() { name=$1; shift; case " $PASSES " in *" all "*|*" $name "*) ;; *) return;; esac; timeout ${PASS_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$REPO/$OUT/$name" -o $name -- $CMD > "$REPO/$OUT/$name.log" 2>&1; echo "$name rc=$?"; }
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES
pass waits SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_BRANCH
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass vmem SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES
pass tcc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL
pass tcp TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_REQUEST TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ_LATENCY
# instruction fetch, scalar cache, latency levels, issue mix (round 5: the long-read kernel's stall)
pass icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass dcache SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_WAVE_CYCLES
pass levels SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS
pass active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES
cd "$REPO"
find "$OUT" -name "*counter_collection.csv" | head
