#!/bin/bash
# instruction-cache and issue counters of the metric kernel (one PMC pass per counter group)
set -u
TAG=${1:-r02ic}
mkdir -p gpurun_out/${TAG}
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|SQ_INSTS_VALU |SQ_INSTS_SALU|SQ_WAIT_INST|SQ_INST_CYCLES|SQ_ACTIVE_INST|SQ_BUSY_CY|SQ_WAVE_CYCLES|SQ_INSTS_LDS|SQ_INSTS_SMEM|SQ_WAIT_ANY|SQ_WAIT_INST_ANY" | head -40 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}/avail.txt
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_IFETCH"; do
  d=/tmp/pmc_$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --sub-configs none --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}/err.txt
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $GRAFT_REPO_ROOT/gpurun_out/${TAG}/counters.txt <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'][:50]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
for k,v in acc.items():
    if 'poa_window' in k:
        print(k, {c: round(x/ max(1,n[(k,c)]),1) for c,x in v.items()}, "launches", max(n[(k,c)] for c in v))
PY
done
cat $GRAFT_REPO_ROOT/gpurun_out/${TAG}/avail.txt | cut -c1-150; cat $GRAFT_REPO_ROOT/gpurun_out/${TAG}/counters.txt; tail -3 $GRAFT_REPO_ROOT/gpurun_out/${TAG}/err.txt
