#!/bin/bash
# kernel timeline of the long-read sub-record (which launches overlap, what the tail looks like)
set -u
TAG=${1:-r02tr}
mkdir -p gpurun_out/${TAG}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $GRAFT_REPO_ROOT/bench.py --sub-configs long_reads --no-cpu-baseline --steps 1 --warmup 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}/bench.err
F=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python3 - "$F" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}/timeline.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
ks=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:60],r.get('Grid_Size_X',''),r.get('LDS_Block_Size','')) for r in rows]
t0=min(k[0] for k in ks)
for s,e,n,g,l in sorted(ks):
    if (e-s)>5e6: print("%9.1f %9.1f %8.1f ms grid=%s lds=%s %s"%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,g,l,n))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/${TAG}/timeline.txt | tail -40
