#!/bin/bash
# kernel timeline of the long-read sub-record (which launches overlap, what the tail looks like)
set -u
TAG=${1:-r03w}
REPO=$(pwd)
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp

(cd /tmp && rocprofv3 --kernel-trace -d $REPO/$OUT/tr -- python $REPO/bench.py --sub-configs long_reads --no-cpu-baseline --steps 1 --warmup 0 > $REPO/$OUT/bench.json 2> $REPO/$OUT/bench.err)
DB=$(find $OUT/tr -name "*.db" | head -1)
python3 - "$DB" > $OUT/timeline.txt <<'PY'
import sqlite3,sys
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
rows=cur.execute("select name,grid_x,lds_size,start,end from kernels order by start").fetchall()
t0=min(r[3] for r in rows)
for name,g,l,s,e in rows:
    if e-s>5e6: print("%9.1f %9.1f %8.1f ms grid=%s lds=%s %s"%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,g,l,name[:70]))
PY
rm -rf $OUT/tr
tail -16 $OUT/timeline.txt
