#!/bin/bash
# round 3, GPU call S: kernel trace of the default aligner's benchmark shapes (which kernel takes what)
set -u
TAG=${1:-r03s}
REPO=$(pwd)
OUT=gpurun_out/${TAG}
mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 420 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/prof -- python $REPO/tools/bench_default_aligner.py quick > $REPO/$OUT/bench.json 2> $REPO/$OUT/err.txt)
DB=$(find $OUT/prof -name "*.db" | head -1)
python3 - "$DB" > $OUT/dispatches.txt <<'PY'
import sqlite3,sys
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
rows=cur.execute("select name,grid_x,start,end from kernels order by start").fetchall()
for name,g,s,e in rows:
    if 'hirschberg' in name or 'hb_' in name:
        print(name.split('(')[0][-40:], g, round((e-s)/1e3,1), 'us')
PY
python tools/rocpd_summary.py "$DB" > $OUT/kernel_stats.csv
rm -rf $OUT/prof
cat $OUT/dispatches.txt | tail -40
