#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd (.db) result into the per-kernel stats table we commit under profiles/ (CSV)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("Name,Calls,TotalDurationUs,AverageUs,Percentage,VGPRs,SGPRs,LDS,Grid,Workgroup")
rows = cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc").fetchall()
for name, calls, total, avg, pct in rows:
    k = cur.execute("select vgpr_count,sgpr_count,lds_size,grid_x,workgroup_x from kernels where name=? limit 1", (name,)).fetchone()
    short = name if len(name) < 120 else name[:117] + "..."
    print('"%s",%d,%d,%.0f,%.2f,%s' % (short, calls, total, avg, pct, ",".join(str(x) for x in k)))
