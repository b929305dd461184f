#!/usr/bin/env python3
"""Kernel timeline (start, end, duration in ms; grid, LDS, name) of a rocprofv3 rocpd (.db) result, in start order:
python tools/kernel_timeline.py results.db [min_ms]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = db.execute("select name, start, end, grid_x, lds_size from kernels order by start").fetchall()
t0 = rows[0][1] if rows else 0
for name, start, end, grid, lds in rows:
    ms = (end - start) / 1e6
    if ms >= min_ms:
        print("%10.1f %10.1f %9.1f ms grid=%d lds=%d %s" % ((start - t0) / 1e6, (end - t0) / 1e6, ms, grid, lds, name[:100]))
