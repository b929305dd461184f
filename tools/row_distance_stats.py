#!/usr/bin/env python3
"""How far up a row's predecessors are on the config-3 windows (CPU, oracle): the share of DP rows a packed forward pass with an
LDS ring of R rows could serve from the ring, for R = 4 (what bands of 384 / 512 columns leave room for at four blocks per
CU) against R = 8 (band 256 today). usage: row_distance_stats.py [windows]"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from genomeworks_amd import synthetic  # noqa: E402

os.makedirs(os.path.join(ROOT, "tools", "bin"), exist_ok=True)
out = os.path.join(ROOT, "tools", "bin", "librds.so")
subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-I", os.path.join(ROOT, "oracle"), "-o", out,
                os.path.join(ROOT, "oracle", "poa_oracle.c"), os.path.join(ROOT, "tools", "row_distance_stats.c")], check=True)
import oracle_poa as O  # noqa: E402
O.lib()  # sets up the signatures on the stock oracle library; the analysis build below takes its place
L = C.CDLL(out)
for name in ("poa_band_start_for_row", "poa_workspace_create", "poa_workspace_destroy", "poa_workspace_overflow_events",
             "poa_process_window", "poa_cfg_init", "poa_cfg_select_types"):
    getattr(L, name).restype = getattr(O._LIB, name).restype
    getattr(L, name).argtypes = getattr(O._LIB, name).argtypes
O._LIB = L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
res = {"windows": n}
L.rds_install()
L.rds_get.argtypes = [C.c_void_p] * 3
prev = np.zeros(17 + 8 + 2, np.int64)
for band in (256, 512):
    cfg = O.make_cfg(1024, 32, band, 1)
    with O.Workspace(cfg) as ws:
        for w in range(n):
            assert ws.process([x.decode() for x in synthetic.generate_window(1000 + w)])["status"] == 0
    far = np.zeros(17, np.int64); cnt = np.zeros(8, np.int64); tot = np.zeros(2, np.int64)
    L.rds_get(far.ctypes.data, cnt.ctypes.data, tot.ctypes.data)
    cur = np.concatenate([far, cnt, tot])
    far, cnt, tot = (cur - prev)[:17], (cur - prev)[17:25], (cur - prev)[25:]
    prev = cur
    rows = int(tot[0])
    within = lambda d: round(float(far[:d + 1].sum()) / rows, 4)  # noqa: E731
    res["band_%d" % band] = {
        "rows": rows, "share_previous_row_only": round(float(tot[1]) / rows, 4),
        "share_all_predecessors_within_rows": {str(d): within(d) for d in (1, 2, 3, 4, 7, 15)},
        "share_more_than_three_predecessors": round(float(cnt[4:].sum()) / rows, 4),
        "farthest_predecessor_hist_0_to_16plus": [int(x) for x in far],
    }
print(json.dumps(res, indent=1))
