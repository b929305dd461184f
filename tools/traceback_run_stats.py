#!/usr/bin/env python3
"""Run-length histogram of the banded traceback on the config-3 windows (CPU, oracle): how many steps a wave-parallel
"run skipping" walk would take per iteration (traceback). usage: traceback_run_stats.py [windows]"""
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

out = os.path.join(ROOT, "tools", "bin", "libtrs.so")
subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-I", os.path.join(ROOT, "oracle"), "-o", out,
                os.path.join(ROOT, "oracle", "poa_oracle.c"), os.path.join(ROOT, "tools", "traceback_run_stats.c")], check=True)
import oracle_poa as O  # noqa: E402
O.lib()  # sets up the signatures on the stock oracle library; the analysis build below takes its place
L = C.CDLL(out)
for name in ("poa_band_start_for_row", "poa_workspace_create", "poa_workspace_destroy", "poa_workspace_overflow_events",
             "poa_process_window", "poa_cfg_init", "poa_cfg_select_types"):
    getattr(L, name).restype = getattr(O._LIB, name).restype
    getattr(L, name).argtypes = getattr(O._LIB, name).argtypes
O._LIB = L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = O.make_cfg(1024, 32, 256, 1)
L.trs_install(16)
with O.Workspace(cfg) as ws:
    for w in range(n):
        L.trs_new_window()
        r = ws.process([x.decode() for x in synthetic.generate_window(1000 + w)])
        assert r["status"] == 0
run = np.zeros((2, 129), np.int64); kind = np.zeros((2, 4), np.int64); dist = np.zeros((2, 9), np.int64); steps = np.zeros(4, np.int64)
L.trs_new_window()
L.trs_get.argtypes = [C.c_void_p] * 4
L.trs_get(run.ctypes.data, kind.ctypes.data, dist.ctypes.data, steps.ctypes.data)
res = {"windows": n}
for k, name in ((0, "reads_1_15"), (1, "reads_16_31")):
    iters = int(run[k].sum())
    s = int(steps[k])
    lens_ = np.arange(129)
    res[name] = {
        "steps": s, "walks": int(steps[2 + k]), "iterations_with_run_skipping": iters, "steps_per_iteration": round(s / max(iters, 1), 2),
        "share_steps_one_row_one_column": round(kind[k][0] / max(s, 1), 4), "other_diagonal": int(kind[k][1]), "vertical": int(kind[k][2]),
        "horizontal": int(kind[k][3]),
        "row_distance_hist_0_to_8plus": [int(x) for x in dist[k]],
        "run_len_hist_0_1_2_3_4_5to8_9to16_17to32_33plus": [int(run[k][0]), int(run[k][1]), int(run[k][2]), int(run[k][3]), int(run[k][4]),
                                                           int(run[k][5:9].sum()), int(run[k][9:17].sum()), int(run[k][17:33].sum()), int(run[k][33:].sum())],
        "iterations_if_capped_at_63": int(sum(int(run[k][r]) * (1 + r // 64) for r in range(129))),
    }
print(json.dumps(res, indent=1))
