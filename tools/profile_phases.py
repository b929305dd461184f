#!/usr/bin/env python3
"""Profiling aid: per-phase share of the graph-build kernel on the config-3 batch (needs a GPU)."""
import json
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomeworks_amd import cudapoa, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mode = sys.argv[2] if len(sys.argv) > 2 else "static_band"   # full_band: the reference benchmarks' BatchConfig(1024, 200)
windows = [[r.decode() for r in synthetic.generate_window(1000 + w)] for w in range(n)]
if mode == "full_band":
    b = cudapoa.CudaPoaBatch(200, 1024, 16 << 30, band_mode="full_band", max_nodes_per_graph=3072, matrix_sequence_dimension=1024)
else:
    b = cudapoa.CudaPoaBatch(32, 1024, 8 << 30, band_mode=mode, alignment_band_width=256, max_nodes_per_graph=3072)
for w in windows:
    assert b.add_poa_group(w)[0] == 0
b.generate_poa()
b.get_consensus_native()
ph = b.profile_phases()
tot = sum(ph.values())
k, o = b.relaunch_timed()
cells = b.total_cells()
rows = cells / 256.0 / n   # (full band: cells / 256 = 256-column passes, not rows)
print(json.dumps({"windows": n, "band_mode": mode, "rows_per_window": rows, "fwd_ticks_per_row": ph["nw_forward"] / max(rows, 1), "kernel_ms": k, "output_ms": o, "ticks_per_window": ph,
                  "share": {a: round(v / tot, 4) for a, v in ph.items()}}))
