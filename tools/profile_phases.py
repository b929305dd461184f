#!/usr/bin/env python3
"""Profiling aid: per-phase share of the graph-build kernel on the config-3 batch (needs a GPU)."""
import json
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomeworks_amd import cudapoa, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
windows = [[r.decode() for r in synthetic.generate_window(1000 + w)] for w in range(n)]
b = cudapoa.CudaPoaBatch(32, 1024, 8 << 30, band_mode="static_band", alignment_band_width=256, max_nodes_per_graph=3072)
for w in windows:
    assert b.add_poa_group(w)[0] == 0
b.generate_poa()
b.get_consensus_native()
ph = b.profile_phases()
tot = sum(ph.values())
k, o = b.relaunch_timed()
cells = b.total_cells()
rows = cells / 256.0 / n
print(json.dumps({"windows": n, "rows_per_window": rows, "fwd_ticks_per_row": ph["nw_forward"] / max(rows, 1), "kernel_ms": k, "output_ms": o, "ticks_per_window": ph,
                  "share": {a: round(v / tot, 4) for a, v in ph.items()}}))
