#!/usr/bin/env python3
"""Timing ablation of the metric kernel's HBM stores (needs a GPU): the graph-build kernel of the config-3 batch relaunched
over the buffers of a complete launch with GWHIP_DEBUG bit 26 (no score-row stores) / bit 27 (no move-row stores) / both.
All arms run the debug instantiation (bit 31 is a no-op selector), so they compare like with like."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomeworks_amd import cudapoa, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
windows = [[r.decode() for r in synthetic.generate_window(1000 + w)] for w in range(n)]
b = cudapoa.CudaPoaBatch(32, 1024, 8 << 30, band_mode="static_band", alignment_band_width=256, max_nodes_per_graph=3072)
for w in windows:
    assert b.add_poa_group(w)[0] == 0
b.generate_poa()
b.get_consensus_native()
out = {}
for name, flags in (("production", 0), ("debug_build_all_stores", 1 << 31), ("no_score_rows", (1 << 31) | (1 << 26)), ("no_move_rows", (1 << 31) | (1 << 27)),
                    ("no_score_no_move_rows", (1 << 31) | (1 << 26) | (1 << 27)), ("debug_build_all_stores_again", 1 << 31)):
    if flags:
        os.environ["GWHIP_DEBUG"] = str(flags - (1 << 32) if flags >= (1 << 31) else flags)
    else:
        os.environ.pop("GWHIP_DEBUG", None)
    ks = [b.relaunch_timed()[0] for _ in range(4)]
    out[name] = round(min(ks[1:]), 3)
os.environ.pop("GWHIP_DEBUG", None)
print(json.dumps({"windows": n, "graph_build_kernel_ms": out}))
