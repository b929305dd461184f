#!/usr/bin/env python3
"""Cycles per DP row of each row kind of the metric kernel's forward pass (needs a GPU; debug instantiation with s_memtime
around the rows of ONE kind per launch, GWHIP_DEBUG bits 28-30 and 12): poa_forward_moves.h."""
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


def other(flags):
    os.environ["GWHIP_DEBUG"] = str(flags - (1 << 32) if flags >= (1 << 31) else flags)
    v = b.profile_phases()
    os.environ.pop("GWHIP_DEBUG", None)
    return v


base = other(1 << 31)
res = {"windows": n, "baseline_ticks_per_window": base, "kinds": {}}
names = ["0 previous row, band not moved", "1 previous row, band moved one quad", "2 one predecessor from the ring", "3 2..6 predecessors from the ring", "4 general"]
for k in range(5):
    cyc = other(((k + 1) << 28))
    cnt = other(((k + 1) << 28) | (1 << 12))
    rows = cnt["other"] - base["other"]
    ticks = cyc["other"] - base["other"]
    res["kinds"][names[k]] = {"rows_per_window": round(rows, 1), "ticks_per_window": round(ticks), "ticks_per_row": round(ticks / max(rows, 1e-9), 1),
                              "forward_ticks_with_timers": round(cyc["nw_forward"])}
print(json.dumps(res))
