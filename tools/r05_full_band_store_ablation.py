#!/usr/bin/env python3
"""Full band: is the four-pass row bound by its HBM stores? Relaunches of the debug instantiation over the buffers of an
unablated launch (the matrices already hold what the skipped stores would write): GWHIP_DEBUG bit 26 = no score-row stores,
bit 27 = no move-row stores (poa_forward_moves_full.h)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomeworks_amd import cudapoa, synthetic

n = 1024
windows = [[r.decode() for r in synthetic.generate_window(1000 + w)] for w in range(n)]
b = cudapoa.CudaPoaBatch(200, 1024, 16 << 30, band_mode="full_band", max_nodes_per_graph=3072, matrix_sequence_dimension=1024)
for w in windows:
    assert b.add_poa_group(w)[0] == 0
b.generate_poa()
b.get_consensus_native()
out = {}
for name, flag in (("production", None), ("debug_instantiation", 32), ("no_score_rows", (1 << 14) | (1 << 26)), ("no_move_rows", (1 << 14) | (1 << 27)),
                   ("neither", (1 << 14) | (1 << 26) | (1 << 27)), ("debug_instantiation_again", 32), ("production_again", None)):
    if flag is None:
        os.environ.pop("GWHIP_DEBUG", None)
    else:
        os.environ["GWHIP_DEBUG"] = str(flag)
    ms = min(b.relaunch_timed()[0] for _ in range(3))
    out[name] = round(ms, 2)
os.environ.pop("GWHIP_DEBUG", None)
print(json.dumps(out))
