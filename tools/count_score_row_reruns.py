#!/usr/bin/env python3
"""How often does the walk of the 256-column pass need a score row the forward pass kept out of HBM (kNwNeedScoreRows)?
GWHIP_DEBUG bit 24 of the debug instantiation adds one to the "other" phase accumulator per rerun."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomeworks_amd import cudapoa, synthetic
n = 1024
windows = [[r.decode() for r in synthetic.generate_window(1000 + w)] for w in range(n)]
out = {}
for mode, flag in (("static_band", 1 << 24), ("static_band_reference_other_ticks", 32), ("adaptive_band", 1 << 24), ("adaptive_band_reference_other_ticks", 32)):
    os.environ["GWHIP_DEBUG"] = str(flag)
    name, mode = mode, mode.replace("_reference_other_ticks", "")
    b = cudapoa.CudaPoaBatch(32, 1024, 8 << 30, band_mode=mode, alignment_band_width=256, max_nodes_per_graph=3072)
    for w in windows:
        assert b.add_poa_group(w)[0] == 0
    b.generate_poa()
    b.get_consensus_native()
    per = b.profile_phases_per_window() if hasattr(b, "profile_phases_per_window") else None
    ph = b.profile_phases()
    out[name] = {"reads_per_window": 31, "other_accumulator_per_window_mean": ph["other"], "forward_ticks_per_window": ph["nw_forward"]}
    os.environ.pop("GWHIP_DEBUG")
print(json.dumps(out))
