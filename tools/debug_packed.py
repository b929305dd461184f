#!/usr/bin/env python3
"""Debug aid: run the first N config-3 windows under several GWHIP_DEBUG ablations and compare consensus/status."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomeworks_amd import cudapoa, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
flags = [int(x) for x in sys.argv[2:]] or [256, 0, 512, 1024, 2048, 512 + 2048]
windows = [[r.decode() for r in synthetic.generate_window(1000 + w)] for w in range(n)]

def run(flag):
    os.environ["GWHIP_DEBUG"] = str(flag)
    b = cudapoa.CudaPoaBatch(32, 1024, 2 << 30, band_mode="static_band", alignment_band_width=256, max_nodes_per_graph=3072)
    for w in windows:
        assert b.add_poa_group(w)[0] == 0
    b.generate_poa()
    return b.get_consensus()

ref = run(flags[0])
print("reference flag", flags[0], "status nonzero:", sum(1 for s in ref[2] if s != 0))
for f in flags[1:]:
    cons, cov, st = run(f)
    bad = [i for i in range(n) if cons[i] != ref[0][i] or cov[i] != ref[1][i] or st[i] != ref[2][i]]
    print("flag", f, "mismatching windows:", len(bad), bad[:10], "statuses:", [st[i] for i in bad[:10]])
