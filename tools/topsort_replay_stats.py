#!/usr/bin/env python3
"""How much of a Kahn re-sort the incremental algorithm replays in blocks, at short-read and at long-read divergence
(CPU only: the scalar model of oracle/topsort_incr_model.inc next to the plain restatement; graphs scaled to the model's
3072-node limit).   python tools/topsort_replay_stats.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_poa as O
from genomeworks_amd import synthetic
# divergence like the long-read set (8-12 %, 1 : 2 : 2 substitutions : insertions : deletions), scaled to a graph the model holds
for (blen, reads, div) in [(960, 32, 0.05), (900, 20, 0.10), (900, 32, 0.10), (600, 32, 0.12)]:
    mut, ins, dele = (int(2 * blen * div * f) for f in (0.2, 0.4, 0.4))
    if div == 0.05:
        mut, ins, dele = 48, 24, 24
    tot = {}
    for w in range(6):
        rd = [r.decode() for r in synthetic.generate_window(7000 + w, blen, reads, mut, ins, dele)]
        cfg = O.make_cfg(1024, 32, 256, 1) if div == 0.05 else O.make_cfg(1200, 32, 256, 2)
        with O.Workspace(cfg) as ws, O.topsort_model() as m:
            ws.process(rd)
            st = m.stats()
        for k, v in st.items():
            tot[k] = tot.get(k, 0) + v
    n = tot["nodes"]
    print("backbone %d, %d reads, divergence %.2f: nodes/read-sort %.0f, replayed in blocks %.1f %%, ordinary steps %.1f %%, nodes per block %.1f, mismatches %d"
          % (blen, reads, div, n / max(tot["reads"], 1), 100.0 * tot["block_nodes"] / n, 100.0 * tot["real_steps"] / n,
             tot["block_nodes"] / max(tot["blocks"], 1), tot["mismatch"]))
