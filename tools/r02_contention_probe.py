#!/usr/bin/env python3
"""Experiment: N copies of the heaviest long-read window in one launch, N = 1, 16, 64, 128: wall time and mean cycles per
phase. Cycles constant while the wall time grows = clocks; cycles growing in the memory-latency-bound phases = the memory
system under load.   python tools/r02_contention_probe.py [window] """
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genomeworks_amd import cudapoa, synthetic
w = int(sys.argv[1]) if len(sys.argv) > 1 else 389
max_seq = 30486
reads = [s for s in synthetic.long_read_window(w, 32768) if len(s) < max_seq]
for n in (1, 16, 64, 128):
    b = cudapoa.CudaPoaBatch(32, max_seq, 200 << 30, output_type="msa", band_mode="adaptive_band",
                             matrix_sequence_dimension=4 * 264, max_nodes_per_graph=3 * max_seq)
    got = 0
    for _ in range(n):
        if b.add_poa_group(reads)[0] == 0:
            got += 1
    b.generate_poa()
    b.get_msa_native()
    ph = b.profile_phases()
    k, o = b.relaunch_timed()
    print(json.dumps({"copies": got, "kernel_ms": round(k, 1), "output_ms": round(o, 1),
                      "mean_Mcycles_per_window": {a: round(v / 1e6, 1) for a, v in ph.items()},
                      "sum_Mcycles": round(sum(ph.values()) / 1e6, 1),
                      "implied_GHz": round(sum(ph.values()) / (k * 1e-3) / 1e9, 3)}), flush=True)
    del b
