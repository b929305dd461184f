#!/usr/bin/env python3
"""Profiling aid: per-phase share of the graph-build kernel on long-read MSA windows (BASELINE configs[3] inputs).
  python tools/profile_long_read.py [first_window] [count] [max_sequence_size]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from genomeworks_amd import cudapoa, synthetic  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 64
max_seq = int(sys.argv[3]) if len(sys.argv) > 3 else 30106
b = cudapoa.CudaPoaBatch(32, max_seq, 100 << 30, output_type="msa", band_mode="adaptive_band",
                         matrix_sequence_dimension=4 * 264, max_nodes_per_graph=3 * max_seq)
n = 0
for w in range(first, first + count):
    reads = [s for s in synthetic.long_read_window(w, 32768) if len(s) < max_seq]
    if reads and b.add_poa_group(reads)[0] == 0:
        n += 1
b.generate_poa()
b.get_msa_native()
ph = b.profile_phases()
tot = sum(ph.values())
k, o = b.relaunch_timed()
print(json.dumps({"windows": n, "cells": b.total_cells(), "kernel_ms": k, "output_ms": o,
                  "mean_ticks_per_window": {a: round(v) for a, v in ph.items()},
                  "share": {a: round(v / tot, 4) for a, v in ph.items()}}))
