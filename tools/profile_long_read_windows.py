#!/usr/bin/env python3
"""Profiling aid: which windows of the long-read set (BASELINE configs[3] inputs) bound their class, and in which phase.
Runs the windows of one size class (those whose longest read is in (max_seq / 2, max_seq)) in one batch with the phase
counters on and prints the slowest ones.
  python tools/profile_long_read_windows.py [max_sequence_size] [top]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genomeworks_amd import cudapoa, synthetic  # noqa: E402

max_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 30486
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
b = cudapoa.CudaPoaBatch(32, max_seq, 230 << 30, output_type="msa", band_mode="adaptive_band",
                         matrix_sequence_dimension=4 * 264, max_nodes_per_graph=3 * max_seq)
ids = []
for w in range(598):
    reads = synthetic.long_read_window(w, 32768)
    longest = max(len(s) for s in reads)
    if longest >= max_seq or longest < max_seq // 2:
        continue
    if b.add_poa_group(reads)[0] == 0:
        ids.append((w, longest, len(reads)))
b.generate_poa()
b.get_msa_native()
per = b.profile_phases_per_window()
rows = []
for (w, longest, n), ph in zip(ids, per):
    tot = sum(ph.values())
    rows.append({"window": w, "longest_read": longest, "reads": n, "ticks_M": round(tot / 1e6, 1),
                 "share": {k: round(v / tot, 3) for k, v in ph.items() if k != "other"}})
rows.sort(key=lambda r: -r["ticks_M"])
k, o = b.relaunch_timed()
print(json.dumps({"class_max_sequence_size": max_seq, "windows": len(ids), "kernel_ms": round(k, 1),
                  "ticks_per_ms_of_the_slowest": round(rows[0]["ticks_M"] * 1e6 / k) if rows else None, "slowest": rows[:top]}, indent=1))
