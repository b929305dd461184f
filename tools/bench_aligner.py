#!/usr/bin/env python3
"""Throughput of the banded-Myers aligner on BASELINE configs[1] (10k pairs x ~1 kbp) and configs[4]
(short-read pairs, 150 bp; default 200k of the 1M per GPU share) on one MI355X.

Reports, per config: pairs/s and band GCUPS with the inputs resident in HBM (kernels only: relaunch + stream
sync), and pairs/s including sync_alignments() (D2H + host materialisation of the Alignment objects)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genomeworks_amd import cudaaligner, synthetic


def run(name, n_pairs, length, mut, ins, dele, max_bw, reps=5, seed=1):
    pairs = synthetic.generate_pairs(seed, n_pairs, length, mut, ins, dele)
    al = cudaaligner.CudaAlignerBatch(max_bandwidth=max_bw, max_device_memory_allocator_caching_size=16 << 30)
    t0 = time.perf_counter()
    for q, t in pairs:
        st = al.add_alignment(q, t)
        assert st == 0, st
    t_add = time.perf_counter() - t0
    al.align_all()
    n = len(pairs)
    cells = al.band_cells()  # stream sync; sync_alignments() would clear the queued pairs (like the reference)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        al.relaunch()
        al.band_cells()  # stream sync + tiny D2H
        ts.append(time.perf_counter() - t0)
    k = min(ts)
    t0 = time.perf_counter()
    al.relaunch()
    assert al.sync() == n
    full = time.perf_counter() - t0
    return {"config": name, "pairs": n, "length": length, "max_bandwidth": max_bw, "band_cells": cells,
            "kernel_resident_ms": round(k * 1e3, 3), "pairs_per_s_resident": round(n / k, 1),
            "band_gcups_resident": round(cells / k / 1e9, 2),
            "with_host_materialisation_ms": round(full * 1e3, 3), "pairs_per_s_with_sync": round(n / full, 1),
            "host_add_alignment_s": round(t_add, 3)}


if __name__ == "__main__":
    n5 = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    out = [run("configs[1]: 10k pairs x 1 kbp, <=33 sub/ins/del", 10000, 1000, 33, 33, 33, 1024),
           run("configs[4]: %d pairs x 150 bp, <=2 sub, <=1 ins, <=1 del" % n5, n5, 150, 2, 1, 1, 150, seed=3)]
    for o in out:
        print(json.dumps(o))
