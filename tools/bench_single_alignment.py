#!/usr/bin/env python3
"""BM_SingleAlignment shapes of the reference (cudaaligner/benchmarks/main.cpp:39-67): one pair of 100 .. 100 000 bases through
the default aligner, align_all() + sync_alignments()."""
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomeworks_amd import cudaaligner

rng = random.Random(5)
rows = []
for n in (100, 1000, 10000, 100000):
    q = "".join(rng.choice("ACGT") for _ in range(n))
    t = list(q)
    for _ in range(n // 10):
        op, p = rng.random(), rng.randrange(len(t))
        if op < 0.4:
            t[p] = rng.choice("ACGT")
        elif op < 0.7:
            t.insert(p, rng.choice("ACGT"))
        else:
            del t[p]
    t = "".join(t)
    mx = max(len(q), len(t)) + 16
    al = cudaaligner.CudaAlignerBatch(mx, mx, 1, max_device_memory_allocator_caching_size=24 << 30)
    best = 1e9
    for rep in range(3):
        al.reset()
        assert al.add_alignment(q, t) == 0
        t0 = time.perf_counter()
        al.align_all()
        al.sync()
        best = min(best, time.perf_counter() - t0)
    rows.append({"bases": n, "ms": round(best * 1e3, 3)})
    print(rows[-1], flush=True)
print(json.dumps(rows))
