#!/usr/bin/env python3
"""The default aligner (create_aligner(max_query, max_target, n): Hirschberg + Myers, the one pygenomeworks reaches) on the
reference's benchmark shapes (cudaaligner/benchmarks/main.cpp:39-67 BM_SingleAlignment: one pair of 100 .. 100 000 bp;
:69-143 BM_SingleBatchAlignment: 1024 pairs x 2048 bp) and on 2 000 pairs x 1 kbp: align_all() + sync_alignments() with the
pairs queued, best of a few repeats. GWHIP_HIRSCHBERG_WAVE=0 selects the one-lane-per-pair kernel (A/B)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomeworks_amd import cudaaligner, synthetic  # noqa: E402


def run(n, size, repeat=3):
    pairs = synthetic.generate_pairs(1, n, size, size // 30, size // 30, size // 30)
    pairs = [(q, t[:size]) for q, t in pairs]
    al = cudaaligner.CudaAlignerBatch(size, size, n, max_device_memory_allocator_caching_size=32 << 30)
    best = None
    for _ in range(repeat):
        for q, t in pairs:
            assert al.add_alignment(q, t) == 0
        t0 = time.perf_counter()
        al.align_all()
        assert al.sync() == n
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        k_ms = min(al.relaunch_timed() for _ in range(3))
        al.reset()
    return {"pairs": n, "length": size, "ms": round(best * 1e3, 3), "pairs_per_s": round(n / best, 1),
            "gcups_full_dp_equivalent": round(n * size * size / best / 1e9, 2), "kernels_ms": round(k_ms, 3)}


if __name__ == "__main__":
    shapes = [(1, 100), (1, 1000), (1, 10000), (1, 100000), (1024, 2048), (2000, 1000)]
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        shapes = [(1, 1000), (1, 10000), (1024, 2048), (2000, 1000)]
    out = {"kernel": "one lane per pair" if os.environ.get("GWHIP_HIRSCHBERG_WAVE") == "0" else "one wavefront per pair", "shapes": []}
    for n, size in shapes:
        out["shapes"].append(run(n, size))
        print(json.dumps(out["shapes"][-1]), file=sys.stderr, flush=True)
    print(json.dumps(out))
