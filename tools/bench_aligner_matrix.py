#!/usr/bin/env python3
"""The reference's cudaaligner benchmark matrix (cudaaligner/benchmarks/main.cpp:96-165, BM_SingleBatchAlignment) on
MI355X: every aligner class x alignments-per-batch x genome size, time of align_all() + sync_alignments() with the
pairs already queued (as the reference times it). Inputs as there: genome_1 random, genome_2 = genome_1 with
size/30 substitutions, insertions and deletions each (about 10 %), minstd_rand(1) streamed over the pairs.

  python tools/bench_aligner_matrix.py [--json out.json] [--full]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomeworks_amd import cudaaligner, synthetic  # noqa: E402

ALGORITHMS = ["ukkonen", "myers", "myers_banded", "hirschberg_myers"]


def bytes_needed(algorithm, n, size):
    if algorithm == "ukkonen":  # int16 band: (101 + size/20) slots x 2 (size + 1) anti-diagonals
        return n * (101 + size // 20) * 2 * (size + 1) * 2
    if algorithm == "myers":  # pv, mv, score per (word, column)
        return n * ((size + 31) // 32) * (size + 1) * 12
    if algorithm == "myers_banded":
        return n * (min(1024, size) // 32 + 1) * (size + 1) * 12
    return n * size * 64


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    ap.add_argument("--full", action="store_true", help="the reference's whole grid (32..1024 x 512..65536)")
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    batches = [32, 128, 512, 1024] if args.full else [32, 1024]
    sizes = [512, 2048, 8192, 32768, 65536] if args.full else [512, 2048, 8192]
    rows = []
    for algorithm in ALGORITHMS:
        for n in batches:
            for size in sizes:
                if bytes_needed(algorithm, n, size) > 120 << 30:
                    print("%-17s %5d x %6d: skipped (matrices would not fit)" % (algorithm, n, size), flush=True)
                    continue
                pairs = synthetic.generate_pairs(1, n, size, size // 30, size // 30, size // 30)
                pairs = [(q, t[:size]) for q, t in pairs]
                if algorithm == "myers_banded":
                    al = cudaaligner.CudaAlignerBatch(max_bandwidth=1024, max_device_memory_allocator_caching_size=160 << 30)
                else:
                    al = cudaaligner.CudaAlignerBatch(size, size, n, algorithm=algorithm,
                                                      max_device_memory_allocator_caching_size=160 << 30)
                best = None
                for _ in range(args.repeat):
                    for q, t in pairs:
                        st = al.add_alignment(q, t)
                        assert st == 0, st
                    t0 = time.perf_counter()
                    al.align_all()
                    al.sync()
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                    al.reset()
                row = {"algorithm": algorithm, "alignments_per_batch": n, "genome_size": size, "ms": round(best * 1e3, 3),
                       "pairs_per_s": round(n / best, 1), "gcups_full_dp_equivalent": round(n * size * size / best / 1e9, 2)}
                rows.append(row)
                print("%-17s %5d x %6d: %9.2f ms  %10.0f pairs/s  %8.1f GCUPS (full-DP equivalent)"
                      % (algorithm, n, size, row["ms"], row["pairs_per_s"], row["gcups_full_dp_equivalent"]), flush=True)
                del al
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
