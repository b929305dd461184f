#!/usr/bin/env python3
"""Oracle goldens of the default aligner (Hirschberg + Myers) on the shapes bench.py publishes.

TEST INFRASTRUCTURE. The reference's aligner benchmark (cudaaligner/benchmarks/main.cpp:39-67 BM_SingleAlignment: one
pair of 100 .. 100 000 bases; :69-143 BM_SingleBatchAlignment: 1024 pairs x 2048 bases) plus 2 000 pairs x 1 kbp run
through `create_aligner(max_query, max_target, n)`. This script runs oracle/hirschberg_oracle.c on exactly the pairs
bench.py's `bench_default_aligner` generates and commits, per shape, a sha256 over every pair's (status, state sequence)
and the pairs' edit distances:

  default_aligner_goldens.json   {"<pairs>x<length>": {"pairs", "length", "states_sha256", "edit_distance_sum", "states_total"}}

  python tests/golden/make_default_aligner_goldens.py [--procs N]

bench.py compares every shape's digest inside the run (`equals_oracle_golden` per row);
tests/test_gpu_aligner.py::test_default_aligner_benchmark_shapes_equal_the_golden does the same under pytest.
"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SHAPES = [(1, 100), (1, 1000), (1, 10000), (1, 100000), (1024, 2048), (2000, 1000)]


def shape_pairs(n, size):
    """The pairs of bench.py's bench_default_aligner for one shape (seed 1, size // 30 trials of each edit kind, target cut
    to `size`)."""
    from genomeworks_amd import synthetic
    pairs = synthetic.generate_pairs(1, n, size, size // 30, size // 30, size // 30)
    return [(q, t[:size]) for q, t in pairs]


def pair_record(status, states):
    """What the digest is taken over for one pair: status, number of states, the states (forward order, one byte each)."""
    b = bytes(bytearray(int(s) & 0xFF for s in states))
    return struct.pack("<ii", int(status), len(b)) + b


def digest(records):
    h = hashlib.sha256()
    for r in records:
        h.update(r)
    return h.hexdigest()


def _one(job):
    import oracle_aligner as A
    q, t, size = job
    ref = A.hirschberg(q, t, size)
    return pair_record(ref["status"], ref["states"]), ref["edit_distance"], len(ref["states"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    args = ap.parse_args()
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    out = {}
    with mp.get_context("fork").Pool(args.procs) as pool:
        for n, size in SHAPES:
            pairs = shape_pairs(n, size)
            res = pool.map(_one, [(q, t, size) for q, t in pairs], chunksize=max(1, n // (args.procs * 8)))
            out["%dx%d" % (n, size)] = {"pairs": n, "length": size, "states_sha256": digest(r for r, _, _ in res),
                                        "edit_distance_sum": int(sum(e for _, e, _ in res)),
                                        "states_total": int(sum(s for _, _, s in res))}
            print("%dx%d" % (n, size), out["%dx%d" % (n, size)], flush=True)
    with open(os.path.join(HERE, "default_aligner_goldens.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
