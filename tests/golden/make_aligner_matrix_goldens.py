#!/usr/bin/env python3
"""Oracle goldens for the cells of the reference's aligner benchmark matrix that bench.py publishes.

TEST INFRASTRUCTURE. The reference benchmarks every aligner class over alignments-per-batch x genome size
(cudaaligner/benchmarks/main.cpp:69-143 BM_SingleBatchAlignment, registered :150-168 for AlignerGlobalUkkonen,
AlignerGlobalMyers, AlignerGlobalMyersBanded, AlignerGlobalHirschbergMyers). bench.py runs six cells of that matrix:
all four classes at 1024 pairs x 2048 bases, Ukkonen and Hirschberg + Myers at 256 pairs x 8192 bases. This script runs
each class's oracle (oracle/global_oracle.c, oracle/aligner_oracle.c, oracle/hirschberg_oracle.c) over exactly those pairs
and commits per cell a sha256 over every pair's (status, state sequence) and the sum of the edit distances:

  aligner_matrix_goldens.json   {"<algorithm>/<pairs>x<length>": {"states_sha256", "edit_distance_sum", "states_total", ...}}

  python tests/golden/make_aligner_matrix_goldens.py [--procs N]
"""
import argparse
import importlib.util
import json
import multiprocessing as mp
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

_spec = importlib.util.spec_from_file_location("make_default_aligner_goldens", os.path.join(HERE, "make_default_aligner_goldens.py"))
base = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(base)

# (algorithm, pairs, length); "myers_banded" = create_aligner(max_bandwidth = 1024), the others the fixed-stride classes
CELLS = [("ukkonen", 1024, 2048), ("myers", 1024, 2048), ("myers_banded", 1024, 2048), ("hirschberg_myers", 1024, 2048),
         ("ukkonen", 256, 8192), ("hirschberg_myers", 256, 8192)]
# round 6: the corners of the reference's grid ({32 .. 1024} x {512 .. 65536}, cudaaligner/benchmarks/main.cpp:150-168) for every class
# (CORNER_CELLS: bench.py --sub-configs aligner_grid and the GPU tests; the default bench line keeps the six cells above)
CORNER_CELLS = [(a, 1024, 512) for a in ("ukkonen", "myers", "myers_banded", "hirschberg_myers")]
CORNER_CELLS += [(a, 32, 32768) for a in ("ukkonen", "myers", "myers_banded", "hirschberg_myers")]
CORNER_CELLS += [(a, 32, 65536) for a in ("ukkonen", "myers_banded", "hirschberg_myers", "myers")]
BANDED_MAX_BANDWIDTH = 1024


def cell_key(algorithm, n, size):
    return "%s/%dx%d" % (algorithm, n, size)


def states_to_result(runs_or_states):
    return runs_or_states


def _one(job):
    import oracle_aligner as A
    algorithm, q, t, size = job
    if algorithm == "ukkonen":
        ref = A.ukkonen(q, t, 100)
    elif algorithm == "myers":
        ref = A.myers_full(q, t)
    elif algorithm == "myers_banded":
        ref = A.align(q, t, BANDED_MAX_BANDWIDTH)
        states = [o for o, c in ref["runs"] for _ in range(c)]
        return base.pair_record(ref["status"], states), ref["edit_distance"], len(states)
    else:
        ref = A.hirschberg(q, t, size)
    return base.pair_record(ref["status"], ref["states"]), ref["edit_distance"], len(ref["states"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--only-missing", action="store_true", help="keep the cells the file already holds (the oracle is deterministic)")
    args = ap.parse_args()
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    out = {}
    path = os.path.join(HERE, "aligner_matrix_goldens.json")
    if args.only_missing and os.path.exists(path):
        out = json.load(open(path))
    with mp.get_context("fork").Pool(args.procs) as pool:
        for algorithm, n, size in CELLS + CORNER_CELLS:
            if cell_key(algorithm, n, size) in out:
                continue
            pairs = base.shape_pairs(n, size)
            res = pool.map(_one, [(algorithm, q, t, size) for q, t in pairs], chunksize=max(1, n // (args.procs * 8)))
            out[cell_key(algorithm, n, size)] = {"algorithm": algorithm, "pairs": n, "length": size,
                                                 "states_sha256": base.digest(r for r, _, _ in res),
                                                 "edit_distance_sum": int(sum(e for _, e, _ in res)),
                                                 "states_total": int(sum(s for _, _, s in res))}
            print(cell_key(algorithm, n, size), out[cell_key(algorithm, n, size)], flush=True)
            with open(path, "w") as f:  # (after every cell: the long ones take minutes)
                json.dump(out, f, indent=1, sort_keys=True)
                f.write("\n")


if __name__ == "__main__":
    main()
