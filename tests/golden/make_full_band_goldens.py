#!/usr/bin/env python3
"""Oracle golden of the 17th cell of the band-mode table: FULL band on all 1024 config-3 windows.

TEST INFRASTRUCTURE. The reference's two POA benchmarks run `BatchConfig(1024, 200)` = full band
(cudapoa/benchmarks/single_batch.hpp:52, multi_batch.hpp:49; kernel cudapoa_nw.cuh:149-454). This script runs the CPU
oracle (oracle/poa_oracle.c) with exactly that BatchConfig over the 1024 metric windows (seeds 1000..2023) and commits

  full_band_goldens.npz    fingerprint[window] (uint64, same text as make_band_mode_goldens.fingerprint), cells[window],
                           status[window]
  full_band_goldens.json   total cells, status histogram, sha256 over the fingerprints (what bench.py compares)

  python tests/golden/make_full_band_goldens.py [--procs N]

tests/test_gpu_poa.py::test_full_band_benchmark_shape_equals_the_golden compares all 1024 windows on the GPU;
tests/test_config_goldens.py re-runs the oracle on a sample on CPU.
"""
import argparse
import importlib.util
import json
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

_spec = importlib.util.spec_from_file_location("make_band_mode_goldens", os.path.join(HERE, "make_band_mode_goldens.py"))
band = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(band)

MAX_SEQ, MAX_SEQS, FIRST_SEED = 1024, 200, 1000


def full_band_cfg():
    """BatchConfig(1024, 200): full band, matrix_sequence_dimension 1024, 3072 nodes (cudapoa/src/batch.cu:34-60) =
    CudaPoaBatch(200, 1024, mem, band_mode="full_band", max_nodes_per_graph=3072, matrix_sequence_dimension=1024)."""
    import oracle_poa as O
    cfg = O.make_cfg(MAX_SEQ, MAX_SEQS, 256, 0)
    cfg.max_nodes_per_graph = 3 * MAX_SEQ
    cfg.matrix_sequence_dimension = MAX_SEQ
    cfg.max_banded_pred_distance = 512
    O.lib().poa_cfg_select_types(cfg)
    return cfg


def _chunk(ids):
    import oracle_poa as O
    from genomeworks_amd import synthetic
    out = []
    with O.Workspace(full_band_cfg()) as ws:
        for w in ids:
            reads = [r.decode() for r in synthetic.generate_window(FIRST_SEED + w)]
            ref = ws.process(reads)
            out.append((w, ref["status"], ref["cells"], band.fingerprint(ref["status"], ref.get("consensus", ""), ref.get("coverage", []))))
        overflow = ws.overflow_events()
    return out, overflow


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--windows", type=int, default=1024)
    args = ap.parse_args()
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    n = args.windows
    jobs = [list(range(n))[k::args.procs * 4] for k in range(args.procs * 4)]
    fp, cells, status = np.zeros(n, np.uint64), np.zeros(n, np.int64), np.zeros(n, np.int16)
    overflow = 0
    with mp.get_context("fork").Pool(args.procs) as pool:
        for rows, ov in pool.imap_unordered(_chunk, [j for j in jobs if j], chunksize=1):
            overflow += ov
            for w, st, c, f in rows:
                fp[w], cells[w], status[w] = f, c, st
    np.savez_compressed(os.path.join(HERE, "full_band_goldens.npz"), fingerprint=fp, cells=cells, status=status)
    summary = {"windows": n, "first_seed": FIRST_SEED, "batch_config": "BatchConfig(1024, 200): full_band",
               "cells": int(cells.sum()), "fingerprint_sha256": band.cell_digest(fp),
               "statuses": {str(int(s)): int((status == s).sum()) for s in sorted(set(status.tolist()))},
               "oracle_int16_overflow_events": int(overflow)}
    print(summary)
    with open(os.path.join(HERE, "full_band_goldens.json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
