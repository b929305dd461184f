#!/usr/bin/env python3
"""Oracle goldens for every cell of the band-mode x band-width table.

TEST INFRASTRUCTURE. The reference accepts every banded mode at every band width that is a multiple of 128
(cudapoa/src/batch.cu:34-70; cudapoa_nw_banded.cuh:177-557 and cudapoa_nw_tb_banded.cuh:264-643 are width-agnostic).
This script runs the CPU oracle (oracle/poa_oracle.c) over ALL 1024 config-3 windows (seeds 1000..2023, the metric batch)
for the four banded modes at band widths 128 / 256 / 384 / 512 and commits

  band_mode_goldens.npz    fingerprint[mode, width, window] (uint64: first 8 bytes of the sha256 of
                           "<status> <consensus> <coverage,...>"), cells[mode, width, window], status[mode, width, window]
  band_mode_goldens.json   per cell: total cells, status histogram, sha256 over the fingerprints (what bench.py compares)

  python tests/golden/make_band_mode_goldens.py [--procs N] [--windows 1024]

tests/test_gpu_poa.py::test_band_mode_table_equals_the_golden compares every window of every cell on the GPU;
tests/test_config_goldens.py re-runs the oracle on a sample of every cell on CPU.
"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

MODES = ["static_band", "adaptive_band", "static_band_traceback", "adaptive_band_traceback"]
MODE_ID = {"static_band": 1, "adaptive_band": 2, "static_band_traceback": 3, "adaptive_band_traceback": 4}
WIDTHS = [128, 256, 384, 512]
FIRST_SEED, MAX_SEQ, MAX_SEQS = 1000, 1024, 32


def cell_cfg(mode, width):
    """The BatchConfig of CudaPoaBatch(32, 1024, mem, band_mode=mode, alignment_band_width=width, max_nodes_per_graph=3072)
    (explicit constructor, cudapoa/src/batch.cu:34-70)."""
    import oracle_poa as O
    cfg = O.make_cfg(MAX_SEQ, MAX_SEQS, width, MODE_ID[mode])
    cfg.max_nodes_per_graph = 3 * MAX_SEQ
    cfg.matrix_sequence_dimension = (width + 8) if mode.startswith("static") else 2 * (width + 8)
    cfg.max_banded_pred_distance = 2 * width
    O.lib().poa_cfg_select_types(cfg)
    return cfg


def fingerprint(status, consensus, coverage):
    text = "%d %s %s" % (int(status), consensus if int(status) == 0 else "", ",".join(str(int(c)) for c in coverage) if int(status) == 0 else "")
    return int.from_bytes(hashlib.sha256(text.encode()).digest()[:8], "little")


def _chunk(job):
    import oracle_poa as O
    from genomeworks_amd import synthetic
    mode, width, ids = job
    out = []
    with O.Workspace(cell_cfg(mode, width)) as ws:
        for w in ids:
            reads = [r.decode() for r in synthetic.generate_window(FIRST_SEED + w)]
            ref = ws.process(reads)
            out.append((w, ref["status"], ref["cells"], fingerprint(ref["status"], ref.get("consensus", ""), ref.get("coverage", []))))
        overflow = ws.overflow_events()
    return mode, width, out, overflow


def cell_digest(fp_row):
    return hashlib.sha256(np.ascontiguousarray(fp_row, dtype=np.uint64).tobytes()).hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--windows", type=int, default=1024)
    args = ap.parse_args()
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    n = args.windows
    jobs = []
    for mode in MODES:
        for width in WIDTHS:
            ids = list(range(n))
            for k in range(args.procs * 2):
                part = ids[k::args.procs * 2]
                if part:
                    jobs.append((mode, width, part))
    fp = np.zeros((len(MODES), len(WIDTHS), n), np.uint64)
    cells = np.zeros((len(MODES), len(WIDTHS), n), np.int64)
    status = np.zeros((len(MODES), len(WIDTHS), n), np.int16)
    overflow = {}
    with mp.get_context("fork").Pool(args.procs) as pool:
        for mode, width, rows, ov in pool.imap_unordered(_chunk, jobs, chunksize=1):
            mi, wi = MODES.index(mode), WIDTHS.index(width)
            overflow[(mode, width)] = overflow.get((mode, width), 0) + ov
            for w, st, c, f in rows:
                fp[mi, wi, w], cells[mi, wi, w], status[mi, wi, w] = f, c, st
    np.savez_compressed(os.path.join(HERE, "band_mode_goldens.npz"), fingerprint=fp, cells=cells, status=status)
    summary = {"modes": MODES, "widths": WIDTHS, "windows": n, "first_seed": FIRST_SEED, "cells": {}}
    for mi, mode in enumerate(MODES):
        for wi, width in enumerate(WIDTHS):
            st = status[mi, wi]
            summary["cells"]["%s/%d" % (mode, width)] = {
                "cells": int(cells[mi, wi].sum()), "fingerprint_sha256": cell_digest(fp[mi, wi]),
                "statuses": {str(int(s)): int((st == s).sum()) for s in sorted(set(st.tolist()))},
                "oracle_int16_overflow_events": int(overflow[(mode, width)])}
            print(mode, width, summary["cells"]["%s/%d" % (mode, width)], flush=True)
    with open(os.path.join(HERE, "band_mode_goldens.json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
