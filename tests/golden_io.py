"""Readers of the committed config goldens (tests/golden/make_config_goldens.py) -- TEST INFRASTRUCTURE."""
import gzip
import hashlib
import importlib.util
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

_spec = importlib.util.spec_from_file_location("make_config_goldens", os.path.join(GOLDEN, "make_config_goldens.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)

run_fingerprints = gen.run_fingerprints
block_digests = gen.block_digests


def summary():
    with open(os.path.join(GOLDEN, "config_goldens.json")) as f:
        return json.load(f)


def config3_windows():
    """-> (list of dict(status, cells, consensus, coverage) by window, sha256 of the text)"""
    with gzip.open(os.path.join(GOLDEN, "config3_windows.txt.gz"), "rb") as f:
        text = f.read().decode()
    rows = []
    for line in text.splitlines():
        w, status, cells, consensus, coverage = (line.split(" ") + ["", ""])[:5]
        assert int(w) == len(rows)
        rows.append(dict(status=int(status), cells=int(cells), consensus=consensus,
                         coverage=[int(c) for c in coverage.split(",")] if coverage else []))
    return rows, hashlib.sha256(text.encode()).hexdigest()


def config2_pairs():
    return dict(np.load(os.path.join(GOLDEN, "config2_pairs.npz")))


def config5_pairs():
    d = dict(np.load(os.path.join(GOLDEN, "config5_pairs.npz")))
    d["optimal"] = np.unpackbits(d["optimal"])[:len(d["edit_distance"])]
    return d


def edit_distances(offsets, ops, counts):
    """Per-alignment edit distance from the run-length form (every run that is not a match)."""
    w = np.where(np.asarray(ops) != 0, np.asarray(counts, np.int64), 0)
    c = np.concatenate([[0], np.cumsum(w)])
    offsets = np.asarray(offsets, np.int64)
    return c[offsets[1:]] - c[offsets[:-1]]


# ---- band mode x band width table (tests/golden/make_band_mode_goldens.py) ----
_bspec = importlib.util.spec_from_file_location("make_band_mode_goldens", os.path.join(GOLDEN, "make_band_mode_goldens.py"))
band_gen = importlib.util.module_from_spec(_bspec)
_bspec.loader.exec_module(band_gen)


def band_mode_summary():
    with open(os.path.join(GOLDEN, "band_mode_goldens.json")) as f:
        return json.load(f)


def band_mode_goldens():
    """-> dict(fingerprint[mode, width, window] uint64, cells[...], status[...]) in the order of summary modes / widths"""
    return dict(np.load(os.path.join(GOLDEN, "band_mode_goldens.npz")))


def band_mode_fingerprints(consensus, coverage, status):
    """The golden's per-window fingerprint from a batch's get_consensus() output."""
    return np.array([band_gen.fingerprint(int(status[i]), consensus[i], coverage[i]) for i in range(len(status))], np.uint64)


# ---- the 17th cell: full band, BatchConfig(1024, 200) (tests/golden/make_full_band_goldens.py) ----
_fspec = importlib.util.spec_from_file_location("make_full_band_goldens", os.path.join(GOLDEN, "make_full_band_goldens.py"))
full_gen = importlib.util.module_from_spec(_fspec)
_fspec.loader.exec_module(full_gen)


def full_band_summary():
    with open(os.path.join(GOLDEN, "full_band_goldens.json")) as f:
        return json.load(f)


def full_band_goldens():
    """-> dict(fingerprint[window] uint64, cells[window], status[window])"""
    return dict(np.load(os.path.join(GOLDEN, "full_band_goldens.npz")))


# ---- default aligner on the benchmark shapes (tests/golden/make_default_aligner_goldens.py) ----
_aspec = importlib.util.spec_from_file_location("make_default_aligner_goldens", os.path.join(GOLDEN, "make_default_aligner_goldens.py"))
aligner_gen = importlib.util.module_from_spec(_aspec)
_aspec.loader.exec_module(aligner_gen)


def default_aligner_goldens():
    with open(os.path.join(GOLDEN, "default_aligner_goldens.json")) as f:
        return json.load(f)


# ---- cells of the aligner benchmark matrix (tests/golden/make_aligner_matrix_goldens.py) ----
_mspec = importlib.util.spec_from_file_location("make_aligner_matrix_goldens", os.path.join(GOLDEN, "make_aligner_matrix_goldens.py"))
matrix_gen = importlib.util.module_from_spec(_mspec)
_mspec.loader.exec_module(matrix_gen)


def aligner_matrix_goldens():
    with open(os.path.join(GOLDEN, "aligner_matrix_goldens.json")) as f:
        return json.load(f)
