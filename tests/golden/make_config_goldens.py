#!/usr/bin/env python3
"""Oracle goldens for EVERY unit of the BASELINE.json configs (SURVEY.md 8(d) "Config 1" replacement).

TEST INFRASTRUCTURE. Runs the CPU oracle (oracle/*.c, the restatement of the reference's kernels) on the seeded
synthetic inputs the benchmarks and GPU tests use and commits what it produced:

  config3_windows.txt.gz   configs[2] (the metric config): all 1024 windows (seeds 1000..2023), one line per window
                           "<w> <status> <cells> <consensus> <coverage,...>" -- full text
  config2_pairs.npz        configs[1]: 10 000 pairs x 1 kbp (seed 1, <=33 sub/ins/del, max_bandwidth 1024):
                           per-pair fingerprint of the run-length CIGAR, edit distance, optimal flag, band cells
  config5_pairs.npz        configs[4]: 1 000 000 pairs x 150 bp (seed 3, <=2 sub / <=1 ins / <=1 del,
                           max_bandwidth 150): per-pair edit distance (uint8) + sha256 of the fingerprints per
                           block of 1024 pairs
  config4_long_reads.json  configs[3]: the long-read MSA set of tools/bench_long_read_msa.py -- per window its
                           BatchConfig, status and the sha256 of its MSA rows
  config_goldens.json      sha256 of each file's payload + the totals the bench line must reproduce (cells, ...)

  python tests/golden/make_config_goldens.py [--only config3,config2,config5,config4] [--procs N]

The GPU tests (tests/test_gpu_config_goldens.py) compare every unit of the HIP path's output with these files; the
CPU suite re-runs the oracle on a sample of each file to catch a drifting oracle.
"""
import argparse
import gzip
import hashlib
import io
import json
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIG3 = dict(windows=1024, first_seed=1000, max_seq=1024, max_seqs=32, band=256, band_mode=1)
CONFIG2 = dict(seed=1, pairs=10000, length=1000, mut=33, ins=33, dele=33, max_bandwidth=1024)
CONFIG5 = dict(seed=3, pairs=1000000, length=150, mut=2, ins=1, dele=1, max_bandwidth=150, block=1024)


# ---------------------------------------------------------------------------------------------------------------
# fingerprints shared with the GPU tests
# ---------------------------------------------------------------------------------------------------------------
def run_fingerprints(offsets, ops, counts):
    """One uint64 per alignment from its run-length CIGAR in forward order: position-weighted sum over the runs of
    (op + 1) * 1000003 + count (wrapping arithmetic). offsets[n + 1] index ops / counts."""
    offsets = np.asarray(offsets, np.int64)
    n = len(offsets) - 1
    ops = np.asarray(ops).astype(np.uint64)
    counts = np.asarray(counts).astype(np.uint64)
    out = np.zeros(n, np.uint64)
    if len(ops) == 0:
        return out
    starts = offsets[:-1]
    local = (np.arange(len(ops), dtype=np.int64) - np.repeat(starts, np.diff(offsets))).astype(np.uint64)
    with np.errstate(over="ignore"):
        term = ((ops + np.uint64(1)) * np.uint64(1000003) + counts) * ((local + np.uint64(1)) * np.uint64(2654435761))
        term ^= term >> np.uint64(29)
        csum = np.concatenate([[np.uint64(0)], np.cumsum(term, dtype=np.uint64)])
        out = csum[offsets[1:]] - csum[offsets[:-1]]
    return out


def block_digests(fp, block):
    return [hashlib.sha256(np.ascontiguousarray(fp[i:i + block]).tobytes()).hexdigest()[:32]
            for i in range(0, len(fp), block)]


def window_line(w, status, cells, consensus, coverage):
    return "%d %d %d %s %s" % (w, status, cells, consensus, ",".join(str(int(c)) for c in coverage))


# ---------------------------------------------------------------------------------------------------------------
# workers
# ---------------------------------------------------------------------------------------------------------------
def _config3_chunk(ws_ids):
    import oracle_poa as O
    from genomeworks_amd import synthetic
    cfg = O.make_cfg(CONFIG3["max_seq"], CONFIG3["max_seqs"], CONFIG3["band"], CONFIG3["band_mode"])
    out = []
    with O.Workspace(cfg) as ws:
        for w in ws_ids:
            reads = [r.decode() for r in synthetic.generate_window(CONFIG3["first_seed"] + w)]
            ref = ws.process(reads)
            out.append((w, ref["status"], ref["cells"], ref.get("consensus", ""), list(ref.get("coverage", []))))
        assert ws.overflow_events() == 0
    return out


def _pairs_chunk(args):
    import oracle_aligner as A
    name, lo, hi = args
    c = CONFIG2 if name == "config2" else CONFIG5
    pairs = _PAIRS[name][lo:hi]
    offs, ops, cnts, ed, opt, cells, status = [0], [], [], [], [], [], []
    for q, t in pairs:
        r = A.align(q, t, c["max_bandwidth"])
        status.append(r["status"])
        for o, k in r["runs"]:
            ops.append(o)
            cnts.append(k)
        offs.append(len(ops))
        ed.append(r["edit_distance"])
        opt.append(1 if r["optimal"] else 0)
        cells.append(r["cells"])
    fp = run_fingerprints(offs, np.array(ops, np.int8), np.array(cnts, np.int32))
    return lo, fp, np.array(ed, np.int32), np.array(opt, np.uint8), np.array(cells, np.int64), np.array(status, np.int8)


_PAIRS = {}


def pairs_of(name):
    from genomeworks_amd import synthetic
    c = CONFIG2 if name == "config2" else CONFIG5
    return synthetic.generate_pairs(c["seed"], c["pairs"], c["length"], c["mut"], c["ins"], c["dele"])


def run_pairs(name, procs):
    _PAIRS[name] = pairs_of(name)
    n = len(_PAIRS[name])
    step = max(256, n // (procs * 8))
    jobs = [(name, lo, min(n, lo + step)) for lo in range(0, n, step)]
    with mp.get_context("fork").Pool(procs) as pool:
        parts = sorted(pool.map(_pairs_chunk, jobs, chunksize=1), key=lambda p: p[0])
    return [np.concatenate([p[k] for p in parts]) for k in range(1, 6)]


# ---------------------------------------------------------------------------------------------------------------
def save_gz(path, text):
    buf = io.BytesIO()
    with gzip.GzipFile(fileobj=buf, mode="wb", mtime=0, compresslevel=9) as f:
        f.write(text.encode())
    with open(path, "wb") as f:
        f.write(buf.getvalue())
    return hashlib.sha256(text.encode()).hexdigest()


def make_config3(procs, summary):
    ids = list(range(CONFIG3["windows"]))
    chunks = [ids[i::procs * 4] for i in range(procs * 4)]
    with mp.get_context("fork").Pool(procs) as pool:
        rows = sorted(r for part in pool.map(_config3_chunk, chunks, chunksize=1) for r in part)
    text = "\n".join(window_line(*r) for r in rows) + "\n"
    sha = save_gz(os.path.join(HERE, "config3_windows.txt.gz"), text)
    summary["config3"] = dict(CONFIG3, sha256=sha, cells=int(sum(r[2] for r in rows)),
                              statuses={str(s): sum(1 for r in rows if r[1] == s) for s in sorted({r[1] for r in rows})},
                              consensus_sha256=hashlib.sha256("\n".join(r[3] for r in rows).encode()).hexdigest())
    print("config3: %d windows, %d cells" % (len(rows), summary["config3"]["cells"]))


def make_config2(procs, summary):
    fp, ed, opt, cells, status = run_pairs("config2", procs)
    np.savez_compressed(os.path.join(HERE, "config2_pairs.npz"), fingerprint=fp, edit_distance=ed.astype(np.int16),
                        optimal=opt, cells=cells, status=status)
    summary["config2"] = dict(CONFIG2, fingerprint_sha256=hashlib.sha256(fp.tobytes()).hexdigest(),
                              band_cells=int(cells.sum()), edit_distance_sum=int(ed.sum()), optimal=int(opt.sum()),
                              status_ok=int((status == 0).sum()))
    print("config2:", summary["config2"])


def make_config5(procs, summary):
    fp, ed, opt, cells, status = run_pairs("config5", procs)
    np.savez_compressed(os.path.join(HERE, "config5_pairs.npz"), edit_distance=ed.astype(np.uint8), optimal=np.packbits(opt),
                        block_sha=np.array(block_digests(fp, CONFIG5["block"])))
    summary["config5"] = dict(CONFIG5, fingerprint_sha256=hashlib.sha256(fp.tobytes()).hexdigest(),
                              band_cells=int(cells.sum()), edit_distance_sum=int(ed.sum()), optimal=int(opt.sum()),
                              status_ok=int((status == 0).sum()))
    print("config5:", summary["config5"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="config3,config2,config5")
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    args = ap.parse_args()
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    path = os.path.join(HERE, "config_goldens.json")
    summary = json.load(open(path)) if os.path.exists(path) else {}
    for name in args.only.split(","):
        {"config3": make_config3, "config2": make_config2, "config5": make_config5,
         "config4": lambda p, s: __import__("make_long_read_goldens").make(p, s)}[name](args.procs, summary)
    with open(path, "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
