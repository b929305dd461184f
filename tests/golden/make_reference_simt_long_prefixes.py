"""The expensive end of BASELINE configs[3] answered by THE REFERENCE ITSELF (TEST INFRASTRUCTURE).

tests/golden/check_goldens_against_reference.py confirms the 245 cheapest of the 598 long-read windows on the SIMT emulator; a
whole window of the two largest size classes (reads of 7.6-30 kbp, 32-bit scores and ids, the adaptive band growing from 256 to its
1536-column cap) would take hours there. This script takes windows of those two classes that are NOT among the 245 and cuts each
to its first reads (PREFIX_READS): the same kernels, score / id types, band growth and graph sizes, at minutes per window. Every
window goes through the reference's create_batch / add_poa_group / generate_poa / get_msa with its class's BatchConfig (storage
factor 4, as the set is run) and, beside it, through the C oracle; the file records the reference's status and MSA digest and
whether the oracle agreed (it has to: the script fails otherwise).

With four reads the adaptive band stays at its initial 256 columns; with eight it has grown (reruns at doubled width) to an
average of ~780 columns per row and with twelve to ~1300 of its 1536-column cap, so the file holds three sets: many windows of 4
reads, fewer of 8, a few of 12 (5 and 14 minutes each on the emulator).

  python tests/golden/make_reference_simt_long_prefixes.py [procs=6] [per_class0=36] [per_class1=24] [prefix_reads=4] [append]
-> tests/golden/reference_simt_long_prefixes.json (inputs are regenerated from seeds: genomeworks_amd.synthetic.long_read_window)
"""
import importlib.util
import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(HERE, "reference_simt_long_prefixes.json")
PREFIX_READS = 4

_S = {}


def _lr():
    spec = importlib.util.spec_from_file_location("make_long_read_goldens", os.path.join(HERE, "make_long_read_goldens.py"))
    lr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lr)
    return lr


def select(per_class, skip=()):
    """Windows of size classes 0 and 1 that the whole-window check has not covered, spread evenly over the class's lengths."""
    lr = _lr()
    windows, cfgs, groups = lr.plan()
    with open(os.path.join(HERE, "reference_simt_config_check.json")) as f:
        done = set(json.load(f)["config4"]["windows_checked"])
    picks = []
    for k, want in enumerate(per_class):
        rest = sorted((w for w in groups[k] if w not in done and w not in skip), key=lambda w: (max(len(r) for r in windows[w]), w))
        step = len(rest) / float(want)
        chosen = sorted({rest[min(len(rest) - 1, int(i * step + step / 2))] for i in range(want)})
        picks += [(w, k) for w in chosen]
    return windows, cfgs, picks


def _run(job):
    import oracle_poa as O
    import ref_cudapoa as R
    lr = _S["lr"]
    w, k = job
    c = _S["cfgs"][k]
    reads = _S["windows"][w][:PREFIX_READS]
    t0 = time.time()
    with O.Workspace(lr.oracle_cfg(c)) as ws:
        o = ws.process(reads)
    with R.RefBatch(c["max_sequence_size"], c["max_sequences_per_poa"], c["alignment_band_width"], c["band_mode"], storage_factor=4.0, output_mask=2,
                    max_mem=3 << 30) as b:
        st, per_read = b.add_poa_group(reads)
        b.generate_poa()
        ref = b.get_msa()[0]
    row = dict(w=w, cfg=k, reads=len(reads), longest=max(len(r) for r in reads), add_status=st, status=ref["status"],
               msa_sha=lr.msa_digest(ref["msa"]) if ref["status"] == 0 else "", msa_rows=len(ref["msa"]),
               msa_columns=len(ref["msa"][0]) if ref["msa"] else 0, cells=int(o["cells"]),
               oracle_equal=bool(o["status"] == ref["status"] and (ref["status"] != 0 or list(o["msa"]) == list(ref["msa"]))),
               seconds=round(time.time() - t0, 1))
    print(row, flush=True)
    return row


def record_in_config_check(rows):
    """reference_simt_config_check.json lists what of the BASELINE configs the reference itself has answered: the cut-down windows
    go in beside the 245 whole ones, per size class."""
    path = os.path.join(HERE, "reference_simt_config_check.json")
    with open(path) as f:
        check = json.load(f)
    per_class = {}
    for r in rows:
        e = per_class.setdefault(str(r["cfg"]), {"windows": 0, "reads": {}, "cells": 0})
        e["windows"] += 1
        e["reads"][str(r["reads"])] = e["reads"].get(str(r["reads"]), 0) + 1
        e["cells"] += r["cells"]
    check["config4_prefixes"] = {"file": "tests/golden/reference_simt_long_prefixes.json",
                                 "what": "windows of size classes 0 and 1 that are not among config4.windows_checked, cut to their first 4 / 8 / 12 reads",
                                 "windows_checked": sorted(r["w"] for r in rows), "windows_differing": sorted(r["w"] for r in rows if not r["oracle_equal"]),
                                 "per_size_class": per_class}
    with open(path, "w") as f:
        json.dump(check, f)
        f.write("\n")


def main():
    global PREFIX_READS
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    per_class = [int(sys.argv[2]) if len(sys.argv) > 2 else 36, int(sys.argv[3]) if len(sys.argv) > 3 else 24]
    PREFIX_READS = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    append = len(sys.argv) > 5 and sys.argv[5] == "append"
    before = json.load(open(OUT))["windows"] if append else []
    windows, cfgs, picks = select(per_class, skip={r["w"] for r in before})
    _S.update(lr=_lr(), windows=windows, cfgs=cfgs)
    # longest first: the pool's tail is short jobs
    picks.sort(key=lambda j: -sum(len(r) for r in windows[j[0]][:PREFIX_READS]))
    t0 = time.time()
    with mp.get_context("fork").Pool(procs) as pool:
        rows = sorted(before + pool.map(_run, picks, chunksize=1), key=lambda r: (r["reads"], r["w"]))
    out = dict(generator="tests/golden/make_reference_simt_long_prefixes.py", storage_factor=4.0,
               what="the first reads (`reads` of each row) of long-read windows of size classes 0 and 1 (tests/golden/config4_long_reads.json "
                    "batch_configs), answered by the reference's own cudapoa library on the SIMT emulator (oracle/_ref/libref_cudapoa_simt.so)",
               batch_configs=[cfgs[0], cfgs[1]], windows=rows, seconds=int(sum(r["seconds"] for r in rows)))
    with open(OUT, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
        f.write("\n")
    record_in_config_check(rows)
    bad = [r["w"] for r in rows if not r["oracle_equal"]]
    print("windows: %d, oracle differs on: %s, %.0f s" % (len(rows), bad, time.time() - t0))
    if bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
