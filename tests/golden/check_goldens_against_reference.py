"""The committed oracle goldens of the BASELINE configs against THE REFERENCE ITSELF (its cudapoa / cudaaligner libraries on the SIMT
emulator of oracle/simt: tests/ref_cudapoa.py, tests/ref_cudaaligner.py). The emulator runs one lane at a time, so this is a
sample, taken on all cores and recorded in tests/golden/reference_simt_config_check.json:
  configs[2] (the metric): windows of the 1024 (about half a minute each) -- consensus, coverage and status of the golden row;
  configs[1]: pairs of the 10 000 -- CIGAR fingerprint, optimality flag and edit distance of the golden;
  configs[4]: whole 1024-pair blocks of the 1 000 000 -- block digest, flags and edit distances.
usage: python tests/golden/check_goldens_against_reference.py [windows=256] [pairs2=10000] [blocks5=64] [procs=all]"""
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
OUT = os.path.join(HERE, "reference_simt_config_check.json")


def check_windows(ids):
    import golden_io as G
    import ref_cudapoa as R
    from genomeworks_amd import synthetic
    rows, _ = G.config3_windows()
    s = G.summary()["config3"]
    bad = []
    for w in ids:
        reads = [r.decode() for r in synthetic.generate_window(s["first_seed"] + w)]
        with R.RefBatch(s["max_seq"], s["max_seqs"], s["band"], s["band_mode"]) as b:
            st, _ = b.add_poa_group(reads)
            b.generate_poa()
            ref = b.get_consensus()[0]
        if st != 0 or (ref["status"], ref["consensus"], list(ref["coverage"])) != (rows[w]["status"], rows[w]["consensus"], list(rows[w]["coverage"])):
            bad.append(w)
    return "config3", list(ids), bad


def _pair_arrays(res):
    offs, ops, cnts = [0], [], []
    for r in res:
        st = r["states"]
        i = 0
        while i < len(st):
            j = i
            while j < len(st) and st[j] == st[i]:
                j += 1
            ops.append(st[i])
            cnts.append(j - i)
            i = j
        offs.append(len(ops))
    return np.array(offs, np.int64), np.array(ops, np.int8), np.array(cnts, np.int32)


def check_pairs(args):
    name, lo, hi = args
    import golden_io as G
    import ref_cudaaligner as RA
    c = G.gen.CONFIG2 if name == "config2" else G.gen.CONFIG5
    g = G.config2_pairs() if name == "config2" else G.config5_pairs()
    pairs = G.gen.pairs_of(name)[lo:hi]
    res = RA.align(pairs, "banded", max_bandwidth=c["max_bandwidth"])
    ok = all(r["add_status"] == 0 and r["status"] == 0 for r in res)
    offs, ops, cnts = _pair_arrays(res)
    fp = G.run_fingerprints(offs, ops, cnts)
    ok = ok and bool((np.array([1 if r["optimal"] else 0 for r in res]) == g["optimal"][lo:hi]).all())
    ok = ok and bool((G.edit_distances(offs, ops, cnts) == g["edit_distance"][lo:hi]).all())
    if name == "config2":
        ok = ok and bool((fp == g["fingerprint"][lo:hi]).all())
    else:
        block = lo // 1024
        ok = ok and hashlib.sha256(np.ascontiguousarray(fp).tobytes()).hexdigest()[:32] == str(g["block_sha"][block])
    return name, [lo, hi], [] if ok else [[lo, hi]]


def main():
    n_windows = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n_pairs2 = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    n_blocks5 = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    procs = int(sys.argv[4]) if len(sys.argv) > 4 else os.cpu_count()
    t0 = time.time()
    windows = [int(w) for w in np.linspace(0, 1023, n_windows).round()] if n_windows < 1024 else list(range(1024))
    windows = sorted(set(windows))
    jobs = [("w", windows[i::procs * 4]) for i in range(procs * 4)]
    jobs += [("p", ("config2", lo, min(10000, lo + 250))) for lo in range(0, n_pairs2, 250)]
    blocks5 = sorted(set(int(b) for b in np.linspace(0, 975, n_blocks5).round()))
    jobs += [("p", ("config5", b * 1024, b * 1024 + 1024)) for b in blocks5]
    with mp.get_context("fork").Pool(procs) as pool:
        results = pool.map(_run, jobs, chunksize=1)
    out = {"generator": "tests/golden/check_goldens_against_reference.py", "config3": {"windows_checked": [], "windows_differing": []},
           "config2": {"pair_ranges_checked": [], "ranges_differing": []}, "config5": {"pair_ranges_checked": [], "ranges_differing": []}}
    for name, checked, bad in results:
        if name == "config3":
            out[name]["windows_checked"] += checked
            out[name]["windows_differing"] += bad
        else:
            out[name]["pair_ranges_checked"].append(checked)
            out[name]["ranges_differing"] += bad
    out["config3"]["windows_checked"].sort()
    out["config2"]["pair_ranges_checked"].sort()
    out["config5"]["pair_ranges_checked"].sort()
    out["seconds"] = round(time.time() - t0)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("windows", len(out["config3"]["windows_checked"]), "differing", out["config3"]["windows_differing"])
    print("config2 ranges", len(out["config2"]["pair_ranges_checked"]), "differing", out["config2"]["ranges_differing"])
    print("config5 ranges", len(out["config5"]["pair_ranges_checked"]), "differing", out["config5"]["ranges_differing"])


def _run(job):
    kind, arg = job
    return check_windows(arg) if kind == "w" else check_pairs(arg)


if __name__ == "__main__":
    main()
