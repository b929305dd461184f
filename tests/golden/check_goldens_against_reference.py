"""The committed oracle goldens of the BASELINE configs against THE REFERENCE ITSELF (its cudapoa / cudaaligner libraries on the SIMT
emulator of oracle/simt: tests/ref_cudapoa.py, tests/ref_cudaaligner.py). The emulator runs one lane at a time, so this is a
sample, taken on all cores and recorded in tests/golden/reference_simt_config_check.json:
  configs[2] (the metric): windows of the 1024 (about half a minute each) -- consensus, coverage and status of the golden row;
  configs[1]: pairs of the 10 000 -- CIGAR fingerprint, optimality flag and edit distance of the golden;
  configs[4]: whole 1024-pair blocks of the 1 000 000 -- block digest, flags and edit distances;
  configs[3]: the long-read windows of the 598 with the fewest cells (from 8 reads of 2.3 kbp: half a minute and up each; 32-bit scores and
              ids, adaptive band with a storage factor of 4) -- status and the digest of the MSA rows.
usage: python tests/golden/check_goldens_against_reference.py [windows=256] [pairs2=10000] [blocks5=64] [procs=all] [long_reads=0]"""
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


def check_long_read_windows(ids):
    import importlib.util
    import ref_cudapoa as R
    spec = importlib.util.spec_from_file_location("make_long_read_goldens", os.path.join(HERE, "make_long_read_goldens.py"))
    lr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lr)
    with open(os.path.join(HERE, "config4_long_reads.json")) as f:
        golden = json.load(f)
    windows, cfgs, _groups = lr.plan()
    bad = []
    for w in ids:
        d = golden["windows_detail"][w]
        assert d["w"] == w
        c = cfgs[d["cfg"]]
        # the size class's BatchConfig through the reference's deriving constructor (storage factor 4: matrix dimension 4 x (band + 8))
        got = R.config(c["max_sequence_size"], c["max_sequences_per_poa"], c["alignment_band_width"], c["band_mode"], 4.0, 3.0, 0)
        if got != c:
            bad.append(w)
            continue
        with R.RefBatch(c["max_sequence_size"], c["max_sequences_per_poa"], c["alignment_band_width"], c["band_mode"], storage_factor=4.0, output_mask=2,
                        max_mem=3 << 30) as b:
            st, _ = b.add_poa_group(windows[w])
            b.generate_poa()
            ref = b.get_msa()[0]
        if st != 0 or ref["status"] != d["status"] or (d["status"] == 0 and lr.msa_digest(ref["msa"]) != d["msa_sha"]):
            bad.append(w)
    return "config4", list(ids), bad


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
    # an existing record is extended: what it already holds is not asked again
    old = json.load(open(OUT)) if os.path.exists(OUT) else None
    done_w = set(old["config3"]["windows_checked"]) if old else set()
    done_2 = set(tuple(r) for r in old["config2"]["pair_ranges_checked"]) if old else set()
    done_5 = set(tuple(r) for r in old["config5"]["pair_ranges_checked"]) if old else set()
    windows = sorted(set(windows) - done_w)
    jobs = [("w", windows[i::procs * 4]) for i in range(procs * 4) if windows[i::procs * 4]]
    jobs += [("p", ("config2", lo, min(10000, lo + 250))) for lo in range(0, n_pairs2, 250) if (lo, min(10000, lo + 250)) not in done_2]
    blocks5 = (list(range(977)) if n_blocks5 >= 977 else sorted(set(int(b) for b in np.linspace(0, 975, n_blocks5).round()))) if n_blocks5 else []
    jobs += [("p", ("config5", b * 1024, min(1000000, b * 1024 + 1024))) for b in blocks5 if (b * 1024, min(1000000, b * 1024 + 1024)) not in done_5]
    n_long = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    done_4 = set(old["config4"]["windows_checked"]) if old and "config4" in old else set()
    if n_long:
        with open(os.path.join(HERE, "config4_long_reads.json")) as f:
            detail = json.load(f)["windows_detail"]
        smallest = [d["w"] for d in sorted(detail, key=lambda d: d["cells"])[:n_long]]  # by the golden's cell count
        jobs += [("l", [w]) for w in smallest if w not in done_4]
    with mp.get_context("fork").Pool(procs) as pool:
        results = pool.map(_run, jobs, chunksize=1)
    out = {"generator": "tests/golden/check_goldens_against_reference.py", "config3": {"windows_checked": [], "windows_differing": []},
           "config2": {"pair_ranges_checked": [], "ranges_differing": []}, "config5": {"pair_ranges_checked": [], "ranges_differing": []},
           "config4": {"windows_checked": [], "windows_differing": []}}
    if old:
        for k in ("config3", "config2", "config5", "config4"):
            for field in out[k]:
                out[k][field] += old.get(k, {}).get(field, [])
    for name, checked, bad in results:
        if name in ("config3", "config4"):
            out[name]["windows_checked"] += checked
            out[name]["windows_differing"] += bad
        else:
            out[name]["pair_ranges_checked"].append(checked)
            out[name]["ranges_differing"] += bad
    out["config3"]["windows_checked"].sort()
    out["config4"]["windows_checked"].sort()
    out["config2"]["pair_ranges_checked"].sort()
    out["config5"]["pair_ranges_checked"].sort()
    out["seconds"] = round(time.time() - t0) + (old.get("seconds", 0) if old else 0)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("windows", len(out["config3"]["windows_checked"]), "differing", out["config3"]["windows_differing"])
    print("config2 ranges", len(out["config2"]["pair_ranges_checked"]), "differing", out["config2"]["ranges_differing"])
    print("config5 ranges", len(out["config5"]["pair_ranges_checked"]), "differing", out["config5"]["ranges_differing"])
    print("long-read windows", out["config4"]["windows_checked"], "differing", out["config4"]["windows_differing"])


def _run(job):
    kind, arg = job
    return check_windows(arg) if kind == "w" else (check_long_read_windows(arg) if kind == "l" else check_pairs(arg))


if __name__ == "__main__":
    main()
