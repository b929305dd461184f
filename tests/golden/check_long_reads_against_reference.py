"""More WHOLE long-read windows of BASELINE configs[3] against THE REFERENCE ITSELF (TEST INFRASTRUCTURE), incrementally.

tests/golden/check_goldens_against_reference.py confirmed the 245 cheapest of the 598 windows in round 5 and writes its record at
the end of a run. This driver takes the windows that record does not hold yet, cheapest first, runs each through the reference's
own cudapoa library on the SIMT emulator (check_long_read_windows of that script: the class's BatchConfig, storage factor 4, MSA
digest against the committed golden) and rewrites tests/golden/reference_simt_config_check.json after EVERY window, so a run can
be stopped at any time. A window costs about cells / 0.33 M seconds of one core (2-90 minutes).

  python tests/golden/check_long_reads_against_reference.py [procs=6] [hours=6]
"""
import importlib.util
import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
OUT = os.path.join(HERE, "reference_simt_config_check.json")


def _checker():
    spec = importlib.util.spec_from_file_location("check_goldens_against_reference", os.path.join(HERE, "check_goldens_against_reference.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _one(w):
    t0 = time.time()
    _name, checked, bad = _checker().check_long_read_windows([w])
    return w, bool(bad), round(time.time() - t0)


def main():
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    hours = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
    with open(OUT) as f:
        record = json.load(f)
    with open(os.path.join(HERE, "config4_long_reads.json")) as f:
        detail = json.load(f)["windows_detail"]
    done = set(record["config4"]["windows_checked"])
    todo = [d for d in sorted(detail, key=lambda d: d["cells"]) if d["w"] not in done]
    # as many of the cheapest as the budget holds (procs cores for `hours`, 0.33 M cells per core-second), with a margin for the last ones
    budget, picked = procs * hours * 3600 * 0.33e6 * 0.9, []
    for d in todo:
        if budget < d["cells"]:
            break
        budget -= d["cells"]
        picked.append(d["w"])
    print("windows to check: %d of the %d open ones" % (len(picked), len(todo)), flush=True)
    t0 = time.time()
    with mp.get_context("fork").Pool(procs) as pool:
        for w, bad, seconds in pool.imap_unordered(_one, picked, chunksize=1):
            with open(OUT) as f:  # (someone else may have written other keys meanwhile)
                record = json.load(f)
            record["config4"]["windows_checked"] = sorted(set(record["config4"]["windows_checked"]) | {w})
            if bad:
                record["config4"]["windows_differing"] = sorted(set(record["config4"]["windows_differing"]) | {w})
            record["seconds"] = int(record.get("seconds", 0)) + seconds
            tmp = OUT + ".tmp"
            with open(tmp, "w") as f:
                json.dump(record, f)
                f.write("\n")
            os.replace(tmp, OUT)
            print("window %d: %s, %d s (%d checked, %.1f h elapsed)" % (w, "DIFFERS" if bad else "equal", seconds,
                                                                         len(record["config4"]["windows_checked"]), (time.time() - t0) / 3600), flush=True)


if __name__ == "__main__":
    main()
