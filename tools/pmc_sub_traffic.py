#!/usr/bin/env python3
"""profiles/r03_pmc_traffic_sub.json from the summary of the PMC passes over the sub-records' kernels
(tools/pmc_passes.sh with SUBS=aligner,default_aligner,long_reads -> tools/pmc_summary.py): HBM bytes per launch of each
sub-record's dominant kernel, FETCH_SIZE x 2 and WRITE_SIZE x 0.97 as calibrated in profiles/r03_pmc_traffic.json.
  python tools/pmc_sub_traffic.py <pmc_sub_summary.csv> <source note> > profiles/r03_pmc_traffic_sub.json"""
import csv
import json
import sys

rows = {}
for r in csv.DictReader(open(sys.argv[1])):
    rows.setdefault(r["kernel"], {})[r["counter"]] = (float(r["mean_per_dispatch"]), int(r["dispatches"]))


def entry(match, what="mean", note=None):
    ks = [k for k in rows if match(k)]
    if not ks:
        return None
    rd = wr = 0.0
    for k in ks:
        f, nf = rows[k].get("FETCH_SIZE", (0.0, 0))
        w, nw = rows[k].get("WRITE_SIZE", (0.0, 0))
        scale_f, scale_w = (nf, nw) if what == "sum" else (1, 1)
        rd += f * 1024 * 2 * scale_f
        wr += w * 1024 * 0.97 * scale_w
    e = {"kernel": " + ".join(k.replace("void ", "") for k in ks), "hbm_bytes": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr)}
    if note:
        e["note"] = note
    return e


out = {"source": (sys.argv[2] if len(sys.argv) > 2 else "tools/pmc_passes.sh with SUBS=aligner,default_aligner,long_reads") +
       "; FETCH_SIZE x 2 and WRITE_SIZE x 0.97 as calibrated in r03_pmc_traffic.json; bytes per launch (mean over the launches of the bench invocation)"}
for key, e in (("configs[1]", entry(lambda k: "myers_banded_group_kernel" in k)),
               ("configs[4]", entry(lambda k: "myers_banded_kernel<true>" in k)),
               ("default_aligner", entry(lambda k: "hirschberg_levels_kernel" in k or "hirschberg_wave_kernel" in k, "mean",
                                         "levels kernel + the depth-first kernel behind it, mean over the launches of the sub-record's shapes (1 .. 2000 pairs)")),
               ("configs[3]", entry(lambda k: "poa_window_kernel<" in k and "2, true, false" in k, "sum",
                                    "sum over the launches of the set's size classes; 1.99e11 cells x 8 B = 1.59 TB algorithmic"))):
    if e:
        out[key] = e
print(json.dumps(out, indent=1))
