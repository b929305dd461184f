#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel, mean counter value per dispatch (what we commit under profiles/)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
# a counter that several passes collect (SQ_WAVE_CYCLES is in four of them, for ratios) is taken from the FIRST pass that has it
# (file order): its dispatch count then equals the other counters', which the per-set sums of tools/pmc_profile.py rely on
owner = {}
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if not any(t in k for t in ("poa_", "myers", "hirschberg", "ukkonen", "cal_")):
            continue
        if owner.setdefault(row["Counter_Name"], f) != f:
            continue
        acc[k.split("(")[0][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("kernel,counter,mean_per_dispatch,dispatches")
for k in sorted(acc):
    for c in sorted(acc[k]):
        v = acc[k][c]
        print('"%s",%s,%.6g,%d' % (k, c, sum(v) / len(v), len(v)))
