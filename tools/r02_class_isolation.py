#!/usr/bin/env python3
"""Experiment: the long-read set's size classes one at a time, and the heaviest class together with each of the others
(where does the time of the concurrent run go?).  python tools/r02_class_isolation.py"""
import importlib.util, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genomeworks_amd import cudapoa
spec = importlib.util.spec_from_file_location("lr", os.path.join(ROOT, "tests", "golden", "make_long_read_goldens.py"))
lr = importlib.util.module_from_spec(spec); spec.loader.exec_module(lr)
windows, cfgs, groups = lr.plan(lr.CONFIG4["windows"])
golden = {d["w"]: d for d in json.load(open(os.path.join(ROOT, "tests", "golden", "config4_long_reads.json")))["windows_detail"]}
base = lr.size_plan(windows)
classes = [list(g) for g in base.groups]
def run(keep, label):
    plan = lr.size_plan(windows)
    plan.keep(keep)
    out = cudapoa.process_windows_size_classes(windows, plan, device=0, memory_budget=lr.CONFIG4["memory_budget_bytes"], output_type="msa", digest=lr.msa_digest)
    cells = sum(golden[w]["cells"] for w in keep)
    print(json.dumps({"run": label, "windows": len(keep), "cells": cells, "ms": round(out["compute_seconds"] * 1e3, 1),
                      "gcups": round(cells / out["compute_seconds"] / 1e9, 2)}), flush=True)
run(sum(classes, []), "warm-up: all")
for k, g in enumerate(classes):
    run(g, "class %d alone (%d windows, <= %d)" % (k, len(g), cfgs[k]["max_sequence_size"]))
run(classes[0] + classes[1], "classes 0+1")
run(classes[0] + classes[2] + classes[3], "classes 0+2+3")
run(sum(classes, []), "all")
heavy = sorted(classes[0], key=lambda w: -golden[w]["cells"])[:16]
run(heavy, "16 heaviest windows")
