#!/usr/bin/env python3
"""The 1024 synthetic windows of BASELINE configs[2] (seeds 1000 + w) in cudapoa's window text format (count line, then the reads):
input of tools/fill_probe and of the cudapoa CLI.   python tools/dump_config3_windows.py out.txt [windows=1024]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomeworks_amd import synthetic  # noqa: E402

n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
with open(sys.argv[1], "w") as f:
    for w in range(n):
        reads = synthetic.generate_window(1000 + w)
        f.write("%d\n" % len(reads))
        for r in reads:
            f.write(r.decode() + "\n")
