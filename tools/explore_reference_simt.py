"""Sweeps seeds over tests/test_reference_simt.py's fresh-window and fresh-pair comparisons (oracle vs the reference itself on the
SIMT emulator): python tools/explore_reference_simt.py <first seed> <count>. Prints the seeds that fail."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(first, first + count):
    env = dict(os.environ, GW_SIMT_SEED=str(seed))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_reference_simt.py"), "-x", "-q", "-k", "fresh"],
                       env=env, capture_output=True, text=True, cwd=ROOT)
    print(seed, "ok" if r.returncode == 0 else "FAIL", flush=True)
    if r.returncode != 0:
        bad.append(seed)
        print(r.stdout[-1500:])
print("failing seeds:", bad)
