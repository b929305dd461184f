"""Kernel / copy timeline of the last bursts of a rocprofv3 --kernel-trace --memory-copy-trace run (csv).
usage: python tools/gpu_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv> [bursts=1] [gap_ms=20]"""
import csv, glob, sys
d = sys.argv[1]; bursts = int(sys.argv[2]) if len(sys.argv) > 2 else 1; gap = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
ev = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s" % (r["Queue_Id"], r["Kernel_Name"].replace("gwhip::myers::", "")[:44])))
for f in glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r["Direction"].replace("MEMORY_COPY_", "")))
ev.sort()
groups = [[ev[0]]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in groups[-1]) > gap * 1e6: groups.append([])
    groups[-1].append(e)
for g in groups[-bursts:]:
    t0 = g[0][0]
    print("--- burst of %d events, %.3f ms" % (len(g), (max(x[1] for x in g) - t0) / 1e6))
    for s, e, n in g: print("%8.3f %7.3f %s" % ((s - t0) / 1e6, (e - s) / 1e6, n))
