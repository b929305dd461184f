#!/bin/bash
# Round-2 measurement session on the GPU box: the driver's bench line, rocprofv3 kernel stats of the same command (all
# configs), PMC passes (own runs, metric config only), HBM traffic per launch, phase breakdown of the metric kernel.
# usage (through gpurun): bash tools/measure_round2.sh <tag>
set -u
TAG=${1:-r02_v16}
REPO=$(pwd)
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/stats -- python $REPO/bench.py --no-cpu-baseline > $REPO/$OUT/stats.log 2>&1)
DB=$(find $OUT/stats -name "*.db" | head -1)
echo "db=$DB"
[ -n "$DB" ] && python tools/rocpd_summary.py "$DB" > $OUT/kernel_stats.csv && head -12 $OUT/kernel_stats.csv
bash tools/pmc_passes.sh $OUT/pmc > $OUT/pmc.log 2>&1
python tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.csv
grep -c . $OUT/pmc_summary.csv
python - "$OUT/pmc_summary.csv" "$TAG" > $OUT/pmc_traffic.json <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "poa_window_kernel" in r["kernel"]]
v = {r["counter"]: float(r["mean_per_dispatch"]) for r in rows}
fetch, write = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
out = {"kernel": "poa_window_kernel<int16,int16,static_band>", "windows": 1024,
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 0 --sub-configs none`, "
                 "mean per dispatch (tools/pmc_passes.sh), build " + sys.argv[2],
       "fetch_size_kb": fetch, "write_size_kb": write,
       "correction": "gfx950: FETCH_SIZE counts 128-B read requests as 64 B (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE taken as reported",
       "hbm_bytes_per_launch": None if fetch is None or write is None else (2 * fetch + write) * 1024}
print(json.dumps(out, indent=1))
PY
cat $OUT/pmc_traffic.json
python tools/profile_phases.py 1024 2>/dev/null | tail -1 > $OUT/phase_breakdown.json
rm -rf $OUT/stats/*/*.db.tmp $OUT/pmc/*/*.db 2>/dev/null
find $OUT -name "*.db" -size +20M -delete
du -sh $OUT
