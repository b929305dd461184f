set -u
REPO=$(pwd)
mkdir -p gpurun_out/v10
python bench.py --steps 5 --warmup 1 > gpurun_out/v10/bench.json 2> gpurun_out/v10/bench.err
tail -c 1500 gpurun_out/v10/bench.json
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/v10/stats -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/v10/stats.log 2>&1)
DB=$(find gpurun_out/v10/stats -name "*.db" | head -1)
echo "db=$DB"
[ -n "$DB" ] && python tools/rocpd_summary.py "$DB" > gpurun_out/v10/kernel_stats.csv && head -5 gpurun_out/v10/kernel_stats.csv
bash tools/pmc_passes.sh gpurun_out/v10/pmc > gpurun_out/v10/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/v10/pmc > gpurun_out/v10/pmc_summary.csv
grep -c . gpurun_out/v10/pmc_summary.csv
python tools/profile_phases.py 1024 2>/dev/null | tail -1 > gpurun_out/v10/phase_breakdown.json
cut -c1-600 gpurun_out/v10/phase_breakdown.json
rm -rf gpurun_out/v10/stats/*/*.db.tmp
du -sh gpurun_out/v10
