#!/bin/bash
# round 2, first GPU call: parity (all units of the configs), the new bench line, aligner phase ablation, kernel stats
set -u
mkdir -p gpurun_out/r02a
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r02a/pytest.log
( timeout 600 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err ) ; echo "bench rc=$?" >> gpurun_out/r02a/bench.err
for s in 0 1 3 7; do
  ( GWHIP_MYERS_SKIP=$s timeout 300 python tools/bench_aligner.py 200000 2>&1 | sed "s/^/skip=$s /" ) >> gpurun_out/r02a/aligner_ablation.txt
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o r02a -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --sub-configs aligner > /tmp/prof_a.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_a -name "*kernel_stats*" -exec cp {} gpurun_out/r02a/ \;
tail -5 /tmp/prof_a.log > gpurun_out/r02a/prof.log
ls -la gpurun_out/r02a
