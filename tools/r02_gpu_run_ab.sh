#!/bin/bash
# A/B on one box: the kernel library of another commit against the current one. genomeworks_amd/lib_old/libgwhip.so is not
# kept in the tree: build it from a worktree of that commit (`git worktree add /tmp/old <commit>`, then
# `python -c "from genomeworks_amd import build; build.build_kernels()"` there) and copy it in before the gpurun call.
set -u
TAG=${1:-r02ab}
mkdir -p gpurun_out/${TAG}
export TMPDIR=/tmp
cp genomeworks_amd/lib/libgwhip.so /tmp/libgwhip_new.so
for which in new old new old; do
  if [ $which = old ]; then cp genomeworks_amd/lib_old/libgwhip.so genomeworks_amd/lib/libgwhip.so; else cp /tmp/libgwhip_new.so genomeworks_amd/lib/libgwhip.so; fi
  echo "$which heaviest: $(timeout 300 python tools/profile_long_read.py 389 1 2>&1 | tail -1)" >> gpurun_out/${TAG}/ab.txt
  ( timeout 600 python bench.py --sub-configs long_reads --no-cpu-baseline > gpurun_out/${TAG}/bench_${which}.json 2> gpurun_out/${TAG}/bench_${which}.err )
  python - gpurun_out/${TAG}/bench_${which}.json >> gpurun_out/${TAG}/ab.txt <<'PY'
import json,sys
d=json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][0]); v=d['sub_records']['configs[3]']
print("   bench", v['value'], v['unit'], v['ms'], "ms", "golden", v['windows_equal_to_oracle_golden'])
PY
done
cp /tmp/libgwhip_new.so genomeworks_amd/lib/libgwhip.so
cat gpurun_out/${TAG}/ab.txt | cut -c1-700
