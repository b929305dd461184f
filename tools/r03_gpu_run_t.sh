#!/bin/bash
# round 3, GPU call T: configs[1] with six lanes per pair (ten pairs per wavefront: one wavefront per SIMD) against eight
set -u
TAG=${1:-r03t}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_aligner.py tests/test_gpu_aligner_vectors.py tests/test_gpu_config_goldens.py -m gpu -q -k "not config4 and not long" 2>&1 | tail -8 ) > $OUT/pytest.log; tail -4 $OUT/pytest.log
for lanes in 6 8 6 8; do
  echo "lanes $lanes: $(GWHIP_MYERS_GROUP_LANES=$lanes timeout 300 python tools/bench_aligner.py 20000 2> $OUT/err_$lanes.txt | head -1 | cut -c1-420)" >> $OUT/group_lanes.txt
done
echo "default: $(timeout 300 python tools/bench_aligner.py 20000 2>/dev/null | head -1 | cut -c1-420)" >> $OUT/group_lanes.txt
cat $OUT/group_lanes.txt
