#!/bin/bash
# round 3, GPU call L: band 128 through the packed pass: parity, headline (does the second instantiation cost the first?), band table
set -u
TAG=${1:-r03l}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_poa.py tests/test_gpu_config_goldens.py -m gpu -q 2>&1 | tail -15 ) > $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 1200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --sub-configs band_modes > $OUT/bench_band_modes.json 2> $OUT/err.txt; tail -2 $OUT/err.txt
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_band_modes.json") if l.startswith("{")][0])
print(d["value"], d["roofline"]["kernel_ms"], d["equals_oracle_golden"])
for r in d["sub_records"]["band_modes"]["rows"]: print(r["band_mode"], r["band_width"], r["kernel_ms"], r["gcups"], r["gcups_vs_static_256"])
PY
