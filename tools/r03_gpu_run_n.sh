#!/bin/bash
# round 3, GPU call N: long-read topological sort with its hot state in LDS (A/B against the HBM routine, heaviest window
# phases, 598-window golden, sub-record), wave-per-pair default aligner with / without the even-spread LDS request
set -u
TAG=${1:-r03n}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_poa.py -m gpu -q -x -k "long_read" 2>&1 | tail -5 ) > $OUT/pytest.log
( timeout 300 python -m pytest tests/test_gpu_aligner.py -m gpu -q -x -k "default or hirschberg" 2>&1 | tail -3 ) >> $OUT/pytest.log
cat $OUT/pytest.log
echo "lds state : $(timeout 300 python tools/profile_long_read.py 389 1 2>&1 | tail -1)" > $OUT/heaviest_window_phases.txt
echo "hbm state : $(GWHIP_DEBUG=65536 timeout 300 python tools/profile_long_read.py 389 1 2>&1 | tail -1)" >> $OUT/heaviest_window_phases.txt
cat $OUT/heaviest_window_phases.txt
( timeout 900 python -m pytest tests/test_gpu_config_goldens.py -m gpu -q -x -k "config4" 2>&1 | tail -3 ) > $OUT/pytest_golden.log; cat $OUT/pytest_golden.log
timeout 600 python bench.py --sub-configs long_reads --no-cpu-baseline --steps 2 --warmup 1 > $OUT/bench_long_reads.json 2> $OUT/bench.err
python - $OUT/bench_long_reads.json <<'PY'
import json,sys
d=json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][0]); v=d['sub_records']['configs[3]']
print("bench", v['value'], v['unit'], v['ms'], "ms", "golden", v['windows_equal_to_oracle_golden'], "differ", v['windows_differing_from_golden'])
PY
GWHIP_HIRSCHBERG_SPREAD=1 timeout 300 python tools/bench_default_aligner.py quick > $OUT/default_aligner_spread.json 2> $OUT/da1.err; cat $OUT/default_aligner_spread.json | head -c 1500; echo
GWHIP_HIRSCHBERG_SPREAD=0 timeout 300 python tools/bench_default_aligner.py quick > $OUT/default_aligner_packed.json 2> $OUT/da0.err; cat $OUT/default_aligner_packed.json | head -c 1500; echo
