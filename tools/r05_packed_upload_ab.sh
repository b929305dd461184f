# banded aligner: parity tests, then configs[1] / configs[4] with the packed (default) and the raw (GW_ALIGNER_RAW_UPLOAD=1) upload
# on one box, and the host timeline of align_all() / sync_alignments() (GW_ALIGNER_TRACE)
mkdir -p gpurun_out/r05r
timeout 900 python -m pytest ${TESTS:-tests/test_gpu_aligner.py tests/test_gpu_aligner_vectors.py tests/test_gpu_config_goldens.py tests/test_gpu_pygenomeworks_bindings.py tests/test_overlap_alignment.py} -x -q -m gpu > gpurun_out/r05r/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05r/pytest.log
for i in ${REPS:-1 2 3}; do
  for mode in ${MODES:-packed raw}; do
    if [ $mode = raw ]; then export GW_ALIGNER_RAW_UPLOAD=1; else unset GW_ALIGNER_RAW_UPLOAD; fi
    timeout 300 python bench.py --sub-configs aligner --steps 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); sr=d['sub_records']
print('$mode', 'c1 ms', sr['configs[1]']['ms'], 'c4 ms', sr['configs[4]']['ms'], 'pairs/s', sr['configs[4]']['value'], [ (k,v) for k,v in sr['configs[4]'].items() if 'golden' in k])
"
  done
done
for mode in ${TRACE_MODES:-packed raw}; do
  if [ $mode = raw ]; then export GW_ALIGNER_RAW_UPLOAD=1; else unset GW_ALIGNER_RAW_UPLOAD; fi
  echo "== $mode"
  GW_ALIGNER_TRACE=1 timeout 300 python bench.py --sub-configs aligner --steps 3 --no-cpu-baseline 2>&1 >/dev/null | grep aligner | tail -8
done
