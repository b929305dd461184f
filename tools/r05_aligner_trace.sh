# host timeline of align_all() / sync_alignments() for configs[4] (GW_ALIGNER_TRACE), last step of a short bench run
for mode in ${MODES:-packed raw}; do
  if [ $mode = raw ]; then export GW_ALIGNER_RAW_UPLOAD=1; else unset GW_ALIGNER_RAW_UPLOAD; fi
  echo "== $mode"
  GW_ALIGNER_TRACE=1 timeout 300 python bench.py --sub-configs aligner --steps 3 --no-cpu-baseline 2>&1 >/dev/null | grep aligner | tail -${LINES:-12}
done
