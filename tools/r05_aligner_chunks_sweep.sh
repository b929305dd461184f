# configs[4] by the number of upload / kernel chunks of align_all() (GW_ALIGNER_CHUNKS), two repetitions per row
out=${1:-r05d}
mkdir -p gpurun_out/$out
for c in ${CHUNKS:-2 4 6 8 12 16 24}; do
  for rep in 1 2; do
  GW_ALIGNER_CHUNKS=$c python bench.py --sub-configs aligner --steps ${STEPS:-2} --no-cpu-baseline > gpurun_out/$out/bench_c$c.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/$out/bench_c$c.json") if l.startswith("{")][0])
s=d["sub_records"]["configs[4]"]
print("chunks", $c, "pairs/s", s["value"], "ms", s["ms"], "device_resident", s["device_resident"]["ms"], "kernels", s["kernel_only"]["ms"])
PY
  done
done
