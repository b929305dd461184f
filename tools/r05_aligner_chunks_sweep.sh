mkdir -p gpurun_out/r05d
for c in 2 4 6 8 12 16 24; do
  GW_ALIGNER_CHUNKS=$c python bench.py --sub-configs aligner --steps 2 --no-cpu-baseline > gpurun_out/r05d/bench_c$c.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r05d/bench_c$c.json") if l.startswith("{")][0])
s=d["sub_records"]["configs[4]"]
print("chunks", $c, "pairs/s", s["value"], "ms", s["ms"], "device_resident", s["device_resident"]["ms"], "kernels", s["kernel_only"]["ms"])
PY
done
