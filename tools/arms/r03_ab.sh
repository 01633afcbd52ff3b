#!/bin/bash
# A/B of the cell-pruned path's kernels by shard size (round 3).  usage: tools/r03_ab.sh out_dir [sizes...]
out=${1:-gpurun_out/r03_ab}; shift
sizes=${@:-"2097152 4194304 16777216"}
mkdir -p "$out"
for n in $sizes; do
  for v in 0 1 2 3; do
    f="$out/n${n}_v${v}.json"
    timeout -k 10 300 python bench.py --workload 16,1024,$n --cpu-queries 0 --steps 300 --warmup 20 --cells-variant $v > "$f" 2> "$out/n${n}_v${v}.err" || { echo "FAILED n=$n v=$v"; tail -5 "$out/n${n}_v${v}.err"; exit 1; }
    python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
r=d["roofline"]
print("n=%s variant=%s inflight=%s: step %.4f ms, kernel alone %.4f, in pipeline %.4f, serial step %.4f, frac %.3f, candidates %s" % (
    d["config"]["n_per_gpu"], d["config"]["cells_variant"], d["config"]["batches_in_flight"], d["ms_per_step"], r["kernel_ms"],
    r["kernel_in_pipeline_ms"], r["serial_step_ms"], r["frac"] or 0, d["config"]["rerank_candidates"]))
PY
  done
done
