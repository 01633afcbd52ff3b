set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_deepk
timeout -k 10 500 python -m pytest tests/test_parity_gpu.py -x -q -k "synthetic_uniform or stay_inside or accumulation or deep_dimensions or ragged" 2>&1 | tail -5
for w in ${WORKLOADS:-256,65536,65536 512,65536,65536 256,1024,1048576}; do
  for p in ${PATHS:-0 1}; do
    echo "== workload $w path $p"
    timeout -k 10 200 python bench.py --workload $w --path $p --steps 10 --warmup 2 --cpu-queries 0 > gpurun_out/r03_deepk/b_${w//,/_}_p$p.json 2> gpurun_out/r03_deepk/b_${w//,/_}_p$p.err || { tail -5 gpurun_out/r03_deepk/b_${w//,/_}_p$p.err; }
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03_deepk/b_${w//,/_}_p$p.json").read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["value"], d["roofline"].get("kernel"), d["roofline"].get("kernel_ms"), d["roofline"].get("frac"))
except Exception as e: print("ERR", e)
PY
  done
done
