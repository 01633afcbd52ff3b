cd $GRAFT_REPO_ROOT
for n in ${SIZES:-16777216}; do
for deal in 1 2; do for blocks in 2 1; do for inf in 2 3 4; do
  timeout -k 10 120 python bench.py --workload 16,1024,$n --steps 300 --warmup 20 --cpu-queries 0 --inflight $inf --opt scan_deal=$deal --opt scan_blocks=$blocks > /tmp/ab.json 2>/dev/null || echo FAILED
  python - <<PY
import json
d=json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
print("n=$n deal=$deal blocks=$blocks inflight=$inf", "step %.4f  alone %.4f  serial %.4f" % (d["ms_per_step"], d["roofline"].get("kernel_ms"), d["roofline"].get("serial_step_ms")))
PY
done; done; done; done
