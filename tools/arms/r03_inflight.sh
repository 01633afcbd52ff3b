#!/bin/bash
# step time against batches in flight.  usage: tools/r03_inflight.sh n "variants" "inflights"
n=$1; vs=${2:-"0 1"}; fs=${3:-"3 4 6 8"}
for v in $vs; do for f in $fs; do
  timeout -k 10 200 python bench.py --workload 16,1024,$n --cpu-queries 0 --steps 400 --warmup 20 --cells-variant $v --inflight $f > /tmp/o.json 2>/tmp/o.err || { echo FAILED; tail -3 /tmp/o.err; exit 1; }
  python - <<'PY'
import json
d=json.load(open("/tmp/o.json")); r=d["roofline"]
print("n=%s variant=%s inflight=%s: step %.4f ms  kernel alone %.4f  serial step %.4f" % (d["config"]["n_per_gpu"], d["config"]["cells_variant"], d["config"]["batches_in_flight"], d["ms_per_step"], r["kernel_ms"], r["serial_step_ms"]))
PY
done; done
