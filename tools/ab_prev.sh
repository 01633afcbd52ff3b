# A/B of the working tree's library against tools/libknn_prev.so (a build of an earlier commit) on the same box
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for n in ${SIZES:-2097152 16777216}; do
  for lib in prev cur; do
    if [ $lib = prev ]; then export KNN_MI355X_LIB=$GRAFT_REPO_ROOT/tools/libknn_prev.so; else unset KNN_MI355X_LIB; fi
    timeout -k 10 120 python bench.py --workload 16,1024,$n --steps 300 --warmup 20 --cpu-queries 0 > /tmp/ab.json 2>/dev/null || echo FAILED
    python - <<PY
import json
d=json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
print("$lib", $n, "step %.4f  alone %.4f  serial %.4f" % (d["ms_per_step"], d["roofline"].get("kernel_ms"), d["roofline"].get("serial_step_ms")))
PY
  done
done
done
