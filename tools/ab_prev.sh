# A/B of the working tree's library against tools/libknn_prev.so (a build of an earlier commit) on the same box.
# usage (on the GPU box): bash tools/ab_prev.sh ["bench args" ...]   (default: C3 and the emulated ranks of 8 and 4)
cd $GRAFT_REPO_ROOT
if [ $# -eq 0 ]; then set -- "--workload c3" "--emulate 8:0" "--emulate 4:0"; fi
for rep in 1 2 3; do
for args in "$@"; do
  for lib in prev cur; do
    if [ $lib = prev ]; then export KNN_MI355X_LIB=$GRAFT_REPO_ROOT/tools/libknn_prev.so; else unset KNN_MI355X_LIB; fi
    timeout -k 10 120 python bench.py $args --cpu-queries 0 > /tmp/ab.json 2>/tmp/ab.err || { echo "$lib $args FAILED"; tail -2 /tmp/ab.err; continue; }
    python - <<PY
import json
d=json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
print("%-5s %-22s step %.4f  kernel alone %.4f  one batch at a time %.4f" % ("$lib", "$args", d["ms_per_step"], d["roofline"].get("kernel_ms"), d["roofline"].get("serial_step_ms")))
PY
  done
done
done
