# A/B of the deep-K scans (k > 512 shapes) against tools/libknn_prev.so on one box; parity suite in front.  usage (GPU box): bash tools/ab_deepk.sh
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_dk; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
run() { tag=$1; shift; timeout -k 10 300 python bench.py --cpu-queries 0 "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -3 $O/$tag.err; return 1; }
  python -c "
import json; d=json.load(open('$O/$tag.json')); r=d['roofline']; print('%-20s step %.4f kernel %.4f frac %.3f serial %.4f' % ('$tag', d['ms_per_step'], r['kernel_ms'], r['frac'], r['serial_step_ms']))"; }
for lib in prev cur prev cur; do
  if [ $lib = prev ]; then export KNN_MI355X_LIB=$GRAFT_REPO_ROOT/tools/libknn_prev.so; else unset KNN_MI355X_LIB; fi
  run ${lib}_k1024big --workload 1024,65536,65536 --steps 10 --warmup 2
  run ${lib}_k640 --workload 640,16384,65536
  run ${lib}_k1024 --workload 1024,16384,65536
  run ${lib}_k2048 --workload 2048,8192,32768
  run ${lib}_k1024small --workload 1024,2048,65536
done
