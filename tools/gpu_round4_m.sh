set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_m; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q -k "deep or c5 or filter or ragged or randomised or synthetic" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit 1
run() { tag=$1; shift; timeout -k 10 300 python bench.py --cpu-queries 0 "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -3 $O/$tag.err; return 1; }
  python -c "
import json; d=json.load(open('$O/$tag.json')); r=d['roofline']; print('%-20s step %.4f kernel %.4f frac %.3f serial %.4f' % ('$tag', d['ms_per_step'], r['kernel_ms'], r['frac'], r['serial_step_ms']))"; }
for rep in 1 2 3; do
KNN_MI355X_X16=0 run c5_32x32_$rep --workload c5
run c5_16x16_$rep --workload c5
done
KNN_MI355X_X16=0 run k96_32x32 --workload 96,16384,65536
run k96_16x16 --workload 96,16384,65536
