set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_j; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_shards_gpu.py tests/test_cells_gpu.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
run() { tag=$1; shift; timeout -k 10 300 python bench.py --cpu-queries 0 "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -3 $O/$tag.err; return 1; }
  python -c "
import json; d=json.load(open('$O/$tag.json')); print('%-28s step %.4f kernel %.4f serial %.4f' % ('$tag', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['serial_step_ms']))"; }
for rep in 1 2; do
run c3_$rep
run emu8_$rep --emulate 8:0
run emu4_$rep --emulate 4:0
done
bash tools/kt_serial.sh $O 16777216
