set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_g; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_shards_gpu.py tests/test_cells_gpu.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit 1
run() { tag=$1; shift; timeout -k 10 300 python bench.py --cpu-queries 0 "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -3 $O/$tag.err; return 1; }
  python -c "
import json; d=json.load(open('$O/$tag.json')); print('%-28s step %.4f kernel %.4f serial %.4f inflight %d' % ('$tag', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['serial_step_ms'], d['config']['batches_in_flight']))"; }
run emu8 --emulate 8:0
KNN_MI355X_PREP_WAVES=2 run emu8_pw2 --emulate 8:0
run emu4 --emulate 4:0
KNN_MI355X_PREP_WAVES=2 run emu4_pw2 --emulate 4:0
run emu2 --emulate 2:0
run c3
run n21 --workload 16,1024,2097152
bash tools/kt_emulate.sh $O 8:0 serial
