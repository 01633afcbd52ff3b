set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_shards_gpu.py tests/test_cells_gpu.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit 1
bash tools/kt_emulate.sh $O 8:0 serial || exit 1
for N in 8 4 2; do
  timeout -k 10 300 python bench.py --emulate $N:0 > $O/emu_${N}_0.json 2> $O/emu_${N}_0.err || { echo "emulate $N failed"; tail -5 $O/emu_${N}_0.err; exit 1; }
  python -c "
import json; d=json.load(open('$O/emu_${N}_0.json')); print('N=$N rank 0: rows', d['config']['n_per_gpu'], 'step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'serial', round(d['roofline']['serial_step_ms'],4), 'rerank', d['config']['rerank_candidates'])"
done
bash tools/kt_serial.sh $O 2097152 || exit 1
bash tools/kt_serial.sh $O 16777216 || exit 1
timeout -k 10 300 python bench.py --cpu-queries 0 > $O/c3.json 2> $O/c3.err && python -c "
import json; d=json.load(open('$O/c3.json')); print('C3 1 GPU: step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'serial', round(d['roofline']['serial_step_ms'],4))"
timeout -k 10 300 python bench.py --cpu-queries 0 --workload 16,1024,2097152 > $O/n21.json 2> $O/n21.err && python -c "
import json; d=json.load(open('$O/n21.json')); print('2^21 index-range shard: step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'serial', round(d['roofline']['serial_step_ms'],4))"
