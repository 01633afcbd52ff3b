set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_d; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_baseline_configs_gpu.py -m gpu -x -q -k "rehearsed" > $O/rehearse.log 2>&1; rc=$?; echo "rehearse rc=$rc"; tail -15 $O/rehearse.log
[ $rc -eq 0 ] || exit 1
for N in 8 4 2; do
  timeout -k 10 300 python bench.py --emulate $N:0 > $O/emu_${N}_0.json 2> $O/emu_${N}_0.err || { echo "emulate $N failed"; tail -5 $O/emu_${N}_0.err; exit 1; }
  python -c "
import json; d=json.load(open('$O/emu_${N}_0.json')); print('N=$N rank 0: rows', d['config']['n_per_gpu'], 'step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'serial', round(d['roofline']['serial_step_ms'],4), 'rerank', d['config']['rerank_candidates'])"
done
timeout -k 10 300 python bench.py --emulate 8:5 > $O/emu_8_5.json 2> $O/emu_8_5.err && python -c "
import json; d=json.load(open('$O/emu_8_5.json')); print('N=8 rank 5: rows', d['config']['n_per_gpu'], 'step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'serial', round(d['roofline']['serial_step_ms'],4))"
timeout -k 10 300 python bench.py --cpu-queries 0 > $O/c3.json 2> $O/c3.err && python -c "
import json; d=json.load(open('$O/c3.json')); print('C3 1 GPU: step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'serial', round(d['roofline']['serial_step_ms'],4))"
