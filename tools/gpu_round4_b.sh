set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_cells_gpu.py tests/test_parity_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q > $O/gpu_suite.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -15 $O/gpu_suite.log
[ $rc -eq 0 ] || exit 1
for n in 2097152 16777216; do bash tools/kt_serial.sh $O $n || exit 1; done
python bench.py --cpu-queries 64 > $O/c3_bench.json 2> $O/c3_bench.err && python -c "
import json; d=json.load(open('$O/c3_bench.json')); print('C3 step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'serial', d['roofline']['serial_step_ms'])"
python bench.py --cpu-queries 0 --workload 16,1024,2097152 > $O/n21_bench.json 2> $O/n21_bench.err && python -c "
import json; d=json.load(open('$O/n21_bench.json')); print('2^21 step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'serial', d['roofline']['serial_step_ms'])"
