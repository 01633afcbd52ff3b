set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_a
python -m pytest tests -m gpu -x -q > gpurun_out/r04_a/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -3 gpurun_out/r04_a/gpu_suite.log
for n in 2097152 4194304 16777216; do bash tools/kt_serial.sh gpurun_out/r04_a $n || exit 1; done
python bench.py --cpu-queries 0 > gpurun_out/r04_a/c3_bench.json 2> gpurun_out/r04_a/c3_bench.err && python -c "
import json; d=json.load(open('gpurun_out/r04_a/c3_bench.json')); print('C3 step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'serial', d['roofline']['serial_step_ms'])"
python bench.py --cpu-queries 0 --workload 16,1024,2097152 > gpurun_out/r04_a/n21_bench.json 2> gpurun_out/r04_a/n21_bench.err && python -c "
import json; d=json.load(open('gpurun_out/r04_a/n21_bench.json')); print('2^21 step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'serial', d['roofline']['serial_step_ms'])"
