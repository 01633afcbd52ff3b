set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_shards_gpu.py -m gpu -x -q > $O/shards.log 2>&1; rc=$?; echo "shards rc=$rc"; tail -25 $O/shards.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m pytest tests/test_cells_gpu.py tests/test_parity_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q > $O/gpu_suite.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -5 $O/gpu_suite.log
