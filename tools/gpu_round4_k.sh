set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_k; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_shards_gpu.py tests/test_cells_gpu.py tests/test_parity_gpu.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
KNN_MI355X_TRACE_BUILD=1 python tools/build_trace.py 2> $O/build_trace.txt; tail -14 $O/build_trace.txt
KNN_MI355X_LIB=$GRAFT_REPO_ROOT/tools/libknn_prev.so KNN_MI355X_TRACE_BUILD=1 python tools/build_trace.py 2> $O/build_trace_prev.txt; tail -9 $O/build_trace_prev.txt
