#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/il; mkdir -p $O; cd $R
timeout -k 10 400 python -m pytest tests/test_shards_gpu.py tests/test_cells_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
bash tools/ab_prev.sh "--workload c3" "--emulate 8:0 --serial" "--emulate 2:0" "--workload 16,1024,8388608" > $O/ab.txt 2>&1
cat $O/ab.txt
export KNN_MI355X_LIB=$R/tools/libknn_timeline.so
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --cpu-queries 0 --scan-stamps $O/s_c3.npz > $O/b.json 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
python tools/scan_timeline.py $O/s_c3.npz > $O/timeline_c3.txt; cat $O/timeline_c3.txt
