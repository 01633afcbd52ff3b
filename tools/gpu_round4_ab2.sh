#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab2; mkdir -p $O; cd $R
timeout -k 10 400 python -m pytest tests/test_shards_gpu.py tests/test_cells_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
bash tools/ab_prev.sh "--workload c3" "--workload 16,1024,8388608" "--emulate 8:0 --serial" "--workload c3 --serial" > $O/ab.txt 2>&1
cat $O/ab.txt
