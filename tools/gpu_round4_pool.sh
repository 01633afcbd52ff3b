#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pool; mkdir -p $O; cd $R
timeout -k 10 400 python -m pytest tests/test_shards_gpu.py tests/test_cells_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2; do
for p in 0 10 20 30 40; do
  timeout -k 10 120 python bench.py --steps 400 --warmup 20 --cpu-queries 0 --opt scan_pool=$p > $O/b_$p.json 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
  python3 -c "
import json
d=json.loads(open('$O/b_$p.json').read().strip().splitlines()[-1])
print('c3 scan_pool=$p ms_per_step %.4f kernel_ms %.4f serial %.4f' % (d['ms_per_step'], d['roofline'].get('kernel_ms',0), d['roofline'].get('serial_step_ms',0)))"
done
done
export KNN_MI355X_LIB=$R/tools/libknn_timeline.so
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --cpu-queries 0 --scan-stamps $O/s_c3.npz > $O/b.json 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
python tools/scan_timeline.py $O/s_c3.npz > $O/timeline_c3.txt; cat $O/timeline_c3.txt
