#!/bin/bash
# A/B of the request-ahead scan (scan_deal 0 = auto vs 3 = fixed deal without it) on emulated ranks, and the shard tests
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/pf; mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_shards_gpu.py tests/test_cells_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for e in 8:0 8:3 4:1 2:0; do
  for d in 3 0 3 0; do
    timeout -k 10 120 python bench.py --emulate $e --steps 400 --warmup 20 --cpu-queries 0 --opt scan_deal=$d > $O/b_${e/:/_}_$d.json 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
    python3 -c "
import json,sys
d=json.loads(open('$O/b_${e/:/_}_$d.json').read().strip().splitlines()[-1])
print('emulate $e scan_deal=$d ms_per_step %.4f kernel_ms %.4f serial %.4f' % (d['ms_per_step'], d['roofline'].get('kernel_ms',0), d['roofline'].get('serial_step_ms',0)))"
  done
done
