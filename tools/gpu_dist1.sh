#!/bin/bash
# the bench's distributed code path on ONE rank (1-rank RCCL group) against the plain loop, per-rank shard of N = 8; and the two-rank rehearsal tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/dist1; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_baseline_configs_gpu.py -x -q -m gpu -k "rehears or exchange" > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for rep in 1 2 3; do
  for mode in plain dist; do
    if [ $mode = dist ]; then export KNN_BENCH_FORCE_DIST=1; else unset KNN_BENCH_FORCE_DIST; fi
    timeout -k 10 200 python3 bench.py --workload 16,1024,2097152 --cpu-queries 0 > $O/${mode}.json 2> $O/${mode}.err || { tail -5 $O/${mode}.err; exit 1; }
    python3 -c "
import json; d=json.loads(open('$O/${mode}.json').read().strip().splitlines()[-1]); print('$mode', round(d['ms_per_step'],4), d['config'].get('collective_ms_per_group_alone'))"
  done
done
