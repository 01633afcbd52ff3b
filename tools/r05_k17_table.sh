#!/bin/bash
# round 5: the k 17..32 table of profiles/r05_cells_k17_32.txt on ONE box: pruned scan (--cells 1) against the full scan (--cells 2).
# usage (on the GPU box): bash tools/r05_k17_table.sh <outdir under gpurun_out> "<k list>" "<n list>"
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for n in $3; do for k in $2; do for c in 1 2; do
  timeout -k 10 240 python3 bench.py --cpu-queries 0 --workload $k,1024,$n --steps 60 --cells $c > $O/k${k}_n${n}_c$c.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
done
python3 - <<PY
import json
a=json.load(open('$O/k${k}_n${n}_c1.json')); b=json.load(open('$O/k${k}_n${n}_c2.json'))
print('k %2d n %9d  pruned %.4f (scan alone %.4f)  full %.4f  ratio %.2f' % ($k, $n, a['ms_per_step'], a['roofline']['kernel_ms'], b['ms_per_step'], b['ms_per_step']/a['ms_per_step']))
PY
done; done
