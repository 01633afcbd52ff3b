#!/bin/bash
# A/B on ONE box: alternate builds of the same C-ABI (KNN_MI355X_LIB) over several rounds.
# usage: tools/ab_bench.sh "<bench args>" libA.so libB.so ...
args="$1"; shift
for round in 1 2 3; do
  for lib in "$@"; do
    KNN_MI355X_LIB=$(realpath $lib) python bench.py $args --cpu-queries 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('%-28s step_ms %.4f  dominant_kernel_ms %.4f  qps %.0f' % (sys.argv[1], d['ms_per_step'], r['kernel_avg_ms'], d['value']))" $(basename $lib)
  done
done
