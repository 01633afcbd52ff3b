#!/bin/bash
# round 5: A/B bench lines on ONE box.  usage: r05_ab.sh <outdir under gpurun_out> <pytest selection or -> -- name|bench args ...
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd $R
if [ "$1" != "-" ]; then
  timeout -k 10 600 python -m pytest $1 -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
  tail -2 $O/pytest.txt
fi
shift
for spec in "$@"; do
  name=${spec%%|*}; args=${spec#*|}
  timeout -k 10 240 python3 bench.py --cpu-queries 0 $args > $O/${name}_bench.json 2>> $O/bench.err || { echo "bench $name failed"; tail -5 $O/bench.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open('$O/${name}_bench.json'))
r=d['roofline']
print('%-28s step %.4f  serial %.4f  kernel %.4f  in-pipe %.4f  host-enqueue %.4f  frac %s' % ('$name', d['ms_per_step'], r['serial_step_ms'] or 0, r['kernel_ms'], r['kernel_in_pipeline_ms'], d['config'].get('host_enqueue_ms_per_step', 0), r.get('frac')))
PY
done
echo done
