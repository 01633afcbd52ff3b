#!/bin/bash
# round 5, first GPU call: parity after the vmcnt(0) fix, baselines of the round's box, ingest stage laps
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_a
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -20 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
B() { out=$1; shift; timeout -k 10 200 python3 bench.py --cpu-queries 0 "$@" > $O/${out}_bench.json 2>> $O/bench.err || { echo "bench $out failed"; tail -5 $O/bench.err; exit 1; }; python3 -c "
import json,sys
d=json.load(open('$O/${out}_bench.json'))
r=d['roofline']
print('$out', 'step %.4f' % d['ms_per_step'], 'serial %.4f' % (r['serial_step_ms'] or 0), 'kernel %.4f' % r['kernel_ms'], 'frac', r.get('frac'))
"; }
B c3
B c3_rank_0_of_8 --emulate 8:0
B c3_rank_0_of_4 --emulate 4:0
timeout -k 10 300 python3 tools/ingest_timing.py > $O/ingest_timing.txt 2>&1 || { tail $O/ingest_timing.txt; exit 1; }
grep -v "knn ingest\|knn build\|repetition" $O/ingest_timing.txt
echo done
