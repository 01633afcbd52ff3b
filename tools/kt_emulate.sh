#!/bin/bash
# Per-kernel durations of ONE emulated rank of a cell-range sharded run (rocprofv3 --kernel-trace --stats of bench.py --emulate).
# usage: tools/kt_emulate.sh out_dir N:r [serial|pipe] [extra bench args]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/$1; e=$2; mode=${3:-serial}; shift 3
mkdir -p $O
tag=${e/:/_}_$mode
extra="--serial --steps 100 --warmup 5"; [ "$mode" = pipe ] && extra="--steps 300 --warmup 10"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_emu_$tag -- python3 $R/bench.py --emulate $e $extra "$@" > $O/kt_emu_$tag.json 2> $O/kt_emu_$tag.err || { echo "FAILED"; tail -5 $O/kt_emu_$tag.err; exit 1; }
f=$(find $O/kt_emu_$tag -name "*kernel_stats.csv" | head -1)
echo "== emulate $e ($mode): kernel, calls, avg us"
python3 - "$f" $O/kt_emu_$tag.json <<'PY'
import csv,sys,json
rows=list(csv.DictReader(open(sys.argv[1])))
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
tot=0
for r in rows:
    if int(r["Calls"])>=100:
        print("  %-70s %6s %9.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3)); tot+=float(r["AverageNs"])/1e3
print("  sum of averages %.1f us; rows %d ms_per_step %.4f serial_step_ms %.4f" % (tot, d["config"]["n_per_gpu"], d["ms_per_step"], d["roofline"]["serial_step_ms"]))
PY
