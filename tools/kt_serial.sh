#!/bin/bash
# Per-kernel durations of one batch at a time (rocprofv3 --kernel-trace --stats, bench.py --serial).
# usage: tools/kt_serial.sh out_dir n variant [extra bench args]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/$1; n=$2; v=$3; shift 3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_n${n}_v${v} -- python3 $R/bench.py --workload 16,1024,$n --cpu-queries 0 --serial --steps 100 --warmup 5 --cells-variant $v "$@" > $O/kt_n${n}_v${v}.json 2> $O/kt_n${n}_v${v}.err || exit 1
f=$(find $O/kt_n${n}_v${v} -name "*kernel_stats.csv" | head -1)
echo "== n=$n variant=$v (serial): kernel, calls, avg us"
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if int(r["Calls"])>=100:
        print("  %-60s %6s %9.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
