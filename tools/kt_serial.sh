#!/bin/bash
# Per-kernel durations of one batch at a time (rocprofv3 --kernel-trace --stats, bench.py --serial).
# usage: tools/kt_serial.sh out_dir n [extra bench args]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/$1; n=$2; shift 2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_n${n} -- python3 $R/bench.py --workload 16,1024,$n --cpu-queries 0 --serial --steps 100 --warmup 5 "$@" > $O/kt_n${n}.json 2> $O/kt_n${n}.err || { echo "FAILED (exit $?)"; tail -5 $O/kt_n${n}.err; exit 1; }
f=$(find $O/kt_n${n} -name "*kernel_stats.csv" | head -1)
echo "== n=$n (serial): kernel, calls, avg us"
python3 - "$f" $O/kt_n${n}.json <<'PY'
import csv,sys,json
rows=list(csv.DictReader(open(sys.argv[1])))
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
tot=0
for r in rows:
    if int(r["Calls"])>=100:
        print("  %-70s %6s %9.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3)); tot+=float(r["AverageNs"])/1e3
print("  sum of averages %.1f us; ms_per_step %.4f serial_step_ms %.4f" % (tot, d["ms_per_step"], d["roofline"]["serial_step_ms"]))
PY
