#!/bin/bash
# Per-kernel durations with three batches in flight (rocprofv3 --kernel-trace --stats of the default bench loop).
# usage: tools/kt_pipe.sh out_dir n [extra bench args]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/$1; n=$2; shift 2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktp_n${n} -- python3 $R/bench.py --workload 16,1024,$n --cpu-queries 0 --steps 300 --warmup 10 "$@" > $O/ktp_n${n}.json 2> $O/ktp_n${n}.err || exit 1
f=$(find $O/ktp_n${n} -name "*kernel_stats.csv" | head -1)
python3 - "$f" $O/ktp_n${n}.json <<'PY'
import csv,sys,json
rows=list(csv.DictReader(open(sys.argv[1])))
d=json.load(open(sys.argv[2]))
print("== n=%s pipelined: step %.4f ms; kernel, calls, avg us (sum of averages = GPU time a batch holds)" % (d["config"]["n_per_gpu"], d["ms_per_step"]))
tot=0
for r in rows:
    if int(r["Calls"])>=300:
        print("  %-60s %6s %9.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3)); tot+=float(r["AverageNs"])/1e3
print("  sum %.1f us" % tot)
PY
