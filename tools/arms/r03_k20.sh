cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt20; mkdir -p $R/gpurun_out/r03_k20
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt20 -- python3 $R/bench.py --cpu-queries 0 --steps 20 --warmup 5 > $R/gpurun_out/r03_k20/line.json 2>/dev/null
f=$(find /tmp/kt20 -name "*kernel_trace.csv" | head -1)
cp $f $R/gpurun_out/r03_k20/kernel_trace.csv
python3 - <<PY
import csv,json
rows=list(csv.DictReader(open("$f")))
rows=[r for r in rows if "knn_" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# find the scan kernels; the timed 20 steps are the 20 scans before the 'alone' section: take last (20 alone + 50 serial + 20 timed)...
scans=[r for r in rows if "cells_scan" in r["Kernel_Name"]]
print("scans", len(scans))
d=json.loads(open("$R/gpurun_out/r03_k20/line.json").read().strip().splitlines()[-1]); print("bench says", d["ms_per_step"])
# sequence: 30 setup + 5 warmup + 20 timed + 20 alone + 50 serial = 125 scans (+ index builds)
t=scans[35:55]
t0=int(t[0]["Start_Timestamp"]); 
prep=[r for r in rows if "prep" in r["Kernel_Name"]]
p=prep[35:55]
unp=[r for r in rows if "unpack" in r["Kernel_Name"]]
print("timed region: first prep start -> last scan end: %.1f us" % ((int(t[-1]["End_Timestamp"])-int(p[0]["Start_Timestamp"]))/1e3))
for i,(a,b) in enumerate(zip(p,t)):
    print(i, "prep start %.1f  scan start %.1f end %.1f dur %.1f" % ((int(a["Start_Timestamp"])-int(p[0]["Start_Timestamp"]))/1e3, (int(b["Start_Timestamp"])-int(p[0]["Start_Timestamp"]))/1e3, (int(b["End_Timestamp"])-int(p[0]["Start_Timestamp"]))/1e3, (int(b["End_Timestamp"])-int(b["Start_Timestamp"]))/1e3))
PY
