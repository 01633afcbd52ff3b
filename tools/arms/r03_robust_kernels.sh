set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_robust
for c in clusters64 heavy_tail; do
rm -rf /tmp/prof_c64
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c64 -o c64 --output-format csv -- python3 tools/distribution_check.py $c > /dev/null 2>&1
f=$(find /tmp/prof_c64 -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r03_robust/${c}_kernel_stats.csv
echo "== $c"
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:16]:
    if "at::" in r["Name"] or "rocclr" in r["Name"]: continue
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MaxNs"])
PY
done
