#!/bin/bash
# SQ counters of the cell-pruned path's kernels, one batch at a time.  usage: tools/pmc_sq.sh out_dir n tag
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/$1; n=$2; v=$3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_${name}_n${n}_v${v} -- python3 $R/bench.py --workload 16,1024,$n --cpu-queries 0 --serial --steps 20 --warmup 2 > /dev/null 2> $O/pmc_${name}_n${n}_v${v}.err || { echo "pmc pass $name failed"; tail -3 $O/pmc_${name}_n${n}_v${v}.err; return 1; }
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES || exit 1
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD || exit 1
python3 - $O $n $v <<'PY'
import csv,glob,sys,collections
O,n,v=sys.argv[1:4]
for name in "ab":
    files=glob.glob("%s/pmc_%s_n%s_v%s/**/*counter_collection.csv"%(O,name,n,v),recursive=True)
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVES","SQ_INSTS_VALU"): cnt[k]+=1
    for k,d in acc.items():
        if not any(t in k for t in ("sweep","scan","match","prep","seed")): continue
        c=max(cnt[k],1)
        print("n=%s v=%s %-40s per launch: "%(n,v,k)+"  ".join("%s=%.3g"%(a[3:],b/c) for a,b in sorted(d.items())))
PY
