#!/bin/bash
# SQ counters of the deep-K scan kernels (C5 and the chunked-K form), one batch at a time.
# usage: tools/pmc_sq_deepk.sh out_dir "workload" tag
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/$1; wl=$2; tag=$3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_${name}_$tag -- python3 $R/bench.py --workload $wl --cpu-queries 0 --serial --steps 6 --warmup 1 > /dev/null 2> $O/pmc_${name}_$tag.err || { echo "pmc pass $name failed"; tail -3 $O/pmc_${name}_$tag.err; return 1; }
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES || exit 1
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD || exit 1
run c SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_EXP_GDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS || echo "pass c failed (counter names)"
python3 - $O $tag <<'PY'
import csv,glob,sys,collections
O,tag=sys.argv[1:3]
for name in "abc":
    files=glob.glob("%s/pmc_%s_%s/**/*counter_collection.csv"%(O,name,tag),recursive=True)
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
    for f in files:
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
    for k,d in acc.items():
        if not any(t in k for t in ("tiled","chunked")): continue
        print("%s %-60s per launch: "%(tag,k)+"  ".join("%s=%.4g"%(a[3:],b/max(cnt[k][a],1)) for a,b in sorted(d.items())))
PY
