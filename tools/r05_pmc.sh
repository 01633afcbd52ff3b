#!/bin/bash
# round 5: HBM traffic per launch from PMC counters, per WORKLOAD KEY: separate rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE, TCC hit/miss) of the same bench command; tools/pmc_traffic.py folds them into
# profiles/<tag>_pmc_traffic.json with the key bench.py prints as roofline.pmc_key.
# usage (GPU box): bash tools/r05_pmc.sh <tag> <bench args...>      e.g.  r05_pmc.sh c3_rank_0_of_8 --emulate 8:0
set -o pipefail
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/r05_pmc_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PMC() { out=$1; shift; timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$out -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-queries 0 $ARGS > $O/pmc_$out.json 2> $O/pmc_$out.err || { echo "pmc $out failed"; tail -5 $O/pmc_$out.err; exit 1; }; }
ARGS="$*"
PMC fetch FETCH_SIZE
PMC write WRITE_SIZE
PMC l2 TCC_HIT_sum TCC_MISS_sum
cd $R
python3 tools/pmc_traffic.py r05_$tag $O/pmc_fetch $O/pmc_write $O/pmc_l2 --bench-json $O/pmc_fetch.json --cmd "bench.py --steps 5 --warmup 1 --cpu-queries 0 $ARGS" | grep -i "scan\|prep\|match\|tail\|filter" | head -12
cp profiles/r05_${tag}_pmc_traffic.json $O/
echo "done $tag"
