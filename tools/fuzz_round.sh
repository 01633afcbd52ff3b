#!/bin/bash
# randomised differential runs against the CPU oracle on the final sources of the round -> gpurun_out/fuzz/
cd $GRAFT_REPO_ROOT; O=gpurun_out/fuzz; mkdir -p $O
run() { tag=$1; shift; echo "== $tag: $*"; ( "$@" ) > $O/$tag.txt 2>&1; rc=$?; tail -2 $O/$tag.txt; echo "rc=$rc"; }
{
run one   timeout -k 10 330 python tools/fuzz_parity.py 1400 ${FUZZ_SEED:-60606}
run cells env FUZZ_CELLS=1 timeout -k 10 560 python tools/fuzz_parity.py 900 ${FUZZ_SEED2:-16180}
run big   env FUZZ_BIG=1 timeout -k 10 200 python tools/fuzz_parity.py 100 ${FUZZ_SEED3:-23}
} > $O/summary.txt 2>&1
cat $O/summary.txt
