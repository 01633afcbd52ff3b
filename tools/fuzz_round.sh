#!/bin/bash
# randomised differential runs against the CPU oracle on the final sources of round 4 -> gpurun_out/fuzz/
cd $GRAFT_REPO_ROOT; O=gpurun_out/fuzz; mkdir -p $O
run() { tag=$1; shift; echo "== $tag: $*"; ( "$@" ) > $O/$tag.txt 2>&1; rc=$?; tail -2 $O/$tag.txt; echo "rc=$rc"; }
{
run one   timeout -k 10 330 python tools/fuzz_parity.py 1200 20264
run cells env FUZZ_CELLS=1 timeout -k 10 400 python tools/fuzz_parity.py 700 31415
run big   env FUZZ_BIG=1 timeout -k 10 200 python tools/fuzz_parity.py 100 17
} > $O/summary.txt 2>&1
cat $O/summary.txt
