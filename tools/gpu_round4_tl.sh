#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl; mkdir -p $O; cd $R
export KNN_MI355X_LIB=$R/tools/libknn_timeline.so
for e in 8:0 4:1 1:0; do
  for d in 0 3; do
    [ $e = 1:0 ] && [ $d = 3 ] && continue
    em="--emulate $e"; [ $e = 1:0 ] && em=""
    timeout -k 10 200 python bench.py $em --steps 200 --warmup 20 --cpu-queries 0 --opt scan_deal=$d --scan-stamps $O/s_${e/:/_}_$d.npz > $O/b.json 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
    echo "=== emulate $e scan_deal=$d"; python tools/scan_timeline.py $O/s_${e/:/_}_$d.npz
  done
done > $O/timeline.txt 2>&1
cat $O/timeline.txt
