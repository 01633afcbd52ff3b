"""Wall time of cudaCallback over a grid of TA-scale shapes (median of 15 calls after 3 warm-ups):
looks for fixed-cost floors.  usage: python tools/small_call_timing.py [path]"""
import sys, time, numpy as np
sys.path.insert(0, ".")
import multicore_hw2_amd as pkg
from tests.oracle_lib import Oracle
o = Oracle("oracle/libknn_oracle.so")
if len(sys.argv) > 1:
    pkg.set_option("path", int(sys.argv[1]))
for k in (3, 16, 17, 128):
    for m in (1, 32, 100, 1024, 5000):
        row = []
        for n in (100, 1024, 10000, 65536, 300000):
            Q, R = o.synth(m * k, 1000), o.synth(n * k, 1001)
            for _ in range(3):
                pkg.cudaCallback(k, m, n, Q, R)
            ts = []
            for _ in range(15):
                t0 = time.perf_counter(); pkg.cudaCallback(k, m, n, Q, R); ts.append(time.perf_counter() - t0)
            row.append(sorted(ts)[7] * 1e3)
        print("k=%3d m=%5d  n=100/1024/10000/65536/300000: " % (k, m) + "  ".join("%6.3f" % t for t in row) + " ms", flush=True)
