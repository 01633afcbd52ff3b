"""Wall time of cudaCallback at TA scale (exact path forced), median / min of 20 calls."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import multicore_hw2_amd as pkg
from tests.oracle_lib import Oracle
o = Oracle("oracle/libknn_oracle.so")
pkg.set_option("path", 1)
for (k, m, n) in [(16, 1024, 1024), (16, 1024, 65536), (16, 1024, 1 << 20), (3, 1024, 65536), (16, 100, 65536), (16, 300, 65536)]:
    Q, R = o.synth(m * k, 1000), o.synth(n * k, 1001)
    for _ in range(3):
        got = pkg.cudaCallback(k, m, n, Q, R)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); pkg.cudaCallback(k, m, n, Q, R); ts.append(time.perf_counter() - t0)
    print((k, m, n), "median %.3f ms min %.3f ms" % (sorted(ts)[10] * 1e3, min(ts) * 1e3), flush=True)
