"""A/B of cudaCallback staged (copy everything, then scan) vs streamed (exact scan of each chunk under the
copy of the next) on one box: median-ish of 4 calls, two rounds each.  usage: python tools/stream_ab.py"""
import sys, time, numpy as np
sys.path.insert(0, ".")
import multicore_hw2_amd as pkg
from tests.oracle_lib import Oracle
o = Oracle("oracle/libknn_oracle.so")
for (k, m, n) in [(16, 1024, 1 << 20), (16, 1024, 1 << 21), (16, 1024, 1 << 22), (16, 1024, 1 << 23), (16, 1024, 1 << 24), (16, 256, 1 << 22), (16, 2048, 1 << 22), (3, 1024, 1 << 24)]:
    Q, R = o.synth(m * k, 1000), o.synth(n * k, 1001)
    res = {}
    for stream in (1, 2, 1, 2):
        pkg.set_option("stream", stream)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); pkg.cudaCallback(k, m, n, Q, R); ts.append(time.perf_counter() - t0)
        res.setdefault(stream, []).append(sorted(ts)[1] * 1e3)
    print((k, m, n), "staged %s ms   streamed %s ms" % ([round(x, 2) for x in res[1]], [round(x, 2) for x in res[2]]), flush=True)
