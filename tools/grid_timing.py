#!/usr/bin/env python3
"""Grid index (k <= 4): build time and query time against the brute-force paths, resident and one-shot.
usage: python tools/grid_timing.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import multicore_hw2_amd as pkg
from tests.oracle_lib import Oracle

o = Oracle(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "libknn_oracle.so"))
dev = torch.device("cuda:0")


def best(fn, reps=5):
    t = 1e30
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        t = min(t, (time.perf_counter() - t0) * 1e3)
    return t


for (k, m, n) in [(3, 1024, 1 << 20), (3, 1024, 1 << 16), (2, 1024, 1 << 20), (4, 1024, 1 << 20), (3, 1024, 1 << 24)]:
    Q, R = o.synth(m * k, 1000), o.synth(n * k, 1001)
    q_d, r_d = torch.from_numpy(Q).to(dev), torch.from_numpy(R).to(dev)
    keys = torch.empty(m, dtype=torch.int64, device=dev)
    line = "(%d, %d, %d):" % (k, m, n)
    for path, name in ((3, "grid"), (2, "filter"), (1, "exact")):
        pkg.set_option("path", path)
        held = []
        create = best(lambda: held.append(pkg.KnnIndex(k, r_d.data_ptr(), n_local=n, refs_on_device=True)), reps=3)
        ix = held[-1]

        def query():
            pkg.keys_init(keys.data_ptr(), m)
            ix.query_keys(m, q_d.data_ptr(), keys.data_ptr())
        qms = best(query)
        for h in held:
            h.close()
        oneshot = best(lambda: pkg.cudaCallback(k, m, n, Q, R), reps=3) if n <= (1 << 20) else float("nan")
        line += "  %s: create %.3f ms, query %.3f ms, one-shot call %.3f ms |" % (name, create, qms, oneshot)
    pkg.set_option("path", 0)
    auto = best(lambda: pkg.cudaCallback(k, m, n, Q, R), reps=3) if n <= (1 << 20) else float("nan")
    print(line + "  one-shot auto %.3f ms" % auto, flush=True)
