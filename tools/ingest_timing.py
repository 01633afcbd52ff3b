#!/usr/bin/env python3
"""what an index created from HOST rows costs (C3: k 16, n 2^24 = 1 GiB), next to the
bare pageable copy, and whether page-locking the caller's buffer first (hipHostRegister) pays.
usage: python tools/ingest_timing.py  -> prints one line per measurement (ms)"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import multicore_hw2_amd as pkg
from tests.oracle_lib import Oracle

o = Oracle(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "libknn_oracle.so"))
k, m, n = 16, 1024, 1 << 24
R, Q = o.synth(n * k, 1001), o.synth(m * k, 1000)
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
buf = torch.empty(n * k, dtype=torch.float32, device=dev)


def t_ms(fn, reps=3):
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


def bare_copy():
    hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), R.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(R.nbytes), 1)


print("bare pageable hipMemcpy of 1 GiB            : %8.2f ms" % t_ms(bare_copy))
for ingest, name in ((0, "layouts built under the copy (ingest 0)"), (1, "copy, then build      (ingest 1)")):
    pkg.set_option("ingest", ingest)
    pkg.set_option("path", 2)
    held = []

    def create():
        held.append(pkg.KnnIndex(k, R))
    ms = t_ms(create, reps=3)
    ix = held[-1]
    q_ms = t_ms(lambda: ix.query(Q), reps=3)
    print("knn_index_create from host rows, %-40s: %8.2f ms   (+ first-batch query from host %.2f ms)" % (name, ms, q_ms))
    for h in held:
        h.close()
pkg.set_option("ingest", 0)
pkg.set_option("path", 0)
# page-locking the caller's buffer: cost of hipHostRegister + copy from the registered range + unregister
t0 = time.perf_counter()
rc = hip.hipHostRegister(R.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(R.nbytes), 0)
reg_ms = (time.perf_counter() - t0) * 1e3
pinned_ms = t_ms(bare_copy) if rc == 0 else float("nan")
t0 = time.perf_counter()
hip.hipHostUnregister(R.ctypes.data_as(ctypes.c_void_p))
unreg_ms = (time.perf_counter() - t0) * 1e3
print("hipHostRegister(1 GiB) rc=%d                 : %8.2f ms, copy from the registered buffer %.2f ms, unregister %.2f ms"
      % (rc, reg_ms, pinned_ms, unreg_ms))
t = t_ms(lambda: pkg.cudaCallback(k, m, n, Q, R), reps=3)
print("drop-in cudaCallback(16, 1024, 2^24) end to end : %8.2f ms" % t)
