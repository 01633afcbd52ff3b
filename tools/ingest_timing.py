#!/usr/bin/env python3
"""what an index created from HOST rows costs (C3: k 16, n 2^24 = 1 GiB), next to the
bare pageable copy, and whether page-locking the caller's buffer first (hipHostRegister) pays.
usage: python tools/ingest_timing.py  -> prints one line per measurement (ms): EVERY repetition, not only the best, and — with
KNN_MI355X_TRACE_BUILD=1, which this script sets — the library's own per-stage laps of every index creation ("[knn ingest] ..."
lines on stderr: host sample + box, allocations, copy call returned, streams drained).  Round 4's final collection read 54.6 ms
for the default ingest on a box whose bare copy took 19.3: best-of-3 with no stage record could not say which stage it was."""
import ctypes
import os
import sys
import time

os.environ.setdefault("KNN_MI355X_TRACE_BUILD", "1")

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import multicore_hw2_amd as pkg
from tests.oracle_lib import Oracle

for _opt in os.environ.get("KNN_IT_OPTS", "").split(","):   # e.g. KNN_IT_OPTS=cells_centre=2
    if _opt:
        pkg.set_option(_opt.split("=")[0], int(_opt.split("=")[1]))

o = Oracle(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "libknn_oracle.so"))
k, m, n = 16, 1024, 1 << 24
R, Q = o.synth(n * k, 1001), o.synth(m * k, 1000)
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
buf = torch.empty(n * k, dtype=torch.float32, device=dev)


last_reps = []


def t_ms(fn, reps=3):
    """best of `reps`; all of them are kept in last_reps and printed by the caller"""
    del last_reps[:]
    for i in range(reps):
        torch.cuda.synchronize()
        sys.stderr.flush()
        sys.stderr.write("-- repetition %d\n" % i)
        sys.stderr.flush()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        last_reps.append((time.perf_counter() - t0) * 1e3)
    return min(last_reps)


def reps_str():
    return "[" + " ".join("%.2f" % v for v in last_reps) + "]"


def bare_copy():
    hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), R.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(R.nbytes), 1)


print("bare pageable hipMemcpy of 1 GiB            : %8.2f ms  all repetitions %s" % (t_ms(bare_copy, reps=5), reps_str()), flush=True)


def bare_copy_async():
    hip.hipMemcpyAsync(ctypes.c_void_p(buf.data_ptr()), R.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(R.nbytes), 1, ctypes.c_void_p(0))
    hip.hipDeviceSynchronize()


print("bare pageable hipMemcpyAsync of 1 GiB + sync : %8.2f ms  all repetitions %s   (what the ingest's copy stream issues)"
      % (t_ms(bare_copy_async, reps=5), reps_str()), flush=True)
for ingest, name in ((0, "layouts built under the copy (ingest 0)"), (1, "copy, then build      (ingest 1)")):
    pkg.set_option("ingest", ingest)
    pkg.set_option("path", 2)
    held = []

    def create():
        # (ONE index alive at a time.  Rounds 2-4 kept every repetition's index: from the third on the library's buffer pool had
        # no free block left for the 1 GiB of rows, the two-pass build's 1.2 GB of scratch and the 0.6 GB of layouts, and the
        # create paid fresh hipMallocs of that size — 32 + 15 ms in the laps "cell codes + counts" and "allocations":
        # profiles/r05_ingest_timing.txt, and the 54.6 ms line of round 4's final collection)
        while held:
            held.pop().close()
        held.append(pkg.KnnIndex(k, R))
    ms = t_ms(create, reps=5)
    create_reps = reps_str()
    ix = held[-1]
    q_ms = t_ms(lambda: ix.query(Q), reps=3)
    print("knn_index_create from host rows, %-40s: %8.2f ms  all repetitions %s   (+ first-batch query from host %.2f ms)"
          % (name, ms, create_reps, q_ms), flush=True)
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
print("drop-in cudaCallback(16, 1024, 2^24) end to end : %8.2f ms  all repetitions %s" % (t, reps_str()))
