#!/usr/bin/env python3
"""Time from an idle GPU to an idle GPU for K back-to-back batches on 4 slots / streams (k 16, m 1024, n 2^24): what the ends of a
short timed region cost on top of K steady-state steps.  usage: fill_drain.py [n]"""
import sys, time
sys.path.insert(0, ".")
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import multicore_hw2_amd as pkg
k, m = 16, 1024
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
dev = torch.device("cuda:0")
R = torch.empty(n * k, dtype=torch.float32, device=dev)
Q = torch.empty(m * k, dtype=torch.float32, device=dev)
pkg.synth_fill_device(R.data_ptr(), n * k, 1001)
pkg.synth_fill_device(Q.data_ptr(), m * k, 1000)
ix = pkg.KnnIndex(k, R.data_ptr(), n_local=n, refs_on_device=True)
B = 4
streams = [torch.cuda.Stream(device=dev) for _ in range(B)]
keys = [torch.empty(m, dtype=torch.int64, device=dev) for _ in range(B)]
outs = [torch.empty(m, dtype=torch.int32, device=dev) for _ in range(B)]
def run(K):
    for i in range(K):
        b = i % B
        ix.query_keys(m, Q.data_ptr(), keys[b].data_ptr(), stream=streams[b].cuda_stream, slot=b, init_keys=True)
        pkg.keys_to_indices(keys[b].data_ptr(), m, outs[b].data_ptr(), stream=streams[b].cuda_stream)
for _ in range(3):
    run(40); torch.cuda.synchronize()
prev = None
for K in (1, 2, 3, 4, 6, 8, 12, 16, 20, 40, 80, 200):
    ts = []
    for rep in range(30 if K <= 40 else 8):
        torch.cuda.synchronize()
        t = time.perf_counter()
        run(K)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    ts.sort()
    med = ts[len(ts) // 2]
    print("K %3d: median %.4f ms  (%.4f per step; min %.4f)" % (K, med, med / K, ts[0]), flush=True)
