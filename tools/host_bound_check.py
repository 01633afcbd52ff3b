#!/usr/bin/env python3
"""Is the pipelined step bound by the host's launch rate?  Enqueues 600 steps (keys init -> query -> unpack, three
streams / workspace slots) and reports how long the host needed to enqueue them against how long the GPU needed to
finish them.  Round 2, one MI355X: n = 2^21: host 23.6 us per step, GPU 47.3; n = 2^24: host 27.7, GPU 119.9 —
the host's launch rate is not what limits the step at either size."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import multicore_hw2_amd as pkg
k, m = 16, 1024
for n in (1 << 21, 1 << 24):
    dev = torch.device("cuda:0")
    r_d = torch.empty(n * k, dtype=torch.float32, device=dev); q_d = torch.empty(m * k, dtype=torch.float32, device=dev)
    pkg.synth_fill_device(r_d.data_ptr(), n * k, 1001); pkg.synth_fill_device(q_d.data_ptr(), m * k, 1000)
    torch.cuda.synchronize()
    ix = pkg.KnnIndex(k, r_d.data_ptr(), n_local=n, refs_on_device=True)
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    keys = [torch.empty(m, dtype=torch.int64, device=dev) for _ in range(3)]
    outs = [torch.empty(m, dtype=torch.int32, device=dev) for _ in range(3)]
    def step(i):
        b = i % 3; s = streams[b].cuda_stream
        pkg.keys_init(keys[b].data_ptr(), m, stream=s)
        ix.query_keys(m, q_d.data_ptr(), keys[b].data_ptr(), stream=s, slot=b)
        pkg.keys_to_indices(keys[b].data_ptr(), m, outs[b].data_ptr(), stream=s)
    for i in range(60): step(i)
    torch.cuda.synchronize()
    N = 600
    t0 = time.perf_counter()
    for i in range(N): step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("n %d: host enqueue %.1f us/step, total %.1f us/step (GPU still busy %.1f us/step after the last enqueue)" % (n, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6, (t2 - t1) / N * 1e6), flush=True)
    ix.close()
