#!/usr/bin/env python3
"""Development aid: one batch through the cell-pruned path with KNN_MI355X_SWEEP_STAMPS=1 (per-phase wall clock of the
sweep kernel's waves).  usage: sweep_stamps.py n [k] [m]"""
import os, sys
os.environ["KNN_MI355X_SWEEP_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multicore_hw2_amd as pkg
n = int(sys.argv[1]); k = int(sys.argv[2]) if len(sys.argv) > 2 else 16; m = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = torch.device("cuda:0")
r = torch.empty(n * k, dtype=torch.float32, device=dev); q = torch.empty(m * k, dtype=torch.float32, device=dev)
pkg.synth_fill_device(r.data_ptr(), n * k, 1001); pkg.synth_fill_device(q.data_ptr(), m * k, 1000)
pkg.set_option("cells", 1)
ix = pkg.KnnIndex(k, r.data_ptr(), n_local=n, refs_on_device=True)
keys = torch.empty(m, dtype=torch.int64, device=dev)
for i in range(3):
    sys.stderr.write("--- batch %d\n" % i)
    ix.query_keys(m, q.data_ptr(), keys.data_ptr(), init_keys=True)
    torch.cuda.synchronize()
print(ix.last_stats(), ix.debug_counters())
