#!/usr/bin/env python3
"""Development: per-wave stamps of the pruned scan on one of tools/distribution_check.py's distributions (stamped library build).
usage: KNN_MI355X_LIB=$PWD/tools/libknn_timeline.so python tools/clusters_timeline.py clusters64 out.npz"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multicore_hw2_amd as pkg
name, out = sys.argv[1], sys.argv[2]
k, m, n = 16, 1024, 1 << 22
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(1)
if name == "clusters64":
    c = torch.rand(64, k, device=dev, generator=g)
    r = c[torch.randint(0, 64, (n,), device=dev, generator=g)] + 1e-3 * torch.randn(n, k, device=dev, generator=g)
    q = c[torch.randint(0, 64, (m,), device=dev, generator=g)] + 1e-3 * torch.randn(m, k, device=dev, generator=g)
else:
    r = torch.rand(n, k, device=dev, generator=g); q = torch.rand(m, k, device=dev, generator=g)
q, r = q.float().contiguous(), r.float().contiguous()
keys = torch.empty(m, dtype=torch.int64, device=dev)
ix = pkg.KnnIndex(k, r.data_ptr(), n_local=n, refs_on_device=True)
for _ in range(6):
    pkg.keys_init(keys.data_ptr(), m); ix.query_keys(m, q.data_ptr(), keys.data_ptr())
torch.cuda.synchronize()
print(ix.last_stats(), ix.debug_counters())
fn = pkg.lib().knn_debug_scan_stamps; fn.argtypes = [ctypes.c_void_p]
buf = np.zeros(8192 * 5, dtype=np.uint64); assert fn(buf.ctypes.data) == 0
np.savez(out, serial=buf.reshape(8192, 5))
