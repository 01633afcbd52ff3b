"""Index build phases (KNN_MI355X_TRACE_BUILD=1 prints them): usage on the GPU box:
   KNN_MI355X_TRACE_BUILD=1 python tools/build_trace.py"""
import sys, time
sys.path.insert(0, ".")
import torch
import multicore_hw2_amd as pkg
k, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 1 << 24)
dev = torch.device("cuda:0")
R = torch.empty(n * k, dtype=torch.float32, device=dev)
pkg.synth_fill_device(R.data_ptr(), n * k, 1001)
torch.cuda.synchronize()
for rep in range(3):
    t = time.perf_counter()
    ix = pkg.KnnIndex(k, R.data_ptr(), n_local=n, refs_on_device=True)
    torch.cuda.synchronize()
    print("create %.3f ms" % ((time.perf_counter() - t) * 1e3), file=sys.stderr)
    ix.close()
