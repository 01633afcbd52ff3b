"""Index build time over repeated builds in one process (allocator / pool effects): usage: build_repeat.py k n reps [cells_build]"""
import sys, time
sys.path.insert(0, ".")
import torch
import multicore_hw2_amd as pkg
k, n, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if len(sys.argv) > 4:
    pkg.set_option("cells_build", int(sys.argv[4]))
dev = torch.device("cuda:0")
R = torch.empty(n * k, dtype=torch.float32, device=dev)
pkg.synth_fill_device(R.data_ptr(), n * k, 1001)
torch.cuda.synchronize()
ts = []
for rep in range(reps):
    t = time.perf_counter()
    ix = pkg.KnnIndex(k, R.data_ptr(), n_local=n, refs_on_device=True)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t) * 1e3)
    ix.close()
print("k %d n %d cells_build %s: create ms %s" % (k, n, sys.argv[4] if len(sys.argv) > 4 else "0", " ".join("%.1f" % x for x in ts)))
