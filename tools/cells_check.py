"""Development check of the cell-pruned scan against the oracle (GPU box).  tools/, not the test suite:
tests/test_cells_gpu.py holds the assertions that gate the build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multicore_hw2_amd as pkg
from tests.oracle_lib import Oracle

orc = Oracle(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "libknn_oracle.so"))
rng = np.random.default_rng(7)
cases = [(16, 1024, 1 << 18), (16, 1000, 300001), (16, 33, 1 << 20), (8, 512, 1 << 19), (5, 700, 1 << 18),
         (16, 2500, 1 << 18), (12, 1, 1 << 18), (3, 256, 1 << 18)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
bad = 0
for (k, m, n) in cases:
    R = rng.random((n, k), dtype=np.float32)
    Q = rng.random((m, k), dtype=np.float32)
    os.environ["KNN_MI355X_TRACE_BUILD"] = "1"
    pkg.set_option("path", 2)
    pkg.set_option("cells", 1)
    t0 = time.time()
    idx = pkg.KnnIndex(k, R)
    t1 = time.time()
    got = idx.query(Q)
    st = idx.last_stats()
    t2 = time.time()
    got2 = idx.query(Q)
    t3 = time.time()
    want = orc.v0(k, Q, R)
    ok = np.array_equal(got, want) and np.array_equal(got2, want)
    bad += not ok
    print("k=%d m=%d n=%d: %s stats=%s build %.1f ms query %.2f / %.2f ms mism=%d" %
          (k, m, n, "OK" if ok else "MISMATCH", st, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3,
           int((got != want).sum())), flush=True)
    idx.close()
sys.exit(1 if bad else 0)
