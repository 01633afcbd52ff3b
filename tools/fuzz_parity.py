#!/usr/bin/env python3
"""Randomised differential test of the drop-in entry against the CPU oracle: random shapes,
distributions, paths, shard counts and one-shot strategies.  usage: fuzz_parity.py [cases] [seed]
(FUZZ_BIG=1: large shapes, 48 sampled queries checked per case; FUZZ_CELLS=1: the cell-pruned scan — shards of
>= 2^17 rows, k <= 32, `cells` = 1, through the drop-in entry and through a resident index queried twice)
Exit status 1 on the first mismatch (prints the case so it can be replayed)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multicore_hw2_amd as pkg              # noqa: E402
from tests.oracle_lib import Oracle           # noqa: E402


def make_data(rng, kind, rows, k):
    if kind == "uniform":
        return rng.random((rows, k), dtype=np.float32)
    if kind == "gauss":
        return rng.normal(0, 1, (rows, k)).astype(np.float32)
    if kind == "offset":
        return (rng.random((rows, k), dtype=np.float32) * np.float32(rng.choice([1e-3, 1.0, 50.0])) +
                np.float32(rng.choice([-1e4, 3.0, 1e5]))).astype(np.float32)
    if kind == "grid":            # few distinct values: many exact ties
        return rng.integers(0, 4, (rows, k)).astype(np.float32)
    if kind == "clusters":
        c = rng.random((8, k), dtype=np.float32)
        return (c[rng.integers(0, 8, rows)] + rng.normal(0, 1e-3, (rows, k))).astype(np.float32)
    if kind == "heavy":
        x = rng.normal(0, 1, (rows, k)) / np.sqrt(np.maximum(rng.random((rows, 1)), 1e-4))
        return x.astype(np.float32)
    if kind == "tight":           # clusters far tighter than the fp16 step: every row of a query's cluster is a candidate
        nc = int(rng.choice([2, 9, 64, 300]))
        c = rng.random((nc, k), dtype=np.float32)
        return (c[rng.integers(0, nc, rows)] + rng.normal(0, float(rng.choice([1e-5, 1e-4, 1e-3])), (rows, k))).astype(np.float32)
    if kind == "lowrank":         # a low-dimensional subspace: most cells empty, a few fat
        r = int(rng.integers(1, max(2, min(5, k))))
        b = rng.normal(0, 1, (r, k))
        return (rng.normal(0, 1, (rows, r)) @ b).astype(np.float32)
    if kind == "mixture":
        c = rng.random((1000, k), dtype=np.float32)
        return (c[rng.integers(0, 1000, rows)] + rng.normal(0, 0.05, (rows, k))).astype(np.float32)
    if kind == "onepoint":        # (almost) every row the same point
        x = np.tile(rng.random((1, k), dtype=np.float32), (rows, 1))
        x[rng.integers(0, rows, max(1, rows // 1000))] += np.float32(0.25)
        return x
    raise ValueError(kind)


def one_case(o, rng, case):
    k = int(rng.choice([1, 2, 3, 4, 5, 8, 15, 16, 17, 24, 32, 33, 48, 64, 100, 128, 130, 200, 256, 257, 400, 512, 513, 700]))
    m = int(rng.choice([1, 2, 3, 7, 31, 32, 33, 64, 100, 257, 1000, 1024, 2049]))
    n = int(rng.choice([1, 2, 31, 33, 1000, 4097, 65535, 65536, 70001, 200000, 600000]))
    if k * n > 40_000_000 or k * m * n > 3e11:
        n = max(1, min(n, 40_000_000 // k, int(3e11 // (k * m))))
    kind = str(rng.choice(["uniform", "gauss", "offset", "grid", "clusters", "heavy"]))
    path = int(rng.choice([0, 0, 1, 2, 3]))     # 3: the grid index where k <= 4, else the exact kernels
    shards = int(rng.choice([0, 0, 2, 5]))
    stream = int(rng.choice([0, 1, 2]))
    R = make_data(rng, kind, n, k)
    Q = make_data(rng, kind, m, k)
    if rng.random() < 0.3 and n > 4:              # some queries coincide with references
        Q[: min(m, 8)] = R[rng.integers(0, n, min(m, 8))]
    if rng.random() < 0.1:
        R[rng.integers(0, n), rng.integers(0, k)] = np.float32(rng.choice([np.nan, np.inf, -np.inf, 3e38]))
    desc = dict(case=case, k=k, m=m, n=n, kind=kind, path=path, shards=shards, stream=stream)
    pkg.set_option("path", path)
    pkg.set_option("shards", shards)
    pkg.set_option("stream", stream)
    got = pkg.cudaCallback(k, m, n, Q, R)
    want = o.v0(k, Q, R)
    if not (got == want).all():
        j = int(np.flatnonzero(got != want)[0])
        print("MISMATCH", desc, "query", j, "got", int(got[j]), "want", int(want[j]), flush=True)
        return False
    return True


def big_case(o, rng, case):
    """Large shapes (multi-piece batches, multi-chunk streams): every query goes through the GPU, a
    random sample of 48 is checked against the oracle."""
    k = int(rng.choice([3, 16, 16, 16, 20, 40]))
    m = int(rng.choice([577, 1057, 1100, 1600, 2100, 3000, 4096]))
    n = int(rng.choice([300001, 1 << 20, 3000000, 5000000]))
    if k * n > 90_000_000:
        n = 90_000_000 // k
    kind = str(rng.choice(["uniform", "gauss", "offset", "heavy"]))
    path = int(rng.choice([0, 0, 2]))
    shards = int(rng.choice([0, 0, 3]))
    stream = int(rng.choice([0, 1, 2]))
    R = make_data(rng, kind, n, k)
    Q = make_data(rng, kind, m, k)
    desc = dict(case=case, k=k, m=m, n=n, kind=kind, path=path, shards=shards, stream=stream)
    pkg.set_option("path", path)
    pkg.set_option("shards", shards)
    pkg.set_option("stream", stream)
    got = pkg.cudaCallback(k, m, n, Q, R)
    sel = rng.choice(m, 48, replace=False)
    want = o.v0(k, np.ascontiguousarray(Q[sel]), R)
    if not (got[sel] == want).all():
        j = int(np.flatnonzero(got[sel] != want)[0])
        print("MISMATCH", desc, "query", int(sel[j]), "got", int(got[sel][j]), "want", int(want[j]), flush=True)
        return False
    return True


def cells_case(o, rng, case):
    """The cell-pruned scan: cells forced on, every distribution (incl. ties, clusters, non-finite values), the
    drop-in entry and a resident index that answers two different batches (the second sees whatever the
    first left in the workspace: lists, flags, the pinned switch-off word)."""
    k = int(rng.choice([3, 5, 8, 12, 15, 16, 16, 17, 19, 20, 27, 32]))   # (round 5: 16 < k <= 32 — two K-steps per tile)
    m = int(rng.choice([1, 7, 33, 100, 257, 1000, 1024, 1300]))
    n = int(rng.choice([1 << 17, 150001, 262144, 400000, 600000, 1200000, 1 << 21]))
    kind = str(rng.choice(["uniform", "gauss", "offset", "grid", "clusters", "heavy", "tight", "lowrank", "mixture", "onepoint"]))
    shards = int(rng.choice([0, 0, 0, 2]))
    R = make_data(rng, kind, n, k)
    if kind in ("tight", "lowrank", "mixture", "onepoint"):   # queries from the same structure: rows of it, jittered
        Q = (R[rng.integers(0, n, m)] + rng.normal(0, float(rng.choice([0.0, 1e-4, 1e-2])), (m, k))).astype(np.float32)
    else:
        Q = make_data(rng, kind, m, k)
    deal, blocks = int(rng.integers(0, 3)), int(rng.integers(0, 3))
    lists, build = int(rng.integers(0, 3)), int(rng.integers(0, 3))   # round 5: who lists the cells' queries, how the layout is built
    pkg.set_option("scan_deal", deal)
    pkg.set_option("scan_blocks", blocks)
    pkg.set_option("cells_lists", lists)
    pkg.set_option("cells_build", build)
    centre = int(rng.choice([0, 1, 1, 2]))                            # round 5: per-cell frames by policy / always / never
    pkg.set_option("cells_centre", centre)
    Q2 = make_data(rng, "uniform" if kind in ("tight", "lowrank", "mixture", "onepoint") else str(rng.choice(["uniform", kind])), m, k)
    if rng.random() < 0.3:
        Q[: min(m, 8)] = R[rng.integers(0, n, min(m, 8))]
    if rng.random() < 0.15:
        R[rng.integers(0, n), rng.integers(0, k)] = np.float32(rng.choice([np.nan, np.inf, -np.inf, 3e38]))
    if rng.random() < 0.1:
        Q2[rng.integers(0, m), rng.integers(0, k)] = np.float32(rng.choice([np.nan, np.inf, 1e30]))
    desc = dict(case=case, k=k, m=m, n=n, kind=kind, shards=shards, deal=deal, blocks=blocks, lists=lists, build=build, centre=centre)
    if os.environ.get("FUZZ_VERBOSE") == "1":
        print("case", desc, "%.1f s" % time.time(), flush=True)
    pkg.set_option("cells", 1)
    pkg.set_option("shards", shards)
    pkg.set_option("path", int(rng.choice([0, 2])))
    want, want2 = o.v0(k, Q, R), o.v0(k, Q2, R)
    got = pkg.cudaCallback(k, m, n, Q, R)
    ix = pkg.KnnIndex(k, R)
    pkg.set_option("path", 0)
    got_a, got_b, got_c = ix.query(Q), ix.query(Q2), ix.query(Q)
    st = ix.last_stats()
    ix.close()
    for name, g, w in (("callback", got, want), ("index 1st", got_a, want), ("index 2nd", got_b, want2), ("index 3rd", got_c, want)):
        if not (g == w).all():
            j = int(np.flatnonzero(g != w)[0])
            print("MISMATCH", name, desc, "stats", st, "query", j, "got", int(g[j]), "want", int(w[j]), flush=True)
            return False
    return True


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    o = Oracle(os.path.join(ROOT, "oracle", "libknn_oracle.so"))
    rng = np.random.default_rng(seed)
    t0 = time.time()
    try:
        for case in range(cases):
            fn = cells_case if os.environ.get("FUZZ_CELLS") == "1" else big_case if os.environ.get("FUZZ_BIG") == "1" else one_case
            if not fn(o, rng, case):
                return 1
            if case % 25 == 24:
                print("%d cases ok, %.0f s" % (case + 1, time.time() - t0), flush=True)
    finally:
        for name in ("path", "shards", "stream", "cells", "scan_deal", "scan_blocks", "cells_lists", "cells_build"):
            pkg.set_option(name, 0)
    print("all %d cases bit-exact (seed %d)" % (cases, seed))
    return 0


if __name__ == "__main__":
    sys.exit(main())
