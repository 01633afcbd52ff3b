#!/usr/bin/env python3
"""Would the cell-pruned scan pay for 16 < k <= 32?  CPU estimate, no GPU needed.

The pruned path rules a cell out when its lower bound LB(c, q) = sum over the cut dimensions of gap_d^2 exceeds Dup_q, an
upper bound of the query's true nearest squared distance (knn_cells.hip).  The best any implementation can do is Dup_q =
the true nearest squared distance.  Uniform data in [0,1)^k, 2^16 cells = one cut (the median) on each of 16 dimensions —
the shape the index picks at n = 2^24 — and, for k = 32, also the kindest alternative (2 bits on each of 8 dimensions).
Prints the fraction of cells a query cannot rule out with that ideal bound; the scan's work is proportional to it.
"""
import numpy as np

rng = np.random.default_rng(1)
n, mq = 1 << 22, 64
for k in (16, 20, 24, 32):
    R = rng.random((n, k), dtype=np.float32)
    Q = rng.random((mq, k), dtype=np.float32)
    best = np.full(mq, np.inf)
    rn = (R.astype(np.float64) ** 2).sum(1)
    for i in range(mq):
        d2 = rn - 2.0 * (R @ Q[i]).astype(np.float64) + float((Q[i].astype(np.float64) ** 2).sum())
        best[i] = d2.min()
    # the metric's n is 2^24: the nearest squared distance shrinks by (n'/n)^(-2/k)
    best24 = best * 4.0 ** (-2.0 / k)
    out = []
    for label, nbits in (("1 bit x 16 dims", [1] * 16), ("2 bits x 8 dims", [2] * 8)):
        frac = []
        for i in range(mq):
            lb = np.zeros(1)
            for d, b in enumerate(nbits):
                cuts = np.arange(1, 1 << b) / float(1 << b)        # quantile cuts of uniform data
                q = float(Q[i, d])
                own = int(np.searchsorted(cuts, q, side="right"))
                gaps = np.zeros(1 << b)
                for bn in range(1 << b):
                    if bn < own:
                        gaps[bn] = q - cuts[bn]                    # rows of the bin lie below cuts[bn] <= q
                    elif bn > own:
                        gaps[bn] = cuts[bn - 1] - q
                lb = (lb[:, None] + gaps[None, :] ** 2).reshape(-1)
            frac.append(float((lb <= best24[i]).mean()))
        out.append(f"{label}: {100 * np.mean(frac):5.1f} % of the cells survive")
    print(f"k = {k:2d}: nearest squared distance at n = 2^24 ~ {best24.mean():.3f} (mean of {mq} queries)   " + "   ".join(out), flush=True)
