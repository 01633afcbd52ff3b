"""CPU restatement of the grid index's stop rule (multicore_hw2_amd/csrc/knn_grid.hip) for ONE live axis.

The rows' cells come from fp32 arithmetic, t = fl(fl(x - lo) * inv_w); the stop rule reasons about cell faces in
double.  The distance between the two grows with the number of cells per axis: with one live axis the grid may
have hundreds of thousands of cells, and the round-2 allowance (1e-3 of a cell per face) was short from ~4000
cells up — the search stopped before the ring that held the true nearest row (knn_grid.hip:313).
Here: numpy emulation of cell assignment + ring walk + stop rule on 1-D data whose box is not anchored at 0,
(a) with the shipped allowance and cell cap: never a wrong answer; (b) with the round-2 constants: wrong
answers show up, i.e. the test can see the bug it guards against.
"""
import numpy as np
import pytest

F32 = np.float32


def build(x, g):
    """knn_grid_build for one live axis: returns (lo, inv_w [fp32], w [double], cell start offsets, rows sorted)."""
    lo, hi = float(x.min()), float(x.max())
    inv_w = F32(g / (hi - lo))
    w = (hi - lo) / g
    t = (x - F32(lo)).astype(F32) * inv_w          # fp32 subtract, fp32 multiply: grid_cell_of
    cell = np.where(t >= 0, np.where(t < F32(g), t.astype(np.int64), g - 1), 0)
    order = np.argsort(x, kind="stable")           # the cell function is monotone: x order is cell order too
    xs = x[order]
    assert (np.diff(cell[order]) >= 0).all()
    start = np.searchsorted(cell[order], np.arange(g + 1))
    return lo, inv_w, w, start, xs


def ring_search(q, lo, inv_w, w, g, start, xs, slack, rmax=64):
    """knn_grid_query_kernel<1>: rings 0..1 first, then ring by ring; returns the squared distance (fp32, v0
    arithmetic) of the best row seen when the stop rule fires (or the walk ends)."""
    t = (q - F32(lo)).astype(F32) * inv_w
    c = np.where(t >= 0, np.where(t < F32(g), t.astype(np.int64), g - 1), 0)
    best = np.full(q.shape, np.inf, dtype=F32)
    done = np.zeros(q.shape, dtype=bool)
    for r in range(1, rmax + 1):
        a = start[np.clip(c - r, 0, g)]
        b = start[np.clip(c + r + 1, 0, g)]
        # nearest row inside xs[a:b] (rows in cell order are in x order: the cell function is monotone)
        pos = np.searchsorted(xs, q)
        cand = np.full(q.shape, np.inf, dtype=F32)
        for p in (np.clip(pos - 1, a, np.maximum(b - 1, a)), np.clip(pos, a, np.maximum(b - 1, a))):
            ok = b > a
            diff = (q - xs[np.minimum(p, len(xs) - 1)]).astype(F32)
            d2 = (diff * diff).astype(F32)
            cand = np.where(ok & (d2 < cand), d2, cand)
        best = np.where(~done & (cand < best), cand, best)
        lb = np.full(q.shape, np.inf)
        covers = np.ones(q.shape, dtype=bool)
        low = c - r > 0
        covers &= ~low
        lb = np.where(low, np.minimum(lb, q.astype(np.float64) - (lo + (c - r) * w) - slack), lb)
        high = c + r < g - 1
        covers &= ~high
        lb = np.where(high, np.minimum(lb, (lo + (c + r + 1) * w) - q.astype(np.float64) - slack), lb)
        stop = covers | ((lb > 0) & (best.astype(np.float64) < lb * lb * (1.0 - 1e-6)))
        done |= stop
        if done.all():
            break
    return best, done


def true_nn_d2(q, xs_sorted):
    pos = np.searchsorted(xs_sorted, q)
    best = np.full(q.shape, np.inf, dtype=F32)
    for p in (np.clip(pos - 1, 0, len(xs_sorted) - 1), np.clip(pos, 0, len(xs_sorted) - 1)):
        diff = (q - xs_sorted[p]).astype(F32)
        d2 = (diff * diff).astype(F32)
        best = np.minimum(best, d2)
    return best


def make_data(n, seed, lo, hi):
    rng = np.random.default_rng(seed)
    x = (lo + (hi - lo) * rng.random(n)).astype(F32)
    # a sparse stretch near the far end of the box (where t = (x - lo) * inv_w is largest and its rounding coarsest):
    # cells there are mostly empty and the walk needs several rings
    mid = lo + 0.9 * (hi - lo)
    keep = (np.abs(x - F32(mid)) > 0.05 * (hi - lo)) | (rng.random(n) < 0.02)
    return x[keep]


@pytest.mark.parametrize("lo,hi", [(-1.0, 1.0), (3.7, 5.7), (1000.25, 1003.0)])
def test_stop_rule_with_the_shipped_allowance_never_stops_early(lo, hi):
    x = make_data(1 << 22, 11, lo, hi)
    n = len(x)
    g = min(int(np.floor(n / 3.0)), 1 << 19)                 # knn_grid_build: ~3 rows per cell, cap for one live axis
    glo, inv_w, w, start, xs = build(x, g)
    slack = (1e-3 + g * 2.0 ** -21) * w                     # GridGeom::slack
    rng = np.random.default_rng(5)
    q = (lo + (hi - lo) * rng.random(300000)).astype(F32)
    got, done = ring_search(q, glo, inv_w, w, g, start, xs, slack)
    want = true_nn_d2(q, np.sort(x))
    assert (got[done] == want[done]).all()


def face_violations(x, g, slack_of):
    """The invariant the stop rule leans on: a row assigned to cell c lies in [lo + c w - slack, lo + (c+1) w + slack).
    Returns how many rows break it."""
    lo, inv_w, w, _, _ = build(x, g)
    t = (x - F32(lo)).astype(F32) * inv_w
    cell = np.where(t >= 0, np.where(t < F32(g), t.astype(np.int64), g - 1), 0)
    xd = x.astype(np.float64)
    slack = slack_of(g, w)
    below = (lo + cell * w) - xd > slack
    above = (xd - (lo + (cell + 1) * w) >= slack) & (cell < g - 1)
    return int(below.sum() + above.sum())


@pytest.mark.parametrize("lo,hi", [(-1.0, 1.0), (3.7, 5.7), (1000.25, 1003.0), (-7e-3, 9e-3)])
def test_every_row_lies_within_the_allowance_of_its_cell(lo, hi):
    x = make_data(1 << 22, 3, lo, hi)
    g = min(int(np.floor(len(x) / 3.0)), 1 << 19)
    assert face_violations(x, g, lambda g_, w: (1e-3 + g_ * 2.0 ** -21) * w) == 0


def test_round2_allowance_is_caught_by_this_test():
    """1e-3 of a cell with up to 2^22 cells on the axis (the round-2 constants): rows sit outside their cell's faces
    by more than the allowance — the premise of the stop rule is false, which is how it came to stop early."""
    x = make_data(1 << 22, 3, -1.0, 1.0)
    g = min(int(np.floor(len(x) / 3.0)), 1 << 22)
    assert face_violations(x, g, lambda g_, w: 1e-3 * w) > 1000
