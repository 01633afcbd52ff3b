"""Cell-range shards (round 4; include/knn_mi355x.h section 2b) against the CPU oracle.

The reference splits the reference set by index range (core.cu:875-883); a resident caller with several GPUs may instead
split ONE global grid's cell codes into contiguous ranges, one per rank (knn_geom_*, knn_index_create_sharded), so that the
ranks' scans add up to one GPU's.  What is under test: every rank's keys carry GLOBAL row numbers when its batch ends, the
minimum over the ranks' keys is v0's answer (reference core.cu:27-62) — ties across two shards included — whatever the
distribution does to the ranks' shares, with and without the replicated seed layer.  All shards of a set live on the one
GPU a test box has; the exchange step is a min over their key arrays (what the all-reduce does between GPUs).  Bar: bit-exact."""
import os

import numpy as np
import pytest
import torch  # imported BEFORE libknn_mi355x.so is dlopen'ed: one HIP runtime (torch's) per process

import multicore_hw2_amd as pkg

pytestmark = pytest.mark.gpu
THREADS = min(16, os.cpu_count() or 1)
DEV = None


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    global DEV
    assert os.path.exists(pkg.lib_path), "libknn_mi355x.so not built (no CPU fallback exists)"
    assert pkg.device_count() >= 1, "no GPU visible to libknn_mi355x.so"
    DEV = torch.device("cuda:0")
    yield


class Shards:
    """N cell-range shards of one reference set on one GPU."""

    def __init__(self, k, R_d, nranks, seed_tiles=0, attach=True, sample_rows=4096):
        n = R_d.shape[0]
        self.k, self.n, self.nranks = k, n, nranks
        stride = max(1, n // sample_rows)
        sample = R_d[::stride][:sample_rows].cpu().numpy()
        self.geom = pkg.KnnGeom(k, n, nranks, sample, seed_tiles)
        owner = torch.empty(n, dtype=torch.int32, device=DEV)
        self.geom.assign(R_d.data_ptr(), n, owner.data_ptr())
        torch.cuda.synchronize()
        self.owner = owner
        self.rows, self.gids, self.idx = [], [], []
        for r in range(nranks):
            g = torch.nonzero(owner == r).reshape(-1)                   # ascending global row numbers
            rows = R_d[g].contiguous()
            gids = g.to(torch.int32)
            self.rows.append(rows)
            self.gids.append(gids)
            self.idx.append(pkg.KnnIndex.sharded(self.geom, r, rows.data_ptr(), gids.data_ptr(), rows.shape[0]))
        self.layer = torch.zeros(self.geom.layer_bytes, dtype=torch.uint8, device=DEV)
        for ix in self.idx:
            ix.seed_export(self.layer.data_ptr())
        torch.cuda.synchronize()
        if attach:
            for ix in self.idx:
                ix.seed_attach(self.layer.data_ptr())

    def query(self, Q):
        Qf = np.ascontiguousarray(Q, dtype=np.float32).reshape(-1)
        m = Qf.size // self.k
        q_d = torch.from_numpy(Qf).to(DEV)
        keys = torch.empty((self.nranks, m), dtype=torch.int64, device=DEV)
        for r, ix in enumerate(self.idx):
            ix.query_keys(m, q_d.data_ptr(), keys[r].data_ptr(), init_keys=True)
        torch.cuda.synchronize()
        stats = [ix.last_stats() for ix in self.idx]
        merged = keys.min(dim=0).values            # keys < 2^63 (distance bits of a non-negative float): int64 min == unsigned min
        return (merged & 0xFFFFFFFF).to(torch.int32).cpu().numpy(), keys, stats

    def close(self):
        for ix in self.idx:
            ix.close()
        self.geom.close()


def _data(rng, kind, k, m, n):
    R = rng.random((n, k), dtype=np.float32)
    Q = rng.random((m, k), dtype=np.float32)
    if kind == "clustered":                   # 16 blobs: some ranks hold most of the rows, others few
        c = rng.random((16, k), dtype=np.float32)
        R = (c[rng.integers(0, 16, n)] + rng.normal(0, 0.02, (n, k))).astype(np.float32)
        Q = (c[rng.integers(0, 16, m)] + rng.normal(0, 0.03, (m, k))).astype(np.float32)
    elif kind == "lattice":                   # rows and queries ON the cuts, many exact ties
        R = (rng.integers(0, 5, (n, k)) * 0.25).astype(np.float32)
        Q = (rng.integers(0, 5, (m, k)) * 0.25).astype(np.float32)
    elif kind == "queries_outside":
        Q = (Q * 3.0 - 1.0).astype(np.float32)
        Q[:3] = np.float32(300.0)             # far away: the gated exact scan answers the batch on every rank
    elif kind == "copies":                    # every query is a reference row, some rows duplicated: lowest index of the copies
        src = rng.integers(0, n, m)
        Q = R[src].copy()
        R[rng.integers(0, n, 64)] = R[src[:64]]
    elif kind == "flat_top_dims":             # the dimensions that decide the rank hold ONE value: one rank has every row, the others none
        R[:, 9:] = np.float32(0.5)           # (n = 2^20: dimensions 9, 10, 11 carry the top three code bits)
        Q[:, 9:] = (0.5 + rng.normal(0, 0.05, (m, k - 9))).astype(np.float32)
    return np.ascontiguousarray(Q), np.ascontiguousarray(R)


@pytest.mark.parametrize("k,nranks,n,kind", [
    (16, 8, (1 << 20) + 777, "uniform"), (16, 4, 1 << 20, "clustered"), (16, 2, (1 << 19) + 5, "lattice"),
    (8, 3, (1 << 19) + 4099, "uniform"), (5, 8, 1 << 20, "queries_outside"), (12, 5, (1 << 20) + 1, "copies"),
    (16, 8, 1 << 20, "flat_top_dims")])
def test_cell_range_shards_fold_to_the_oracle(oracle, k, nranks, n, kind):
    m = 700
    rng = np.random.default_rng(k * 131 + nranks + len(kind))
    Q, R = _data(rng, kind, k, m, n)
    want = oracle.v0(k, Q, R, threads=THREADS)
    sh = Shards(k, torch.from_numpy(R).to(DEV), nranks)
    try:
        got, keys, stats = sh.query(Q)
        again, _, _ = sh.query(Q)
        assert int(sum(r.shape[0] for r in sh.rows)) == n             # the ranges partition the set
        np.testing.assert_array_equal(got, want, err_msg=f"{kind} k={k} N={nranks} stats={stats}")
        np.testing.assert_array_equal(again, got)
        for st, rows in zip(stats, sh.rows):
            assert st[0] == (4 if rows.shape[0] else 0), st           # the pruned path (0: a rank without rows answers nothing)
        if kind == "flat_top_dims":
            assert sorted(r.shape[0] for r in sh.rows)[-2] == 0       # ... and here all ranks but one are such
        # every rank's own answer is a row of ITS range (or nothing: (+INF, 0))
        for r in range(nranks):
            kr = keys[r].cpu().numpy()
            found = (kr >> 32) != 0x7F800000
            assert (sh.owner.cpu().numpy()[(kr[found] & 0xFFFFFFFF)] == r).all()
    finally:
        sh.close()


def test_a_tie_across_two_shards_goes_to_the_lower_global_index(oracle):
    """Two rows at exactly the same distance from a query, on either side of the cut of the dimension that decides the rank:
    v0 keeps the first (core.cu:50-54).  Each rank only sees its own row; the minimum over the ranks' packed keys decides."""
    k, m, n, nranks = 16, 64, 1 << 20, 8
    rng = np.random.default_rng(5)
    R = rng.random((n, k), dtype=np.float32)
    Q = rng.random((m, k), dtype=np.float32)
    planted = []
    # (n = 2^20: a grid of 2^12 cells, one cut on each of dimensions 0 .. 11; the top three code bits — dimensions 9, 10, 11 —
    # decide the rank at N = 8)
    for j, (lo_side_first, d) in enumerate([(True, 11), (False, 11), (True, 10), (False, 9)]):
        q = Q[j]
        q[d] = 0.5
        i1, i2 = 1000 + 37 * j, 900000 + 41 * j
        a, b = q.copy(), q.copy()
        a[d], b[d] = (0.25, 0.75) if lo_side_first else (0.75, 0.25)      # |0.5 - x| = 0.25 exactly on both sides
        R[i1], R[i2] = a, b
        planted.append((j, i1, i2))
    want = oracle.v0(k, Q, R, threads=THREADS)
    sh = Shards(k, torch.from_numpy(R).to(DEV), nranks)
    try:
        got, keys, _ = sh.query(Q)
        np.testing.assert_array_equal(got, want)
        own = sh.owner.cpu().numpy()
        for j, i1, i2 in planted:
            assert want[j] == i1 and own[i1] != own[i2], (j, want[j], own[i1], own[i2])   # the tie really spans two ranks
            k1, k2 = int(keys[own[i1]][j]), int(keys[own[i2]][j])
            assert k1 >> 32 == k2 >> 32 and (k1 & 0xFFFFFFFF, k2 & 0xFFFFFFFF) == (i1, i2)   # equal distance bits, own rows
    finally:
        sh.close()


def test_shards_answer_without_the_seed_layer_and_with_other_layer_depths(oracle):
    """The layer only tightens the bound: without it (a rank bounds from its own cells, or from a strided sample of its
    layout when the query's seed cells are all elsewhere) and with 1 or 4 tiles per cell the answers are the same."""
    k, m, n, nranks = 16, 500, (1 << 20) + 333, 4
    rng = np.random.default_rng(11)
    Q, R = _data(rng, "uniform", k, m, n)
    want = oracle.v0(k, Q, R, threads=THREADS)
    R_d = torch.from_numpy(R).to(DEV)
    records = {}
    for label, kw in (("no layer", dict(attach=False)), ("T=1", dict(seed_tiles=1)), ("T=2", dict()), ("T=4", dict(seed_tiles=4))):
        sh = Shards(k, R_d, nranks, **kw)
        try:
            got, _, stats = sh.query(Q)
            np.testing.assert_array_equal(got, want, err_msg=label)
            records[label] = sum(st[1] for st in stats)
        finally:
            sh.close()
    assert records["T=2"] <= records["no layer"]      # a tighter bound lets fewer candidates through


def test_sharded_index_refuses_what_it_cannot_answer(oracle):
    k, n, nranks = 16, 1 << 20, 4
    rng = np.random.default_rng(3)
    R_d = torch.from_numpy(rng.random((n, k), dtype=np.float32)).to(DEV)
    sh = Shards(k, R_d, nranks)
    try:
        # rows of another rank's range
        with pytest.raises(pkg.KnnError, match="outside rank"):
            pkg.KnnIndex.sharded(sh.geom, 0, sh.rows[1].data_ptr(), sh.gids[1].data_ptr(), sh.rows[1].shape[0])
        # global numbers that do not ascend
        bad = sh.gids[0].clone()
        bad[5], bad[6] = sh.gids[0][6], sh.gids[0][5]
        with pytest.raises(pkg.KnnError, match="ascending"):
            pkg.KnnIndex.sharded(sh.geom, 0, sh.rows[0].data_ptr(), bad.data_ptr(), sh.rows[0].shape[0])
        # folding into keys somebody else wrote
        keys = torch.empty(8, dtype=torch.int64, device=DEV)
        q = torch.zeros(8 * k, dtype=torch.float32, device=DEV)
        with pytest.raises(pkg.KnnError, match="INIT_KEYS"):
            sh.idx[0].query_keys(8, q.data_ptr(), keys.data_ptr(), init_keys=False)
        # a layer whose parts were never exported
        with pytest.raises(pkg.KnnError, match="header"):
            sh.idx[0].seed_attach(torch.full((sh.geom.layer_bytes,), 0xFF, dtype=torch.uint8, device=DEV).data_ptr())
    finally:
        sh.close()


def test_c3_as_eight_cell_range_shards_every_query_against_the_oracle(oracle):
    """The metric's shape (k 16, m 1024, n 2^24) as the eight cell-range shards of an 8-GPU run, all on one GPU: every one of
    the 1024 answers against the oracle over all 2^24 rows, and the indices the batch's last kernel writes against the keys."""
    k, m, n, nranks = 16, 1024, 1 << 24, 8
    R_d = torch.empty(n * k, dtype=torch.float32, device=DEV)
    pkg.synth_fill_device(R_d.data_ptr(), n * k, 1001)
    torch.cuda.synchronize()
    Q = oracle.synth(m * k, 1000)
    # (the cuts are sample quantiles: the ranks' shares are as even as the sample's medians are sharp — 2^16 sample rows put
    # the three cuts that decide the rank within 0.2 % of the true medians; with 4096 the largest share was 8.7 % over)
    sh = Shards(k, R_d.reshape(n, k), nranks, sample_rows=1 << 16)
    try:
        assert sh.geom.bits == 16 and sh.geom.cells_per_rank == 8192
        shares = [r.shape[0] for r in sh.rows]
        assert max(shares) < 1.03 * n / nranks, shares               # uniform data: equal ranges hold equal shares
        got, _, stats = sh.query(Q)
        want = oracle.v0(k, Q, oracle.synth(n * k, 1001), threads=THREADS)
        np.testing.assert_array_equal(got, want)
        for st in stats:
            assert st[0] == 4 and st[2] == 0, st
    finally:
        sh.close()
