"""The global grid of a cell-range sharded set (knn_geom_*, include/knn_mi355x.h section 2b): host arithmetic, no GPU.
Every rank must derive the SAME grid from the same sample, the ranks' ranges must partition the cell codes in granules
the kernels rely on (whole entries of the high pruning table, whole blocks of the match pass), and sets that do not suit
must be refused (the caller then shards by index range, the reference's way: core.cu:875-883)."""
import numpy as np
import pytest

import multicore_hw2_amd as pkg


def _sample(k, rows=4096, seed=1):
    return np.random.default_rng(seed).random((rows, k), dtype=np.float32)


@pytest.mark.parametrize("k,n,nranks", [(16, 1 << 24, 8), (16, 1 << 24, 3), (16, 1 << 27, 8), (16, 1 << 20, 7), (12, (1 << 20) + 1, 5),
                                        (8, 1 << 19, 3), (5, 1 << 20, 8), (16, 1 << 24, 1)])
def test_rank_ranges_partition_the_grid_in_whole_granules(k, n, nranks):
    g = pkg.KnnGeom(k, n, nranks, _sample(k))
    try:
        assert g.ncells == 1 << g.bits and 6 <= g.sa <= 8
        firsts = [g.first_cell(r) for r in range(nranks + 1)]
        assert firsts[0] == 0 and firsts[-1] == g.ncells and g.first_cell(nranks + 5) == g.ncells
        sizes = np.diff(firsts)
        assert (sizes >= 512).all() and (sizes <= 65536).all() and sizes.max() == g.cells_per_rank
        assert all(f % (1 << g.sa) == 0 for f in firsts)          # whole entries of the high table, whole 64-cell blocks
        assert sizes.max() - sizes.min() <= (1 << g.sa)           # as even as the granule allows
        # at least 144 rows per cell on average (the single index's rule) and never more than 4 bits per dimension
        assert n >> g.bits >= 144 and g.bits <= 4 * k
        # the seed layer: header + (1 KiB of fragments + 128 B of split norms) per tile, the same room in every part
        assert g.part_bytes == 256 + g.cells_per_rank * g.seed_tiles * 1152 and g.layer_bytes == nranks * g.part_bytes
    finally:
        g.close()


def test_finer_grid_than_one_gpu_could_hold_when_the_ranks_allow_it():
    """One index holds at most 2^16 cells; eight ranks of C4's 2^27 rows hold 2^16 each of a 2^19-cell grid."""
    s = _sample(16)
    one, eight = pkg.KnnGeom(16, 1 << 27, 1, s), pkg.KnnGeom(16, 1 << 27, 8, s)
    assert one.bits == 16 and eight.bits == 19 and eight.cells_per_rank == 1 << 16
    one.close()
    eight.close()


def test_sets_that_do_not_suit_are_refused():
    for k, n, nranks in ((17, 1 << 24, 8), (16, 1 << 18, 8), (16, 1 << 24, 0), (16, 1 << 24, 65)):   # k > 16, too few rows, ranks
        with pytest.raises(pkg.KnnError):
            pkg.KnnGeom(k, n, nranks, _sample(k))
    with pytest.raises(pkg.KnnError):                       # too few sample rows
        pkg.KnnGeom(16, 1 << 24, 8, _sample(16, rows=10))
    bad = _sample(16)
    bad[7, 3] = np.nan
    with pytest.raises(pkg.KnnError):                       # a sample that is not finite
        pkg.KnnGeom(16, 1 << 24, 8, bad)
    with pytest.raises(pkg.KnnError):
        pkg.KnnGeom(16, 1 << 24, 8, _sample(16), seed_tiles=9)
