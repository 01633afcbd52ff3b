"""Oracle vs the reference's golden vectors, and the oracle's own invariants (CPU only)."""
import ctypes
import os

import numpy as np
import pytest

from tests.oracle_lib import TA_SAMPLES

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_golden_indices():
    with open(os.path.join(GOLDEN, "ta_indices.txt")) as f:
        lines = [ln for ln in f.read().splitlines() if ln and not ln.startswith("#")]
    assert len(lines) == len(TA_SAMPLES)
    return [np.array([int(t) for t in ln.split()], dtype=np.int32) for ln in lines]


def test_rand_stream_matches_golden_draws(oracle):
    with open(os.path.join(GOLDEN, "ta_first_draws.txt")) as f:
        want = [int(t) for t in f.read().splitlines()[1].split()]
    oracle.L.ta_srand(1000)
    got = [oracle.L.ta_rand() for _ in want]
    assert got == want


def test_rand_stream_matches_libc_when_glibc(oracle):
    try:
        libc = ctypes.CDLL("libc.so.6")
    except OSError:
        pytest.skip("no glibc here")
    libc.rand.restype = ctypes.c_int
    for seed in (1000, 1, 0, 12345):
        oracle.L.ta_srand(seed)
        libc.srand(seed)
        for i in range(5000):
            assert oracle.L.ta_rand() == libc.rand(), (seed, i)


def test_oracle_reproduces_reference_results_csv(oracle):
    """All 8 index lines of the reference's results.csv (committed as tests/golden/ta_indices.txt)."""
    gold = read_golden_indices()
    for i, (k, m, n, Q, R) in enumerate(oracle.ta_samples()):
        assert gold[i].size == m
        got = oracle.v0_serial(k, Q, R) if m * n <= (1 << 24) else oracle.v0(k, Q, R)
        np.testing.assert_array_equal(got, gold[i], err_msg=f"TA sample {i} {(k, m, n)}")
    assert gold[2][0] == 811  # config C1


def test_parallel_oracle_is_bit_identical_to_serial(oracle):
    rng = np.random.default_rng(7)
    k, m, n = 5, 37, 3001
    Q = rng.random((m, k), dtype=np.float32)
    R = rng.random((n, k), dtype=np.float32)
    np.testing.assert_array_equal(oracle.v0(k, Q, R, threads=4), oracle.v0_serial(k, Q, R))


def test_first_minimum_wins_on_ties(oracle):
    k = 3
    R = np.zeros((10, k), dtype=np.float32)
    R[:] = [[1, 2, 3]] * 10
    R[4] = R[7] = [0.5, 0.5, 0.5]
    Q = np.array([[0.5, 0.5, 0.5], [1, 2, 3]], dtype=np.float32)
    np.testing.assert_array_equal(oracle.v0_serial(k, Q, R), [4, 0])


def test_nan_never_wins_and_all_inf_gives_zero(oracle):
    k = 2
    R = np.array([[np.nan, 0], [3, 4], [np.nan, np.nan], [1, 1]], dtype=np.float32)
    Q = np.array([[0, 0]], dtype=np.float32)
    assert oracle.v0_serial(k, Q, R)[0] == 3
    Rn = np.full((5, k), np.nan, dtype=np.float32)
    assert oracle.v0_serial(k, Q, Rn)[0] == 0  # nothing beats +INF -> index 0 (core.cu:39-40)
    big = np.full((3, k), 3e38, dtype=np.float32)
    assert oracle.v0_serial(k, np.array([[-3e38, -3e38]], dtype=np.float32), big)[0] == 0  # dist = +INF


def test_no_fma_in_oracle_arithmetic(oracle):
    """A case where a fused multiply-add would change the rounded sum."""
    q = np.array([1.0 + 2**-12, 1.0], dtype=np.float32)
    r = np.array([0.0, 2**-13 + 1.0], dtype=np.float32)
    d0 = np.float32(q[0] - r[0])
    d1 = np.float32(q[1] - r[1])
    want = np.float32(np.float32(np.float32(0) + np.float32(d0 * d0)) + np.float32(d1 * d1))
    assert oracle.dist2(q, r) == float(want)


@pytest.mark.parametrize("shards", [1, 2, 3, 4, 8])
def test_min_over_shard_keys_equals_v0(oracle, shards):
    """The multi-GPU scheme (partition of core.cu:875-883 + packed-key min) on the CPU."""
    from multicore_hw2_amd import shard_bounds, KEY_INIT
    rng = np.random.default_rng(shards)
    k, m, n = 3, 64, 4099
    Q = rng.random((m, k), dtype=np.float32)
    R = rng.random((n, k), dtype=np.float32)
    R[100] = R[3000] = R[4098]  # cross-shard exact ties
    Q[0] = R[100]
    keys = np.full(m, KEY_INIT, dtype=np.uint64)
    for lo, hi in shard_bounds(n, shards):
        if hi > lo:
            keys = np.minimum(keys, oracle.v0_keys(k, Q, R[lo:hi], base=lo))
    np.testing.assert_array_equal((keys & np.uint64(0xFFFFFFFF)).astype(np.int32), oracle.v0_serial(k, Q, R))
    assert (keys[0] & np.uint64(0xFFFFFFFF)) == 100


def test_synth_fill_is_uniform_unit_interval_and_counter_based(oracle):
    a = oracle.synth(1 << 16, 1001)
    assert a.min() >= 0.0 and a.max() < 1.0
    assert abs(float(a.mean()) - 0.5) < 0.01
    b = oracle.synth(1000, 1001, first=5000)
    np.testing.assert_array_equal(b, a[5000:6000])


def test_oracle_reproduces_the_reference_distance_lines(oracle):
    """results.csv's even lines ('%.3f' of sqrtf(v0 distance), main.cu:16-25) for TA samples 2-7, from the
    oracle's inputs, indices and fp32 arithmetic (samples 0-1 in the reference file are use-after-free
    values: SURVEY §4)."""
    with open(os.path.join(GOLDEN, "ta_distances.txt")) as f:
        gold = [ln.split() for ln in f.read().splitlines() if ln and not ln.startswith("#")]
    assert len(gold) == len(TA_SAMPLES)
    for i, (k, m, n, Q, R) in enumerate(oracle.ta_samples()):
        if i < 2:
            continue
        idx = oracle.v0_serial(k, Q, R)
        Qm, Rm = Q.reshape(m, k), R.reshape(n, k)
        mine = ["%.3f" % float(np.sqrt(np.float32(oracle.dist2(Qm[j], Rm[idx[j]])), dtype=np.float32)) for j in range(m)]
        assert mine == gold[i], f"sample {i}"
