"""The pruning argument of the cell-pruned scan (multicore_hw2_amd/csrc/knn_cells.hip), restated in numpy — no GPU.
Rows are binned per dimension by `x >= cut` compares; for a query q the squared gap to a row's bin, summed over
the dimensions, must never exceed the squared distance to the row — for rows and queries ON the cuts, duplicated
cuts, values far from the origin, and under the kernel's own rounding (terms rounded DOWN to float32)."""
import numpy as np
import pytest


def assign_bins(x, cuts):
    """bin b <=> cuts[b-1] <= x < cuts[b] (ascending cuts; NaN -> bin 0), as cell_bin() does."""
    return (x[:, None] >= cuts[None, :]).sum(axis=1)


def gap_table(q, cuts):
    """gap[b] = distance from q to bin b's interval, 0 inside (as knn_cells_seed_kernel computes it, in float64)."""
    nb = len(cuts) + 1
    g = np.zeros(nb)
    for b in range(nb):
        if b > 0 and cuts[b - 1] > q:
            g[b] = float(cuts[b - 1]) - float(q)
        if b < nb - 1 and q > cuts[b]:
            g[b] = float(q) - float(cuts[b])
    return g


def round_down_f32(v):
    f = np.float32(v)
    return np.where(f.astype(np.float64) > v, np.nextafter(f, np.float32(-np.inf)), f).astype(np.float32)


@pytest.mark.parametrize("kind", ["uniform", "lattice", "offset", "duplicate_cuts", "wide_range"])
@pytest.mark.parametrize("k,bins", [(16, 2), (5, 8), (3, 16)])
def test_cell_lower_bound_never_exceeds_the_distance(kind, k, bins):
    rng = np.random.default_rng(hash((kind, k, bins)) % (1 << 32))
    n, m = 4000, 64
    R = rng.random((n, k)).astype(np.float32)
    Q = rng.random((m, k)).astype(np.float32)
    if kind == "lattice":
        R = (rng.integers(0, 5, (n, k)) * 0.25).astype(np.float32)
        Q = (rng.integers(0, 5, (m, k)) * 0.25).astype(np.float32)
    elif kind == "offset":
        R += np.float32(4096)
        Q += np.float32(4096)
    elif kind == "wide_range":
        R = (R * np.float32(1e6) - np.float32(5e5)).astype(np.float32)
        Q = (Q * np.float32(3e6) - np.float32(1.5e6)).astype(np.float32)
    cuts = [np.sort(R[:: max(1, n // 256), d])[[(j * (len(R[:: max(1, n // 256)]))) // bins for j in range(1, bins)]]
            for d in range(k)]
    if kind == "duplicate_cuts":
        cuts = [np.sort(np.concatenate([c[: bins // 2], c[: bins - 1 - bins // 2]])).astype(np.float32) for c in cuts]
    rbin = np.stack([assign_bins(R[:, d], cuts[d]) for d in range(k)], axis=1)          # n x k
    R64, Q64 = R.astype(np.float64), Q.astype(np.float64)
    for qi in range(m):
        tabs = [gap_table(Q[qi, d], cuts[d]) for d in range(k)]
        # the kernel's arithmetic: squared terms rounded down to float32, summed in float64, rounded down again
        terms = np.stack([round_down_f32(tabs[d][rbin[:, d]] ** 2) for d in range(k)], axis=1).astype(np.float64)
        lb = round_down_f32(terms.sum(axis=1)).astype(np.float64)
        d2 = ((R64 - Q64[qi]) ** 2).sum(axis=1)                                        # exact enough: float64 of floats
        assert (lb <= d2 * (1 + 1e-12) + 1e-300).all(), (kind, qi, float((lb - d2).max()))
        # and the bound is not vacuous: rows of the query's own cell have bound 0
        qbin = np.array([assign_bins(Q[qi:qi + 1, d], cuts[d])[0] for d in range(k)])
        own = (rbin == qbin).all(axis=1)
        assert (lb[own] == 0).all()


def test_cell_code_layout_matches_the_build_rules():
    """bits per dimension = B / k rounded up for the first B % k dimensions; the low pruning table ends at the last
    dimension boundary at or below bit 8 and must cover >= 64 entries (knn_cells_build)."""
    def plan(k, bits):
        nb = [bits // k + (1 if d < bits % k else 0) for d in range(k)]
        pos, sa, shift = 0, 0, []
        for d in range(k):
            shift.append(pos)
            if pos <= 8:
                sa = pos
            pos += nb[d]
        if pos <= 8:
            sa = pos
        return nb, shift, sa
    for k in range(3, 17):
        for bits in range(9, min(16, 4 * k) + 1):
            nb, shift, sa = plan(k, bits)
            assert sum(nb) == bits and max(nb) <= 4
            assert sa <= 8 and bits - sa <= 10
            if sa >= 6:                                   # else the build declines the cells
                low = [d for d in range(k) if nb[d] and shift[d] < sa]
                assert sum(nb[d] for d in low) == sa      # the low table holds whole dimensions only


def test_norm_kept_as_two_fp16_halves_is_within_the_allowance_rho_carries():
    """knn_cells.hip pack_norm22: N ~ fp16(N) + fp16((N - fp16(N)) * 2^11) * 2^-11, summed exactly by the matrix core.
    knn_bound_consts allows 2^-21 Nmax + 2^-24 for it; restated in numpy over the range a norm can take (k <= 16
    coordinates of magnitude <= 1), including the case where the matrix core flushes a subnormal low half."""
    rng = np.random.default_rng(4)
    n = np.concatenate([rng.random(200000) * 16.0, rng.random(200000) * 1e-3, 2.0 ** -rng.integers(0, 40, 1000),
                        np.array([0.0, 16.0, 1.0, 2.0 ** -14, 2.0 ** -24])]).astype(np.float32)
    hi = n.astype(np.float16)
    rem = (n - hi.astype(np.float32)).astype(np.float32)
    assert (rem.astype(np.float64) == n.astype(np.float64) - hi.astype(np.float64)).all()   # the subtraction is exact
    mid = (rem * np.float32(2048.0)).astype(np.float16)
    back = hi.astype(np.float64) + mid.astype(np.float64) * 2.0 ** -11
    err = np.abs(back - n.astype(np.float64))
    assert (err <= 2.0 ** -22 * n + 2.0 ** -35).all()
    flushed = np.where(np.abs(mid.astype(np.float64)) < 2.0 ** -14, 0.0, mid.astype(np.float64))
    err_flush = np.abs(hi.astype(np.float64) + flushed * 2.0 ** -11 - n.astype(np.float64))
    assert (err_flush <= 2.0 ** -22 * n + 2.0 ** -25).all()
    # the sum of the two products is exact in fp32: it fits 24 bits
    s32 = (hi.astype(np.float32) + (mid.astype(np.float32) * np.float32(2.0 ** -11))).astype(np.float32)
    assert (s32.astype(np.float64) == back).all()
    # what rho allows
    assert (np.maximum(err, err_flush) <= 2.0 ** -21 * 16.0 + 2.0 ** -24).all()


# ---- sizes of the scan launch (every size the kernels index with, checked on the CPU) -------------
CELL_SCAN_WAVES, CELL_TILES_PER_PASS = 12, 9          # knn_cells.hip
DEFAULT_DYNAMIC_LDS_LIMIT = 64 * 1024                 # what a launch may ask for without hipFuncSetAttribute


@pytest.mark.parametrize("m", [1, 31, 1000, 1024])     # (longer batches run as passes of <= 1024: 1025 = 1024 + 1, 4096 = 4 x 1024)
@pytest.mark.parametrize("cells_log2", [9, 13, 14, 15, 16])
@pytest.mark.parametrize("blocks_per_cu", [1, 2])
@pytest.mark.parametrize("num_cu", [256, 304, 8])
def test_scan_launch_sizes_fit_the_buffers_the_kernels_index(m, cells_log2, blocks_per_cu, num_cu):
    """knn_cells_scan_plan is the one place the scan grid, the record lists, the shared overflow area and the dynamic LDS of
    a batch are sized; the scan, the prep kernel's counter reset and the re-rank all index with these numbers."""
    import multicore_hw2_amd as pkg
    for nitems in (1 << cells_log2, (1 << cells_log2) * 3 + 7, 5, 1):     # uniform data, fat cells cut into several items, tiny
        p = pkg.debug_scan_plan(num_cu, blocks_per_cu, nitems, m)
        m_padded = (m + 31) // 32 * 32
        assert 1 <= p["blocks"] <= num_cu * blocks_per_cu
        assert p["nlists"] == p["blocks"] * CELL_SCAN_WAVES
        assert p["nlists"] <= p["max_lists"]                              # counts[nlists], zeroed by the prep kernel
        assert p["nlists"] <= max(nitems, CELL_SCAN_WAVES)                # never more waves than items (one block at least)
        assert p["slice"] >= 1 and p["nlists"] * p["slice"] <= p["ovf_base"]   # the waves' slices end where the shared area starts
        assert p["ovf_base"] + p["ovf_cap"] == p["rec_cap"]               # ... which ends with the buffer
        assert p["ovf_cap"] * 16 < 1 << 32                                # the re-rank numbers (record, row) pairs in 32 bits
        # dynamic LDS: B operands (32 B) + thresholds (4 B) per padded query, one norm window of 9 tiles x 8 float4 per wave
        assert p["lds_bytes"] == m_padded * 36 + CELL_SCAN_WAVES * CELL_TILES_PER_PASS * 8 * 16
        assert p["lds_bytes"] + 4 * (3 * 256 + 8) <= DEFAULT_DYNAMIC_LDS_LIMIT   # + the kernel's static tables
        assert p["lds_bytes"] % 16 == 0 and (m_padded * 36) % 16 == 0     # the window is read with ds_read_b128
        # the self-listing scan (round 5): + the batch's Dup values and one list room of 256 entries per wave; two blocks of it
        # must still fit a CU's 160 KiB beside each other
        ps = pkg.debug_scan_plan(num_cu, blocks_per_cu, nitems, m, self_lists=True)
        assert ps["lds_bytes"] == p["lds_bytes"] + m_padded * 4 + CELL_SCAN_WAVES * 256 * 2
        assert {k_: v for k_, v in ps.items() if k_ != "lds_bytes"} == {k_: v for k_, v in p.items() if k_ != "lds_bytes"}
        assert ps["lds_bytes"] + 4 * (3 * 256 + 8) <= DEFAULT_DYNAMIC_LDS_LIMIT and 2 * (ps["lds_bytes"] + 4 * (3 * 256 + 8)) <= 160 * 1024


def test_scan_plan_refuses_batches_longer_than_one_pass():
    import multicore_hw2_amd as pkg
    for bad in ((256, 2, 100, 1025), (256, 3, 100, 64), (0, 1, 100, 64), (256, 1, 100, 0)):
        with pytest.raises(pkg.KnnError):
            pkg.debug_scan_plan(*bad)


@pytest.mark.parametrize("nitems", [1, 15, 16, 17, 8192, 65536, 65536 * 3 + 7])
@pytest.mark.parametrize("blocks", [1, 2, 256, 512, 304])
def test_block_counter_deal_hands_every_item_to_exactly_one_block(nitems, blocks):
    """knn_cells_scan_kernel<DYN = true>: a block's share is one run of CELL_SCAN_RUN items out of every stripe of `blocks`
    runs, the run rotated by a golden-ratio step from stripe to stripe.  The slot -> item arithmetic restated: every item
    belongs to exactly one (block, slot), holes only behind the last item, and a block's runs are spread over the stripe."""
    RUN = 16
    nruns = (nitems + RUN - 1) // RUN
    slots = (nruns + blocks - 1) // blocks * RUN            # i1 in the kernel: the share, in slots
    rot = ((blocks * 2654435769) >> 32) | 1                 # __umulhi(gridDim.x, 2654435769u) | 1u
    s = np.arange(slots, dtype=np.int64)
    seen = np.zeros(nitems, dtype=np.int64)
    stripe = s // RUN
    for b in range(blocks):
        run = stripe * blocks + (b + stripe * rot) % blocks
        it = run * RUN + s % RUN
        ok = (run < nruns) & (it < nitems)
        np.add.at(seen, it[ok], 1)
    assert (seen == 1).all()
    if blocks >= 256 and nitems >= 65536:                   # the rotation walks a block's runs over the stripe: no two stripes at one position
        pos = (0 + np.arange(slots // RUN) * rot) % blocks
        assert len(set(pos.tolist())) == len(pos)


@pytest.mark.parametrize("ranges,qgroups", [(16, 256), (16, 3), (32, 64), (16, 1), (128, 8), (16, 255), (20, 17)])
def test_chunked_scan_block_order_covers_every_pair_once(ranges, qgroups):
    """knn_filter_chunked_kernel's one-dimensional grid (round 4): block L is the (L / 8)-th block of XCD L % 8, which walks its
    query groups (g = XCD mod 8) with all `ranges` tile ranges of a group side by side.  Restated: every (range, query group)
    pair is scored by exactly one block, blocks past the last query group do nothing, and the 64 blocks an XCD has resident
    at a time cover at most ceil(64 / ranges) + 1 query groups (their B fragments are what has to fit its L2)."""
    grid = ranges * ((qgroups + 7) // 8) * 8
    L = np.arange(grid, dtype=np.int64)
    xcd, j = L & 7, L >> 3
    bx, by = j % ranges, (j // ranges) * 8 + xcd
    live = by < qgroups
    pairs = set(zip(bx[live].tolist(), by[live].tolist()))
    assert len(pairs) == int(live.sum()) == ranges * qgroups
    for c in range(8):                                      # per XCD, in dispatch order: any 64 consecutive blocks
        mine = by[(xcd == c) & live]
        for start in range(0, max(1, len(mine) - 64), 64):
            assert len(set(mine[start:start + 64].tolist())) <= (64 + ranges - 1) // ranges + 1


def test_centred_pair_threshold_in_fp32_is_never_below_the_double_formula():
    """Per-cell frames (round 5): the scan makes the threshold of a (query, cell) pair in fp32 (cell_centred_operand,
    knn_cells.hip) where the prep kernel and every other path use knn_threshold's double arithmetic (knn_filter_dev.h).  The fp32
    value only has to be an UPPER bound of the double one — a looser threshold passes more candidates, a tighter one could lose
    the answer.  Both formulas restated here (numpy float32 operation by operation / Python floats), 200 000 random pairs over
    the whole range of their inputs, and the constants of the fp32 form against the ones knn_bound_consts derives."""
    rng = np.random.default_rng(11)
    u, f32 = 2.0 ** -24, np.float32
    theta = 2.0 ** -11 + 2.0 ** -23
    thp, nu0 = theta / (1.0 - theta), 2.0 ** -14 * 1.001
    assert float(f32(4.8865e-4)) >= thp and float(f32(1.2220e-4)) >= 2.0 * nu0
    omega, gam = 2.0 ** -18, 18.0 * u                       # kt = 1: kp = 16
    assert float(f32(1.1921e-5)) >= (omega + 2.0 * gam) * 2.0
    assert float(f32(1.79e-7)) >= 16.0 * 2.0 ** -27 + 2.0 ** -24 and float(f32(4.77e-7)) >= 2.0 ** -21
    assert float(f32(0.999996)) <= (1.0 - gam) * (1.0 - 1e-6) * (1.0 - 2.0 ** -22)
    n = 200000
    k = rng.integers(1, 17, n)
    a = np.where(rng.random(n) < 0.1, 0.0, 10.0 ** rng.uniform(-6, np.log10(16384.0), n))      # the pair's amax (fp16 values)
    b = 10.0 ** rng.uniform(-5, 0.3, n)                                                         # the cell's bmax
    nmax = (k * b * b) * rng.uniform(0.05, 1.0, n)
    mq = (k * a * a) * rng.uniform(1.0 / 16, 1.0, n)
    dup = np.where(rng.random(n) < 0.05, 0.0, 10.0 ** rng.uniform(-12, 9, n))
    ratio = 2.0 ** rng.integers(0, 9, n)
    # --- double: knn_bound_consts + the last lines of knn_threshold, in the cell's units
    dupc = dup * ratio * ratio
    emax = thp * (a + b) + 2.0 * nu0
    eta2 = k * emax * emax
    eta = np.sqrt(eta2)
    rho = (omega + 2.0 * gam) * 2.0 * (nmax + 16.0 * a * a) + 16.0 * 2.0 ** -27 + 2.0 ** -21 * nmax + 2.0 ** -24
    thr = dupc + 2.0 * eta * np.sqrt(dupc) + eta2 + rho - mq * (1.0 - gam)
    thr = thr + np.abs(thr) * 1e-6 + 1e-30
    thr_f = thr.astype(f32)
    thr_f = np.where(thr_f.astype(np.float64) < thr, np.nextafter(thr_f, f32(np.inf)), thr_f)
    thr_f = np.nextafter(thr_f, f32(np.inf))
    # --- fp32: cell_centred_operand, operation by operation (inputs as the kernel gets them: Dup rounded up, sqrt rounded up twice)
    dupq = dup.astype(f32)
    dupq = np.where(dupq.astype(np.float64) < dup, np.nextafter(dupq, f32(np.inf)), dupq)
    sqd = np.nextafter(np.nextafter(np.sqrt(dupq), f32(np.inf)), f32(np.inf))
    A, B, NM, MQ, R_, KF = a.astype(f32), b.astype(f32), nmax.astype(f32), mq.astype(f32), ratio.astype(f32), k.astype(f32)
    B = np.where(B.astype(np.float64) < b, np.nextafter(B, f32(np.inf)), B)     # (bmax / nmax are exact fp32 maxima in the kernel; here rounded up)
    NM = np.where(NM.astype(np.float64) < nmax, np.nextafter(NM, f32(np.inf)), NM)
    A = np.where(A.astype(np.float64) < a, np.nextafter(A, f32(np.inf)), A)
    MQ = np.where(MQ.astype(np.float64) > mq, np.nextafter(MQ, f32(-np.inf)), MQ)
    emax32 = f32(4.8865e-4) * (A + B) + f32(1.2220e-4)
    sqk = np.where(k <= 1, 1.0, np.where(k <= 4, 2.0, np.where(k <= 9, 3.0, 4.0))).astype(f32)
    eta32, eta232 = sqk * emax32, KF * emax32 * emax32
    rho32 = f32(1.1921e-5) * (NM + f32(16.0) * A * A) + f32(1.79e-7) + f32(4.77e-7) * NM
    dupc32, sqdc32 = dupq * R_ * R_, sqd * R_
    P = dupc32 + f32(2.0) * eta32 * sqdc32 + eta232 + rho32
    T = (P * f32(1.00001) + (P + MQ) * f32(2.4e-7) + f32(1e-30)) - MQ * f32(0.999996)
    assert T.dtype == np.float32 and P.dtype == np.float32
    ok = np.isfinite(thr_f) & np.isfinite(T)
    assert ok.mean() > 0.99
    bad = ok & (T < thr_f)
    assert not bad.any(), (int(bad.sum()), [(float(x[bad][0])) for x in (dup, a, b, nmax, mq, ratio, thr, T)])
    # and not vacuous: within 10^-3 of the double value wherever the threshold is not a small difference of large terms (k a
    # square: the kernel takes ceil(sqrt(k)) for sqrt(k), up to 1.41x on the 2 eta sqrt(Dup) term at k = 2)
    big = ok & (thr > 0.1 * (dupc + mq)) & np.isin(k, (1, 4, 9, 16))
    assert float(np.max((T[big] - thr[big]) / thr[big])) < 2e-2     # (the additive constants of rho are rounded up to three digits)
    rest = ok & (thr > 0.1 * (dupc + mq))
    assert float(np.max((T[rest] - thr[rest]) / thr[rest])) < 1.0


@pytest.mark.parametrize("width", [1.0, 1e-2, 1e-4, 3e-7])
@pytest.mark.parametrize("k", [16, 11, 3])
def test_rounding_in_a_cells_own_frame_stays_inside_eta(width, k):
    """Per-cell frames (round 5), the operand-rounding half of the bound (the accumulation half, rho, is frame-independent and
    measured on the GPU: test_parity_gpu.py).  Rows of a cell `width` wide somewhere in a unit box and queries in and around
    it are rounded exactly as knn_cells_recentre_kernel / cell_centred_operand round them — fp32 subtract of the cell's centre,
    power-of-two scale, fp16 to nearest even — and the distance between the ROUNDED points must stay within eta = sqrt(k)
    (theta' (amax + bmax) + 2 nu0) of the true one, in the cell's units: |sqrt(D~) - sqrt(D)| <= eta for every pair, where the
    derivation of knn_filter_dev.h starts.  Down to cells whose coordinates are fp16 subnormals after the 2^8 cap of the scale."""
    rng = np.random.default_rng(int(k * 1000 + width * 1e7) % (2 ** 31))
    f32, f16 = np.float32, np.float16
    n, m = 4000, 300
    sigma = f32(2.0)                                   # the shard's frame: a unit box scaled to [-1, 1]
    where = rng.random(k).astype(f32)
    R = (where + (rng.random((n, k)) - 0.5) * width).astype(f32)
    Q = (where + (rng.random((m, k)) - 0.5) * width * rng.choice([0.5, 2.0, 30.0], (m, 1))).astype(f32)
    lo, hi = R.min(axis=0), R.max(axis=0)
    ctr = (f32(0.5) * lo + f32(0.5) * hi).astype(f32)  # knn_cells_frame_kernel
    hw = float(np.max(np.maximum(hi - ctr, ctr - lo)))
    ratio = 1.0
    for _ in range(8):
        if f32(hw) * sigma * f32(2.0 * ratio) <= f32(0.999):
            ratio *= 2.0
    scale = f32(sigma * f32(ratio))
    a = ((R - ctr).astype(f32) * scale).astype(f32).astype(f16)          # the fragments
    b = ((Q - ctr).astype(f32) * scale).astype(f32).astype(f16)          # the queries in the cell's frame
    fits = np.isfinite(b.astype(np.float64)).all(axis=1) & (np.abs(b.astype(np.float64)).max(axis=1) <= 16384.0)
    assert fits.mean() > 0.5
    a64, b64 = a.astype(np.float64), b.astype(np.float64)[fits]
    Qf = Q.astype(np.float64)[fits]
    d_true = np.sqrt(((Qf[:, None, :] - R.astype(np.float64)[None, :, :]) ** 2).sum(-1)) * float(scale)
    d_round = np.sqrt(((b64[:, None, :] - a64[None, :, :]) ** 2).sum(-1))
    theta = 2.0 ** -11 + 2.0 ** -23
    thp, nu0 = theta / (1.0 - theta), 2.0 ** -14 * 1.001
    amax = np.abs(b64).max(axis=1)
    bmax = np.abs(a64).max()
    eta = np.sqrt(k) * (thp * (amax + bmax) + 2.0 * nu0)
    worst = float((np.abs(d_round - d_true) / eta[:, None]).max())
    assert worst <= 1.0, (width, k, worst)
    # and the frame earns its keep: in the shard's one frame the same bound would be 2^-11 of the BOX
    eta_one_frame = np.sqrt(k) * (thp * 2.0 + 2.0 * nu0) * ratio       # in the cell's units
    if width <= 1e-2:
        assert float(np.median(eta)) < 0.2 * eta_one_frame
