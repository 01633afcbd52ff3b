#!/usr/bin/env python3
"""How should the reference set be split over N GPUs so that the pruned scan's total work is conserved?
CPU simulation (numpy, no GPU): uniform data in [0,1)^16, m = 1024 queries, n = 2^24 (the metric's shape).

The pruned path (knn_cells.hip) rules cell c out for query q when LB(c, q) = sum_d gap_d^2 > Dup_q, where Dup_q bounds the
distance of the answer from above (it comes from the best score among the query's SEED cells).  A batch's scan costs
    tile steps = sum over cells of  tiles(c) * ceil(listed queries(c) / 32)          (one MFMA + min tree each).
Partitions compared, per rank of an N-GPU run:
  index-range   (reference core.cu:875-883, rounds 1-3): rank r holds rows [r n/N, (r+1) n/N), grids them by itself
                (2^bits(n/N) cells) and bounds every query from its own rows.
  cell-range    ONE global grid (2^bits(n) cells); rank r owns the cells whose top log2(N) code bits equal r.
     local bound    Dup_q from the rank's own rows only (seeds: the query's cell mirrored into the rank's range)
     global bound   Dup_q as one GPU would compute it (seeds: the query's own cell + the 3 cells across its two nearest
                    cuts, all their rows) — what a second all-reduce(min) of Dup_q would give every rank
     seed layer T   every rank keeps the first T tiles of EVERY cell of the global grid (replicated: 65536 T tiles of
                    1152 B); seeds that are not local come from it, local ones use the whole cell
     seed layer T, wide   the same layer, but the non-local seeds are the 32 / T cells nearest to the query (by LB)
                    instead of 4: the same number of seed tiles per query as today
Prints tile steps per rank (mean and busiest rank) and cells surviving per query; the 1-GPU row calibrates against the
measured INSTS_MFMA = 6.71e5 (profiles/r03_c3_sq_counters.txt) and 2.98e5 for a 2^21-row index-range shard.
"""
import sys
import time

import numpy as np

K = 16
ROWS_MIN = 144          # knn_cells_build: cells of 144 .. 288 rows
DUP_SLACK = 1.012       # what the fp16 error band adds to the seed's squared distance (~1 %)


def grid_bits(n):
    bits = 0
    while (ROWS_MIN << (bits + 1)) <= n:
        bits += 1
    return min(bits, 16, 4 * K)


def geometry(bits):
    nb = [bits // K + (1 if d < bits % K else 0) for d in range(K)]
    shift, pos = [], 0
    for d in range(K):
        shift.append(pos)
        pos += nb[d]
    return nb, shift


def codes_of(X, cuts, nb, shift):
    c = np.zeros(X.shape[0], dtype=np.uint32)
    for d in range(K):
        if nb[d]:
            b = np.searchsorted(cuts[d], X[:, d], side="right").astype(np.uint32)
            c |= b << np.uint32(shift[d])
    return c


def gaps_of(q, cuts, nb):
    """gap2[d][bin] = squared distance from q_d to the bin's interval (0 inside)."""
    out = []
    for d in range(K):
        if not nb[d]:
            out.append(np.zeros(1))
            continue
        nbins = 1 << nb[d]
        g = np.zeros(nbins)
        for b in range(nbins):
            if b > 0 and cuts[d][b - 1] > q[d]:
                g[b] = cuts[d][b - 1] - q[d]
            if b < nbins - 1 and q[d] > cuts[d][b]:
                g[b] = q[d] - cuts[d][b]
        out.append(g * g)
    return out


def lb_all_cells(gap2, nb, shift, bits):
    """LB(c, q) for every cell code, as the kernels do it: separable sum over the dimensions."""
    lb = np.zeros(1)
    for d in range(K):          # dimension 0 holds the lowest code bits
        if nb[d]:
            lb = (gap2[d][:, None] + lb[None, :]).reshape(-1)
    assert lb.shape[0] == 1 << bits
    return lb


class Grid:
    def __init__(self, R, bits, sample_stride=4096):
        self.bits = bits
        self.nb, self.shift = geometry(bits)
        samp = R[:: max(1, R.shape[0] // 1024)][:1024]
        self.cuts = []
        for d in range(K):
            if self.nb[d]:
                col = np.sort(samp[:, d])
                nbins = 1 << self.nb[d]
                self.cuts.append(np.array([col[j * len(col) // nbins] for j in range(1, nbins)]))
            else:
                self.cuts.append(np.zeros(0))
        code = codes_of(R, self.cuts, self.nb, self.shift)
        self.order = np.argsort(code, kind="stable")
        self.counts = np.bincount(code, minlength=1 << bits)
        self.start = np.concatenate([[0], np.cumsum(self.counts)])
        self.tiles = (self.counts + 31) // 32
        self.R = R

    def rows_of(self, c, limit=None):
        a, e = self.start[c], self.start[c + 1]
        if limit is not None:
            e = min(e, a + limit)
        return self.R[self.order[a:e]]

    def seed_cells(self, q, ndims=2, frozen=()):
        """own cell + every combination of moves across the `ndims` nearest cuts (dimensions in `frozen` never move)."""
        own, moves = 0, []
        for d in range(K):
            if not self.nb[d]:
                continue
            nbins = 1 << self.nb[d]
            b = int(np.searchsorted(self.cuts[d], q[d], side="right"))
            own |= b << self.shift[d]
            if d in frozen:
                continue
            best = None
            if b > 0:
                best = (q[d] - self.cuts[d][b - 1], b - 1)
            if b + 1 < nbins and (best is None or self.cuts[d][b] - q[d] < best[0]):
                best = (self.cuts[d][b] - q[d], b + 1)
            if best is not None:
                moves.append((best[0], d, best[1]))
        moves.sort()
        cells = []
        for mask in range(1 << min(ndims, len(moves))):
            c = own
            for j in range(min(ndims, len(moves))):
                if (mask >> j) & 1:
                    _, d, alt = moves[j]
                    c = (c & ~(((1 << self.nb[d]) - 1) << self.shift[d])) | (alt << self.shift[d])
            cells.append(c)
        return own, cells


def nearest_d2(q, rows):
    if rows.shape[0] == 0:
        return np.inf
    d = rows.astype(np.float64) - q.astype(np.float64)
    return float((d * d).sum(1).min())


def work(listed, tiles):
    """tile steps of a batch: listed[c] = queries that could not rule cell c out."""
    return int((tiles * ((listed + 31) // 32)).sum())


def main():
    n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 24)
    m = 1024
    rng = np.random.default_rng(7)
    t0 = time.time()
    R = rng.random((n, K), dtype=np.float32)
    Q = rng.random((m, K), dtype=np.float32)
    print(f"uniform [0,1)^16, n = 2^{n.bit_length() - 1}, m = {m}; Dup = seed d2 x {DUP_SLACK}", flush=True)

    gbits = grid_bits(n)
    G = Grid(R, gbits)
    print(f"global grid: 2^{gbits} cells, {G.counts.mean():.0f} rows per cell, {G.tiles.sum()} tiles  ({time.time() - t0:.0f} s)", flush=True)

    # ---- per query on the global grid: LB of every cell, the one-GPU seeds, the ideal bound
    lbs = np.empty((m, 1 << gbits), dtype=np.float32)
    dup_global = np.empty(m)
    owns = np.empty(m, dtype=np.int64)
    seeds4 = []
    for i in range(m):
        gap2 = gaps_of(Q[i], G.cuts, G.nb)
        lbs[i] = lb_all_cells(gap2, G.nb, G.shift, gbits)
        own, cells = G.seed_cells(Q[i])
        owns[i] = own
        seeds4.append(cells)
        dup_global[i] = min(nearest_d2(Q[i], G.rows_of(c)) for c in cells) * DUP_SLACK
    surv = lbs <= dup_global[:, None].astype(np.float32)
    listed = surv.sum(0)
    w1 = work(listed, G.tiles)
    print(f"\n1 GPU: {surv.sum(1).mean():.0f} cells per query, {listed.mean():.1f} queries per cell, tile steps {w1:.3e}"
          f"   (measured INSTS_MFMA 6.71e5 at C3)", flush=True)

    print("\nN   partition / bound                         cells/query/rank   tile steps per rank: mean   busiest   x ideal (1-GPU / N)")
    for N in (2, 4, 8):
        rb = N.bit_length() - 1
        ideal = w1 / N
        # ---- index-range: rank 0's shard, its own grid, its own seeds (all ranks are statistically alike)
        Rl = R[: n // N]
        lbits = grid_bits(n // N)
        L = Grid(Rl, lbits)
        cnt = np.zeros(1 << lbits, dtype=np.int64)
        per_q = 0
        for i in range(m):
            gap2 = gaps_of(Q[i], L.cuts, L.nb)
            lb = lb_all_cells(gap2, L.nb, L.shift, lbits)
            _, cells = L.seed_cells(Q[i])
            dup = min(nearest_d2(Q[i], L.rows_of(c)) for c in cells) * DUP_SLACK
            s = lb <= dup
            cnt += s
            per_q += int(s.sum())
        w = work(cnt, L.tiles)
        print(f"{N}   index-range (2^{lbits} cells per rank)          {per_q / m:10.0f}        {w:12.3e}  {w:10.3e}   {w / ideal:5.2f}", flush=True)

        # ---- cell-range partitions of the global grid
        rank_of_cell = np.arange(1 << gbits) >> (gbits - rb)
        top_dims = [d for d in range(K) if G.nb[d] and G.shift[d] >= gbits - rb]

        def report(label, dup_per_rank):
            """dup_per_rank[r][i] = the bound rank r has for query i"""
            ws, cq = [], 0
            for r in range(N):
                mine = rank_of_cell == r
                s = lbs[:, mine] <= dup_per_rank[r][:, None].astype(np.float32)
                ws.append(work(s.sum(0), G.tiles[mine]))
                cq += int(s.sum())
            print(f"{N}   {label:<40s}  {cq / m / N:10.0f}        {np.mean(ws):12.3e}  {max(ws):10.3e}   {max(ws) / ideal:5.2f}", flush=True)

        report("cell-range, global bound", [dup_global] * N)

        # local bound: the query's seed cells mirrored into the rank's range (top bits forced), moves on the other dimensions
        dl = []
        for r in range(N):
            d_r = np.empty(m)
            for i in range(m):
                _, cells = G.seed_cells(Q[i], frozen=top_dims)
                lowmask = (1 << (gbits - rb)) - 1
                d_r[i] = min(nearest_d2(Q[i], G.rows_of((c & lowmask) | (r << (gbits - rb)))) for c in cells) * DUP_SLACK
            dl.append(d_r)
        report("cell-range, rank-local bound", dl)

        for T in (1, 2, 4):
            # seed layer: non-local seed cells contribute their first T tiles only; local ones all their rows
            dl = []
            for r in range(N):
                d_r = np.empty(m)
                for i in range(m):
                    best = np.inf
                    for c in seeds4[i]:
                        local = rank_of_cell[c] == r
                        best = min(best, nearest_d2(Q[i], G.rows_of(c, None if local else 32 * T)))
                    d_r[i] = best * DUP_SLACK
                dl.append(d_r)
            report(f"cell-range, seed layer T={T} ({(1 << gbits) * T * 1152 / 1e6:.0f} MB), 4 cells", dl)
        for T in (1, 2):
            # the same layer, the 32 / T nearest cells as seeds (own cell + moves over the log2(32 / T) nearest cuts)
            nd = (32 // T).bit_length() - 1
            d_all = np.empty(m)
            for i in range(m):
                _, cells = G.seed_cells(Q[i], ndims=nd)
                d_all[i] = min(nearest_d2(Q[i], G.rows_of(c, 32 * T)) for c in cells) * DUP_SLACK
            report(f"cell-range, seed layer T={T}, {32 // T} nearest cells", [d_all] * N)


# ------------------------------------------------------------------------------------------------------------
# Analytic mode (`shard_sim.py model <log2 n> [<global bits>]`): the same rules on a MODEL of uniform data — cuts at
# j / 2^nb exactly, rows per cell ~ Poisson(n / cells), a cell's rows drawn on demand inside its box — so that sets that do
# not fit this container (C4: n = 2^27) and global grids finer than one GPU's 2^16 cells can be projected.  Validated against
# the real-data mode above at n = 2^24 (same table within a few per cent).
# ------------------------------------------------------------------------------------------------------------
def grid_bits_uncapped(n):
    bits = 0
    while (ROWS_MIN << (bits + 1)) <= n:
        bits += 1
    return bits


class ModelGrid:
    def __init__(self, n, bits, seed):
        self.n, self.bits = n, bits
        self.nb, self.shift = geometry(bits)
        self.cuts = [np.arange(1, 1 << b) / float(1 << b) if b else np.zeros(0) for b in self.nb]
        rng = np.random.default_rng(seed)
        self.counts = rng.poisson(n / float(1 << bits), 1 << bits)
        self.tiles = (self.counts + 31) // 32
        self.seed = seed
        self.cache = {}

    def rows_of(self, c, limit=None):
        if c not in self.cache:
            rng = np.random.default_rng((self.seed, int(c)))
            lo, w = np.zeros(K), np.ones(K)
            for d in range(K):
                if self.nb[d]:
                    b = (int(c) >> self.shift[d]) & ((1 << self.nb[d]) - 1)
                    w[d] = 1.0 / (1 << self.nb[d])
                    lo[d] = b * w[d]
            self.cache[c] = (lo + w * rng.random((int(self.counts[c]), K))).astype(np.float32)
        r = self.cache[c]
        return r if limit is None else r[:limit]

    seed_cells = Grid.seed_cells


def model(n, gbits_forced=None):
    m = 1024
    Q = np.random.default_rng(7).random((m, K), dtype=np.float32)
    bits1 = grid_bits(n)     # what one GPU holding everything picks (capped at 2^16 cells)
    print(f"MODEL of uniform [0,1)^16, n = 2^{n.bit_length() - 1}, m = {m}; Dup = seed d2 x {DUP_SLACK}", flush=True)

    def seeds_dup(G, q, limit_tiles=None, ndims=2, local=None):
        """bound from the seed cells; limit_tiles: tiles a NON-local seed cell contributes (None: all)."""
        _, cells = G.seed_cells(q, ndims=ndims)
        best = np.inf
        for c in cells:
            full = limit_tiles is None or (local is not None and local(c))
            rows = G.rows_of(c) if full else G.rows_of(c, 32 * limit_tiles)
            if full and G.tiles[c] > 36:   # the library samples a fat seed cell: every stride-th tile, <= 36
                rows = rows[: 36 * 32]
            best = min(best, nearest_d2(q, rows))
        return best * DUP_SLACK

    def run(G, rank_of_cell, nranks, dups, label, ideal):
        """dups[r][i]; prints the row of the table"""
        masks = [rank_of_cell == r for r in range(nranks)]
        listed = [np.zeros(int(masks[r].sum()), dtype=np.int64) for r in range(nranks)]
        cq = 0
        for i in range(m):
            lb = lb_all_cells(gaps_of(Q[i], G.cuts, G.nb), G.nb, G.shift, G.bits).astype(np.float32)
            for r in range(nranks):
                s = lb[masks[r]] <= np.float32(dups[r][i])
                listed[r] += s
                cq += int(s.sum())
        ws = [work(listed[r], G.tiles[masks[r]]) for r in range(nranks)]
        print(f"   {label:<72s} {cq / m / nranks:8.0f}   {np.mean(ws):10.3e}  {max(ws):10.3e}" +
              (f"   {max(ws) / ideal:5.2f}" if ideal else ""), flush=True)
        return max(ws)

    G1 = ModelGrid(n, bits1, 11)
    print(f"\n   partition / bound                                                 cells per query and rank  tile steps: mean  busiest   x (1-GPU / N)")
    d1 = np.array([seeds_dup(G1, Q[i]) for i in range(m)])
    w1 = run(G1, np.zeros(1 << bits1, dtype=np.int64), 1, [d1], f"1 GPU, 2^{bits1} cells of {n >> bits1} rows", None)
    for N in (2, 4, 8):
        print(f"N = {N}")
        # index-range: n / N rows over the whole cube, the rank's own grid and seeds
        lb_ = grid_bits(n // N)
        L = ModelGrid(n // N, lb_, 100 + N)
        dl = np.array([seeds_dup(L, Q[i]) for i in range(m)])
        run(L, np.zeros(1 << lb_, dtype=np.int64), 1, [dl], f"index-range, 2^{lb_} cells of {(n // N) >> lb_} rows per rank", w1 / N)
        # cell-range on a global grid: by default the finest whose per-rank share is <= 2^16 cells of >= ROWS_MIN rows
        rb = N.bit_length() - 1
        gb = gbits_forced if gbits_forced else min(16 + rb, grid_bits_uncapped(n))
        Gg = G1 if gb == bits1 else ModelGrid(n, gb, 11)
        roc = np.arange(1 << gb) >> (gb - rb)
        dg = d1 if Gg is G1 else np.array([seeds_dup(Gg, Q[i]) for i in range(m)])
        run(Gg, roc, N, [dg] * N, f"cell-range of 2^{gb} cells, global bound (4 whole cells)", w1 / N)
        for T, nd in ((1, 2), (2, 2), (2, 4), (1, 5), (4, 2)):
            dr = [np.array([seeds_dup(Gg, Q[i], limit_tiles=T, ndims=nd, local=lambda c, r=r: roc[c] == r) for i in range(m)]) for r in range(N)]
            run(Gg, roc, N, dr, f"cell-range, seed layer T={T} ({(1 << gb) * T * 1152 / 1e6:.0f} MB), {1 << nd} seed cells, local ones whole", w1 / N)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "model":
        model(1 << int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else None)
    else:
        main()
