// knn_grid.hip — spatial index for low dimensions (k <= 4): a uniform grid over the shard's bounding
// box, cells in counting-sort order, queried ring by ring with a stop rule that keeps the answer
// bit-identical to the brute-force scan (gfx950 / MI355X).
//
// The reference's counterpart is its KD-tree pair (sources/src/core.cu:960-1050 host build + recursive
// CPU query `v9`, core.cu:1051-1191 one-thread-per-query GPU traversal `v10`; README.md:339-346: it wins
// only at k = 3 and collapses at k = 16).  A tree is the wrong shape for this chip — a recursive descent
// per thread diverges in every wave and its build is a host-side nth_element pass — so the index here is
// flat: no pointers, no recursion, no host work beyond sizing the grid.
//   build  : cell of every row (one pass) -> histogram -> exclusive scan -> scatter rows into cell order
//   query  : ONE WAVE per query.  Ring r = the cells at Chebyshev distance r from the query's cell; the 64
//            lanes split the ring's cells, evaluate their rows with the exact v0 arithmetic (core.cu:44-49)
//            into packed (distance, index) keys, min-reduce across the wave.  After ring r every row not
//            yet seen lies outside the (2r+1)^k block of cells, i.e. at least LB(r) away along some axis;
//            the search stops as soon as  best < LB(r)^2 (1 - 1e-6)  — strictly below what v0 could compute
//            for any unseen row, so neither the minimum nor a tie for it can be outside (lowest index among
//            equal distances is decided by the packed key among the rows seen).
// Never used when the rows hold non-finite values or the grid degenerates (a cell with > 4096 rows):
// the brute-force kernels take those shards.
#include "knn_common.h"

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#define GRID_BLOCK 256

typedef float f4g __attribute__((ext_vector_type(4)));

struct GridGeom {
    int k;
    int g[4];            // cells per dimension
    unsigned stride[4];  // linear cell id = sum c_d * stride[d]
    float lo[4];         // lower corner of the box
    float inv_w[4];      // cells per unit length (0 when the dimension is degenerate)
    double dlo[4], w[4]; // the same corner / cell width in double, for the stop rule
    double slack[4];     // what a face gives away to the fp32 rounding of the rows' cell assignment (see the stop rule)
};

__device__ __forceinline__ unsigned grid_cell_of(const GridGeom &gg, const float *x, int c_out[4])
{
    unsigned cell = 0u;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        int c = 0;
        if (d < gg.k) {
            const float t = (x[d] - gg.lo[d]) * gg.inv_w[d];
            c = t >= 0.0f ? (t < (float)gg.g[d] ? (int)t : gg.g[d] - 1) : 0;   // NaN -> 0 (queries only)
        }
        c_out[d] = c;
        cell += (unsigned)c * gg.stride[d];
    }
    return cell;
}

// ---- build ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(GRID_BLOCK) void grid_minmax_kernel(const float *__restrict__ R, long long n, int k,
                                                                 unsigned *__restrict__ stats)
{
    // stats[d] = ordered min, stats[4 + d] = ordered max, stats[8] = #non-finite values
    __shared__ unsigned s_lo[4], s_hi[4], s_bad;
    if (threadIdx.x < 4) {
        s_lo[threadIdx.x] = 0xFFFFFFFFu;
        s_hi[threadIdx.x] = 0u;
    }
    if (threadIdx.x == 0)
        s_bad = 0u;
    __syncthreads();
    float lo[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, hi[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    unsigned bad = 0u;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        for (int d = 0; d < k; ++d) {
            const float v = R[(size_t)i * k + d];
            if (!(fabsf(v) < INFINITY))
                ++bad;
            lo[d] = fminf(lo[d], v);
            hi[d] = fmaxf(hi[d], v);
        }
    auto ord = [](float f) {
        const unsigned u = __float_as_uint(f);
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    };
    for (int d = 0; d < k; ++d)
        if (lo[d] <= hi[d]) {
            atomicMin(&s_lo[d], ord(lo[d]));
            atomicMax(&s_hi[d], ord(hi[d]));
        }
    if (bad)
        atomicAdd(&s_bad, bad);
    __syncthreads();
    if (threadIdx.x < (unsigned)k) {
        atomicMin(&stats[threadIdx.x], s_lo[threadIdx.x]);
        atomicMax(&stats[4 + threadIdx.x], s_hi[threadIdx.x]);
    }
    if (threadIdx.x == 0 && s_bad)
        atomicAdd(&stats[8], s_bad);
}

__global__ __launch_bounds__(GRID_BLOCK) void grid_count_kernel(const float *__restrict__ R, long long n, GridGeom gg,
                                                                unsigned *__restrict__ cell_of_row,
                                                                unsigned *__restrict__ counts)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < gg.k; ++d)
        x[d] = R[(size_t)i * gg.k + d];
    int c[4];
    const unsigned cell = grid_cell_of(gg, x, c);
    cell_of_row[i] = cell;
    atomicAdd(&counts[cell], 1u);
}

// Exclusive scan of counts[0..cells) in three steps: per-block scan (2048 entries per block) + block
// totals, scan of the totals by one block, add.  Also the largest single count (degenerate-grid guard).
#define SCAN_PER_BLOCK 2048
__global__ __launch_bounds__(GRID_BLOCK) void grid_scan_blocks_kernel(const unsigned *__restrict__ counts, unsigned cells,
                                                                      unsigned *__restrict__ start,
                                                                      unsigned *__restrict__ totals,
                                                                      unsigned *__restrict__ maxcount)
{
    __shared__ unsigned s_sum[GRID_BLOCK];
    const unsigned base = blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * 8u;
    unsigned v[8], sum = 0u, mx = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = base + j < cells ? counts[base + j] : 0u;
        mx = v[j] > mx ? v[j] : mx;
        sum += v[j];
    }
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (unsigned off = 1; off < GRID_BLOCK; off <<= 1) {   // Hillis-Steele over the 256 thread sums
        const unsigned add = threadIdx.x >= off ? s_sum[threadIdx.x - off] : 0u;
        __syncthreads();
        s_sum[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned run = s_sum[threadIdx.x] - sum;   // exclusive prefix of this thread inside the block
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (base + j < cells)
            start[base + j] = run;
        run += v[j];
    }
    if (threadIdx.x == GRID_BLOCK - 1)
        totals[blockIdx.x] = s_sum[threadIdx.x];
    if (mx)
        atomicMax(maxcount, mx);
}

__global__ __launch_bounds__(GRID_BLOCK) void grid_scan_totals_kernel(unsigned *__restrict__ totals, unsigned nblocks)
{
    // one block: serial over chunks of 256 totals (nblocks <= 8192 for 2^24 cells)
    __shared__ unsigned s_v[GRID_BLOCK];
    __shared__ unsigned s_carry;
    if (threadIdx.x == 0)
        s_carry = 0u;
    __syncthreads();
    for (unsigned c0 = 0; c0 < nblocks; c0 += GRID_BLOCK) {
        const unsigned i = c0 + threadIdx.x;
        const unsigned v = i < nblocks ? totals[i] : 0u;
        s_v[threadIdx.x] = v;
        __syncthreads();
        for (unsigned off = 1; off < GRID_BLOCK; off <<= 1) {
            const unsigned add = threadIdx.x >= off ? s_v[threadIdx.x - off] : 0u;
            __syncthreads();
            s_v[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nblocks)
            totals[i] = s_carry + s_v[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == GRID_BLOCK - 1)
            s_carry += s_v[threadIdx.x];
        __syncthreads();
    }
}

__global__ __launch_bounds__(GRID_BLOCK) void grid_scan_add_kernel(unsigned *__restrict__ start, unsigned cells,
                                                                   const unsigned *__restrict__ totals, unsigned n_rows)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cells)
        start[i] += totals[i / SCAN_PER_BLOCK];
    if (i == 0)
        start[cells] = n_rows;
}

__global__ __launch_bounds__(GRID_BLOCK) void grid_scatter_kernel(const float *__restrict__ R, long long n, int k,
                                                                  const unsigned *__restrict__ cell_of_row,
                                                                  const unsigned *__restrict__ start,
                                                                  unsigned *__restrict__ fill,
                                                                  f4g *__restrict__ pts, unsigned *__restrict__ orig)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const unsigned cell = cell_of_row[i];
    const unsigned pos = start[cell] + atomicAdd(&fill[cell], 1u);   // any order inside a cell is fine
    f4g p = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < k; ++d)
        p[d] = R[(size_t)i * k + d];
    pts[pos] = p;
    orig[pos] = (unsigned)i;
}

// ---- query ----------------------------------------------------------------------------------------
__device__ __forceinline__ u64 grid_wave_min(u64 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const u64 o = __shfl_xor(v, off, KNN_WAVE);
        v = o < v ? o : v;
    }
    return v;
}

template <int K>
__global__ __launch_bounds__(GRID_BLOCK) void knn_grid_query_kernel(const float *__restrict__ Q, int m, GridGeom gg,
                                                                    const unsigned *__restrict__ start,
                                                                    const f4g *__restrict__ pts,
                                                                    const unsigned *__restrict__ orig, long long base,
                                                                    u64 *__restrict__ keys, int rmax,
                                                                    unsigned *__restrict__ giveup,
                                                                    unsigned *__restrict__ giveup_next)
{
#pragma clang fp contract(off)
    // (the flag word of the NEXT batch on this workspace slot is cleared here: the two words of a slot alternate,
    // calls on a slot are stream-ordered, so no memset launch is needed between batches)
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *giveup_next = 0u;
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * (GRID_BLOCK / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (qi >= m)
        return;
    float q[4] = {0.f, 0.f, 0.f, 0.f};
    bool finite = true;
#pragma unroll
    for (int d = 0; d < K; ++d) {
        q[d] = Q[(size_t)qi * K + d];
        finite = finite && fabsf(q[d]) < INFINITY;
    }
    if (!finite)
        return;   // every distance is NaN or +INF: v0 keeps (+INF, index 0) = the key's initial value
    int c[4];
    (void)grid_cell_of(gg, q, c);
    int gmax = 1;
#pragma unroll
    for (int d = 0; d < K; ++d)
        gmax = gg.g[d] > gmax ? gg.g[d] : gmax;

    u64 best = kKeyInit;
    bool done = false;
    // (a query far outside the box, or in an empty region, would walk O(r^K) cells per ring: past rmax rings
    // it gives up and raises the flag that un-gates the brute-force scan queued behind this kernel)
    // The first pass takes rings 0 and 1 together (the whole 3^K block): with ~3 rows per cell the own cell alone
    // almost never satisfies the stop rule, and every ring is a chain of dependent loads (cell bounds -> rows).
    bool first = true;
    for (int r = gmax > 1 ? 1 : 0; r < gmax && r <= rmax; ++r) {
        // the cells of ring r: positions of the (2r+1)^K block whose largest |offset| is exactly r
        const int side = 2 * r + 1;
        int total = 1;
#pragma unroll
        for (int d = 0; d < K; ++d)
            total *= side;
        u64 mine = kKeyInit;
        for (int idx = lane; idx < total; idx += KNN_WAVE) {
            int rem = idx, far = 0;
            bool inside = true;
            unsigned cell = 0u;
#pragma unroll
            for (int d = 0; d < K; ++d) {
                const int off = rem % side - r;
                rem /= side;
                const int cd = c[d] + off;
                far = abs(off) > far ? abs(off) : far;
                inside = inside && cd >= 0 && cd < gg.g[d];
                cell += (unsigned)(inside ? cd : 0) * gg.stride[d];
            }
            if (!inside || (first ? far > r : far != r))
                continue;
            const unsigned p0 = start[cell], p1 = start[cell + 1];
            for (unsigned p = p0; p < p1; ++p) {
                const f4g x = pts[p];
                float acc = 0.0f;
#pragma unroll
                for (int d = 0; d < K; ++d) {
                    const float diff = q[d] - x[d];   // v0: search - reference, squared, summed in order
                    const float sq = diff * diff;
                    acc = acc + sq;
                }
                if (acc < INFINITY) {                 // NaN / +INF never beat +INF (v0's strict >)
                    const u64 key = ((u64)__float_as_uint(acc) << 32) | (u64)(unsigned)(base + orig[p]);
                    mine = key < mine ? key : mine;
                }
            }
        }
        first = false;
        mine = grid_wave_min(mine);
        best = mine < best ? mine : best;
        // stop rule: rows not yet seen are outside the block of rings 0..r; along the axis where they leave
        // it they are at least `lb` from the query (faces beyond the grid bound nothing: no rows out there).
        // A row's cell comes from fp32 arithmetic, t = fl(fl(x - lo) * inv_w): a row of a cell below face F
        // (an integer) has t < F, hence x < lo + F w (1 + 2^-22) — two roundings and the rounding of inv_w itself —
        // and a row of a cell at or above F has x >= lo + F w (1 - 2^-21).  The error grows with the face's distance
        // from the corner, at most g w 2^-21: gg.slack[d] = (1e-3 + g 2^-21) w covers it (knn_grid_build keeps
        // g <= 2^19 so that this stays a quarter of a cell; the 1e-3 w alone, the round-2 form, was short of it
        // from g ~ 4000 up on a single live axis).
        double lb = INFINITY;
        bool covers_all = true;
#pragma unroll
        for (int d = 0; d < K; ++d) {
            if (c[d] - r > 0) {
                covers_all = false;
                const double face = gg.dlo[d] + (double)(c[d] - r) * gg.w[d];
                lb = fmin(lb, (double)q[d] - face - gg.slack[d]);
            }
            if (c[d] + r < gg.g[d] - 1) {
                covers_all = false;
                const double face = gg.dlo[d] + (double)(c[d] + r + 1) * gg.w[d];
                lb = fmin(lb, face - (double)q[d] - gg.slack[d]);
            }
        }
        if (covers_all) {
            done = true;
            break;
        }
        if (lb > 0.0) {
            const float bd = __uint_as_float((unsigned)(best >> 32));
            if ((double)bd < lb * lb * (1.0 - 1e-6)) {
                done = true;
                break;
            }
        }
    }
    if (lane == 0) {
        if (best < kKeyInit)   // a real row's key: folding it in is right whether or not the search finished
            __hip_atomic_fetch_min(&keys[qi], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!done && gmax > rmax + 1)
            *giveup = 1u;      // benign race: every writer stores 1
    }
}

// ---- host -------------------------------------------------------------------------------------------
#define GTRY(call)                       \
    do {                                 \
        hipError_t e_ = (call);          \
        if (e_ != hipSuccess)            \
            return e_;                   \
    } while (0)

struct GridState {
    bool usable = false;
    GridGeom geom;
    unsigned cells = 0;
    unsigned *start = nullptr;   // device [cells + 1]
    f4g *pts = nullptr;          // device [n]: rows in cell order, padded to 4 floats
    unsigned *orig = nullptr;    // device [n]: shard-local row number of pts[i]
    unsigned max_cell = 0;
    unsigned *giveup = nullptr;  // device [KNN_SLOTS][2]: != 0 after a query batch = some query left the grid search
    mutable unsigned calls[KNN_SLOTS] = {};   // batches issued per slot (picks the slot's flag word)
};

void knn_grid_free(GridState *&gs)
{
    if (!gs)
        return;
    (void)knn_dev_free(gs->start);
    (void)knn_dev_free(gs->pts);
    (void)knn_dev_free(gs->orig);
    (void)knn_dev_free(gs->giveup);
    delete gs;
    gs = nullptr;
}

static inline float grid_ord2f(unsigned o)
{
    const unsigned u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// Builds the grid for refs[0..n) (device, AoS, k <= 4).  Synchronous.  *out stays null (hipSuccess) when
// the data rules the index out: non-finite values, a degenerate box, a cell with more than 4096 rows.
hipError_t knn_grid_build(GridState **out, int k, long long n, const float *r, hipStream_t s)
{
    *out = nullptr;
    if (k < 1 || k > 4 || n < 64 || n > 0x7FFFFFFFll)
        return hipSuccess;
    unsigned hstats[9] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
    unsigned *dstats = nullptr;
    GTRY(knn_dev_alloc((void **)&dstats, sizeof hstats));
    hipError_t e = hipMemcpyAsync(dstats, hstats, sizeof hstats, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(grid_minmax_kernel, dim3(1024), dim3(GRID_BLOCK), 0, s, r, n, k, dstats);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(hstats, dstats, sizeof hstats, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);
    (void)knn_dev_free(dstats);
    GTRY(e);
    if (hstats[8] != 0u)
        return hipSuccess;   // NaN / Inf among the rows
    // ~3 rows per cell on average: g cells per axis over the k axes that have any extent
    GridGeom gg;
    memset(&gg, 0, sizeof gg);
    gg.k = k;
    int live = 0;
    double lo[4], hi[4];
    for (int d = 0; d < k; ++d) {
        lo[d] = grid_ord2f(hstats[d]);
        hi[d] = grid_ord2f(hstats[4 + d]);
        if (!(hi[d] - lo[d] < 1e30))
            return hipSuccess;   // the box itself overflows fp32 arithmetic
        if (hi[d] > lo[d])
            ++live;
    }
    if (live == 0)
        return hipSuccess;       // all rows identical
    int g = (int)floor(pow((double)n / 3.0, 1.0 / live));
    // <= ~4M cells; one live axis: <= 2^19 so that the stop rule's rounding slack (g 2^-21 cells) stays a quarter of a cell
    const int gcap = live == 1 ? (1 << 19) : live == 2 ? 2048 : live == 3 ? 160 : 45;
    g = g < 1 ? 1 : g > gcap ? gcap : g;
    unsigned cells = 1u;
    for (int d = 0; d < 4; ++d) {
        gg.g[d] = 1;
        gg.stride[d] = 0u;
    }
    for (int d = 0; d < k; ++d) {
        gg.g[d] = hi[d] > lo[d] ? g : 1;
        gg.stride[d] = cells;
        cells *= (unsigned)gg.g[d];
        gg.lo[d] = (float)lo[d];
        gg.dlo[d] = lo[d];
        gg.w[d] = hi[d] > lo[d] ? (hi[d] - lo[d]) / gg.g[d] : 1.0;
        gg.slack[d] = (1e-3 + (double)gg.g[d] * 0x1p-21) * gg.w[d];
        gg.inv_w[d] = hi[d] > lo[d] ? (float)((double)gg.g[d] / (hi[d] - lo[d])) : 0.0f;
        if (!(gg.inv_w[d] < INFINITY))
            return hipSuccess;
    }
    GridState *gs = new (std::nothrow) GridState();
    if (!gs)
        return hipErrorOutOfMemory;
    gs->geom = gg;
    gs->cells = cells;
    unsigned *cell_of_row = nullptr, *counts = nullptr, *totals = nullptr, *dmax = nullptr;
    const unsigned nblocks = (cells + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    e = knn_dev_alloc((void **)&gs->start, ((size_t)cells + 1) * sizeof(unsigned));
    if (e == hipSuccess)
        e = knn_dev_alloc((void **)&gs->pts, (size_t)n * sizeof(f4g));
    if (e == hipSuccess)
        e = knn_dev_alloc((void **)&gs->orig, (size_t)n * sizeof(unsigned));
    if (e == hipSuccess)
        e = knn_dev_alloc((void **)&gs->giveup, 2 * KNN_SLOTS * sizeof(unsigned));
    if (e == hipSuccess)
        e = hipMemsetAsync(gs->giveup, 0, 2 * KNN_SLOTS * sizeof(unsigned), s);
    if (e == hipSuccess)
        e = knn_dev_alloc((void **)&cell_of_row, (size_t)n * sizeof(unsigned));
    if (e == hipSuccess)
        e = knn_dev_alloc((void **)&counts, ((size_t)cells + 1) * sizeof(unsigned));
    if (e == hipSuccess)
        e = knn_dev_alloc((void **)&totals, ((size_t)nblocks + 1) * sizeof(unsigned));
    if (e == hipSuccess)
        e = knn_dev_alloc((void **)&dmax, sizeof(unsigned));
    if (e == hipSuccess)
        e = hipMemsetAsync(counts, 0, ((size_t)cells + 1) * sizeof(unsigned), s);
    if (e == hipSuccess)
        e = hipMemsetAsync(dmax, 0, sizeof(unsigned), s);
    unsigned hmax = 0u;
    if (e == hipSuccess) {
        const unsigned rb = (unsigned)((n + GRID_BLOCK - 1) / GRID_BLOCK);
        hipLaunchKernelGGL(grid_count_kernel, dim3(rb), dim3(GRID_BLOCK), 0, s, r, n, gg, cell_of_row, counts);
        hipLaunchKernelGGL(grid_scan_blocks_kernel, dim3(nblocks), dim3(GRID_BLOCK), 0, s, counts, cells, gs->start, totals, dmax);
        hipLaunchKernelGGL(grid_scan_totals_kernel, dim3(1), dim3(GRID_BLOCK), 0, s, totals, nblocks);
        hipLaunchKernelGGL(grid_scan_add_kernel, dim3((cells + GRID_BLOCK - 1) / GRID_BLOCK), dim3(GRID_BLOCK), 0, s, gs->start,
                           cells, totals, (unsigned)n);
        e = hipMemsetAsync(counts, 0, ((size_t)cells + 1) * sizeof(unsigned), s);   // reused as the fill counters
        if (e == hipSuccess) {
            hipLaunchKernelGGL(grid_scatter_kernel, dim3(rb), dim3(GRID_BLOCK), 0, s, r, n, k, cell_of_row, gs->start, counts,
                               gs->pts, gs->orig);
            e = hipGetLastError();
        }
        if (e == hipSuccess)
            e = hipMemcpyAsync(&hmax, dmax, sizeof hmax, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess)
            e = hipStreamSynchronize(s);
    }
    (void)knn_dev_free(cell_of_row);
    (void)knn_dev_free(counts);
    (void)knn_dev_free(totals);
    (void)knn_dev_free(dmax);
    if (e != hipSuccess || hmax > 4096u) {
        knn_grid_free(gs);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            e = hipSuccess;
        }
        return e;
    }
    gs->max_cell = hmax;
    gs->usable = true;
    *out = gs;
    return hipSuccess;
}

// Asynchronous on `s`.  *gate_out = the device word the brute-force scan queued behind this must be gated on.
hipError_t knn_grid_query(const GridState *gs, int slot, int m, const float *q, long long base, u64 *keys,
                          const unsigned **gate_out, hipStream_t s)
{
    *gate_out = nullptr;
    if (!gs || !gs->usable || m <= 0)
        return hipSuccess;
    const unsigned call = gs->calls[slot]++;
    unsigned *giveup = gs->giveup + 2 * slot + (call & 1u);
    unsigned *giveup_next = gs->giveup + 2 * slot + ((call + 1u) & 1u);
    const int k = gs->geom.k;
    const int rmax = k == 1 ? 64 : k == 2 ? 16 : k == 3 ? 6 : 4;
    const dim3 grid((unsigned)((m + GRID_BLOCK / 64 - 1) / (GRID_BLOCK / 64))), block(GRID_BLOCK);
    switch (k) {
    case 1: hipLaunchKernelGGL(knn_grid_query_kernel<1>, grid, block, 0, s, q, m, gs->geom, gs->start, gs->pts, gs->orig, base, keys, rmax, giveup, giveup_next); break;
    case 2: hipLaunchKernelGGL(knn_grid_query_kernel<2>, grid, block, 0, s, q, m, gs->geom, gs->start, gs->pts, gs->orig, base, keys, rmax, giveup, giveup_next); break;
    case 3: hipLaunchKernelGGL(knn_grid_query_kernel<3>, grid, block, 0, s, q, m, gs->geom, gs->start, gs->pts, gs->orig, base, keys, rmax, giveup, giveup_next); break;
    default: hipLaunchKernelGGL(knn_grid_query_kernel<4>, grid, block, 0, s, q, m, gs->geom, gs->start, gs->pts, gs->orig, base, keys, rmax, giveup, giveup_next); break;
    }
    *gate_out = giveup;
    return hipGetLastError();
}

void knn_grid_info(const GridState *gs, long long info[4])
{
    info[0] = gs ? gs->cells : 0;
    info[1] = gs ? gs->max_cell : 0;
    info[2] = gs ? gs->geom.g[0] : 0;
    info[3] = gs && gs->usable ? 1 : 0;
}
