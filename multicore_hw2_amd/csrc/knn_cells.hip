// knn_cells.hip — the cell-pruned form of the MFMA filter (k <= 16, resident shards; gfx950 / MI355X).
// knn_filter.hip holds the filter it prunes: the fp16 layouts, the error bound, the full scan, the query
// entry point that picks between the two.  Here: the cell index (build), and per batch the kernels that
// score a query only against the cells it cannot rule out.
#include "knn_filter_dev.h"
#include "knn_exact_dev.h"

#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <type_traits>
#include <vector>

// ------------------------------------------------------------------------------------------
// Cell-pruned scan (k <= 16, resident index).  Scoring every (query, reference) pair costs ~50 issue
// cycles per 32x32 tile pair whatever the schedule, so the only way under it is not to
// score most pairs.  The index sorts the rows into 2^B cells — every dimension cut into 2^nb[d] bins at
// sample quantiles, cell = the tuple of bin numbers — and lays the fp16 fragments out cell by cell (each
// cell padded to whole 32-row tiles, `perm` maps a layout position back to its row).  Per batch (the default chain):
//   prep    : one block of 2 or 4 waves per query (knn_cells_prep_kernel).  Rounds the query to its fp16 B operand, scores
//             the query's own cell and the 3 cells across its two nearest cuts with the MFMA (a cell of many tiles is
//             sampled) — the minimum is a score of a real reference, which is all knn_threshold needs — derives thr_q for
//             the scores and Dup_q = the largest real (scaled) squared distance any candidate for the answer can have,
//             and tabulates the separable halves of the cell lower bound: a row of cell c differs from the query by at
//             least gap_d(bin_d(c)) in every dimension, so LB(c, q) = sum_d gap_d^2 <= |q - r|^2 for every row of the
//             cell = lo_q[low bits of c] + hi_q[high bits], both rounded DOWN.  LB > Dup_q rules the whole cell out, ties
//             included (a row that ties with the answer obeys the Dup bound too).  The kernel also starts the batch's
//             keys at (+INF, 0) when asked to and clears the control words of the NEXT batch.
//   match   : cell-major (knn_cells_match_kernel): which queries cannot rule a cell out — queries on the lanes against the
//             high table, survivors against 64-byte runs of the low table — written as per-cell lists (a list longer than
//             its room marks the cell `dense`: scored against every query of the batch, which is what a list that long asks
//             for anyway).
//   scan    : knn_cells_scan_kernel.  The unit of work is an ITEM: a run of at most 18 tiles of one cell (fat cells of
//             clustered data are several items, empty cells none).  A wave loads the item's tiles once, gathers the listed
//             queries' B operands from LDS 32 at a time and runs MFMA + min3 tree + threshold test per tile; hits become
//             records (query, tile, half) in the wave's slice, overflow in a small area all waves share.
//   re-rank : knn_rerank_kernel (knn_exact.hip): v0's arithmetic on the records' 16 rows each, through `perm`.
//   gated, on the device: more candidates than the record buffers hold (rows of a cluster tighter than the fp16 step:
//             every row of a query's cells passes) -> knn_cells_exact_kernel evaluates the batch's listed (item, query) pairs
//             with the exact arithmetic — the geometry's verdict stands, it never depended on fp16; a query nothing bounds
//             (not finite, far away, no real row seen) -> the exact scan of the whole shard.  Nothing switches the cells off
//             for later batches.
// (The round-2 chain, the scan with its norm tile out of an extra MFMA and the fused match + scan + re-rank kernel of round 3
// all lost their A/B against this chain and live on as records under tools/arms/ and profiles/r03_sweep_experiments.txt.)
// Uniform data in 16 dimensions, n = 2^24: ~1600 of 65536 cells survive per query, 25 queries per cell.
// ------------------------------------------------------------------------------------------
#define CELL_MAX_BINS 16
#define CELL_TILES_PER_PASS 9                 // reference tiles a wave holds in registers at a time (one more costs the sixth wave per SIMD)

struct CellGeom {
    int k, bits, sa;                 // dimensions, total bits, bits of the low table
    unsigned char nb[16], shift[16]; // bits of dimension d (0 = not cut), position of its bin number in the cell code
    // cell-range shards: `bits` / `nb` / `shift` describe the GLOBAL grid; this index holds codes [cell_base, cell_base + ncells)
    // of it under local numbers (cell_base is a multiple of 2^sa), nh = its entries of the high table.  Else 0, 2^bits, 2^(bits - sa).
    unsigned cell_base, ncells, nh;
};

static CellGeom cell_geom_of(const CellIndex &c, int k)
{
    CellGeom g;
    memset(&g, 0, sizeof g);
    g.k = k;
    g.bits = c.bits;
    g.sa = c.sa;
    memcpy(g.nb, c.nb, 16);
    memcpy(g.shift, c.shift, 16);
    g.cell_base = c.cell_base;
    g.ncells = c.ncells;
    g.nh = (c.ncells + (1u << c.sa) - 1u) >> c.sa;
    return g;
}

// Shape of a grid of 2^bits cells over k dimensions: bits per dimension (the first bits % k dimensions get one more), the
// position of each dimension's bin number in the cell code, and sa = the number of low code bits that are whole dimensions
// and at most 8 (the low pruning table has 2^sa entries).
static void cell_grid_shape(int k, int bits, unsigned char nb[16], unsigned char shift[16], int *sa_out)
{
    int pos = 0, sa = 0;
    for (int d = 0; d < 16; ++d) {
        nb[d] = 0;
        shift[d] = 0;
    }
    for (int d = 0; d < k; ++d) {
        nb[d] = (unsigned char)(bits / k + (d < bits % k ? 1 : 0));
        shift[d] = (unsigned char)pos;
        if (pos <= 8)
            sa = pos;
        pos += nb[d];
    }
    if (pos <= 8)
        sa = pos;
    *sa_out = sa;
}

// Cuts at the sample quantiles: bounds[d][j - 1] = the j-th of 2^nb[d] quantiles of dimension d (+INF beyond).
static void cell_quantile_cuts(int k, const unsigned char nb[16], const float *samp, long long samples, float *bounds)
{
    std::vector<float> col((size_t)samples);
    for (int i = 0; i < 16 * (CELL_MAX_BINS - 1); ++i)
        bounds[i] = INFINITY;
    for (int d = 0; d < std::min(k, 16); ++d) {   // (16 < k <= 32: the cells cut the first 16 dimensions only)
        if (!nb[d])
            continue;
        for (long long i = 0; i < samples; ++i)
            col[(size_t)i] = samp[(size_t)i * k + d];
        const int nbins = 1 << nb[d];
        if (nbins == 2)   // one cut: the median, without sorting the column
            std::nth_element(col.begin(), col.begin() + samples / 2, col.end());
        else
            std::sort(col.begin(), col.end());
        for (int j = 1; j < nbins; ++j)
            bounds[(size_t)d * (CELL_MAX_BINS - 1) + (j - 1)] = col[(size_t)(j * samples / nbins)];
    }
}

static int cell_bits_for_rows(long long n)
{
    int bits = 0;
    long long rows_min = 144;            // cells of 144 .. 288 rows on average: 5-9 tiles each
    if (const char *e = getenv("KNN_MI355X_CELL_ROWS"))   // experiment: other cell sizes
        rows_min = std::max(32, atoi(e));
    while ((rows_min << (bits + 1)) <= n)
        ++bits;
    return bits;
}

__device__ __forceinline__ unsigned cell_bin(const float *__restrict__ bnd, int nbins, float x)
{
    unsigned b = 0u;
    for (int j = 0; j < nbins - 1; ++j)   // ascending cuts: bin b <=> bnd[b-1] <= x < bnd[b]; NaN -> bin 0
        b += x >= bnd[j] ? 1u : 0u;
    return b;
}

// ---- the norm as two fp16 halves ------------------------------------------------------------------
// word = fp16(N) | fp16((N - fp16(N)) * 2^11) << 16.  N - fp16(N) is exact in fp32 (both are fp32 values of about
// the same exponent), the scaled remainder keeps 11 more bits: hi + mid 2^-11 = N to within 2^-22 N, or 2^-25
// absolute where the scaled remainder falls into fp16's subnormal range (< 2^-14) and the matrix core flushes it.
// knn_bound_consts carries both in rho.  +INF (padding, rows outside the box) -> (+INF, 0).
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_norm22(float n)
{
    if (!(n < INFINITY))
        return 0x00007C00u;
    const _Float16 hi = (_Float16)n;
    const float rem = n - (float)hi;
    const _Float16 mid = (_Float16)(rem * 2048.0f);
    return (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, mid) << 16);
}

// A operand whose rows carry (hi, mid) in K-slots 0 and 1 (nw: the row's word on lanes 0..31, ZERO on lanes 32..63,
// which hold K-slots 8..15), and the B operand that sums them: every column = {1, 2^-11, 0, ...}.
__device__ __forceinline__ h8 norm_a_operand(unsigned nw)
{
    const u4v t = {nw, 0u, 0u, 0u};
    return __builtin_bit_cast(h8, t);
}
__device__ __forceinline__ h8 norm_b_operand()
{
    const u4v t = {0x10003C00u, 0u, 0u, 0u};   // fp16 1.0 = 0x3C00, fp16 2^-11 = 0x1000
    return __builtin_bit_cast(h8, t);
}
__device__ __forceinline__ f16v zero_acc()
{
    return (f16v){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
}

// Before a block counts itself done on a counter another block's finaliser watches: every vector-memory operation of this
// wave — the no-return global atomics on keys[] and ctl[] among them — has been performed.  `s_waitcnt vmcnt(0)` only
// (0x0F70: expcnt and lgkmcnt left alone): gfx9 counts stores and no-return atomics in vmcnt and releases the count when the
// memory side has acknowledged them; an agent-scope release fence would add `buffer_wbl2 sc1`, a write-back of the XCD's
// whole L2 per wave (110 -> 261 us at C3, round 4), which atomics performed at the memory side do not need.
__device__ __forceinline__ void cells_wait_own_atomics()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0x0F70);
}

// LDS written by some lanes of a wave, read by others of the SAME wave: order the two without a block barrier.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the LOCAL number of a cell code; a row outside the index's range of the grid (cell-range shards: the caller's partition
// was not this geometry's) is counted and parked in cell 0 — the build then fails
__device__ __forceinline__ unsigned cell_local(unsigned code, const CellGeom &g, unsigned *__restrict__ bad)
{
    const unsigned l = code - g.cell_base;
    if (code < g.cell_base || l >= g.ncells) {
        atomicAdd(bad, 1u);
        return 0u;
    }
    return l;
}

__global__ __launch_bounds__(256) void knn_cells_code_kernel(const float *__restrict__ R, long long n, CellGeom g,
                                                             const float *__restrict__ bounds,
                                                             unsigned *__restrict__ code, unsigned *__restrict__ counts,
                                                             unsigned *__restrict__ bad)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float *__restrict__ x = R + (size_t)i * g.k;
    unsigned c = 0u;
    for (int d = 0; d < min(g.k, 16); ++d)
        if (g.nb[d])
            c |= cell_bin(bounds + d * (CELL_MAX_BINS - 1), 1 << g.nb[d], x[d]) << g.shift[d];
    c = cell_local(c, g, bad);
    code[i] = c;
    atomicAdd(&counts[c], 1u);
}

// The one-pass placement for 16 < k <= 32 (two K-steps per tile: fragments [tile][2][64 lanes][8], round 5): what
// knn_cells_scatter_frag_kernel does for k <= 16, with the row's 32 (padded) dimensions as four 8-wide halves.
__global__ __launch_bounds__(256) void knn_cells_scatter_frag2_kernel(
    const float *__restrict__ R, long long n, int k, const unsigned *__restrict__ code,
    const unsigned *__restrict__ tile_start, unsigned *__restrict__ fill, const float *__restrict__ center, float sigma,
    h8 *__restrict__ frag, float *__restrict__ norms, unsigned *__restrict__ norms2, unsigned *__restrict__ perm,
    unsigned *__restrict__ out, unsigned *__restrict__ olist, unsigned ocap)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float vmax = 0.0f, nrm = 0.0f;
    if (i < n) {
        const float *__restrict__ x = R + (size_t)i * k;
        bool real = true;
        h8 v[4];
#pragma unroll
        for (int part = 0; part < 4; ++part)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = part * 8 + j;
                const float sc = d < k ? (x[d] - center[d]) * sigma : 0.0f;  // fp32 subtract, exact power-of-two scale
                const _Float16 hval = (_Float16)sc;                          // round to nearest even
                const float back = (float)hval;
                real = real && fabsf(back) <= 1.0f;                          // outside the robust box, NaN included
                vmax = fmaxf(vmax, fabsf(back));
                nrm = nrm + back * back;                                     // exact products, fp32 sum in dimension order
                v[part][j] = hval;
            }
        if (!real) {   // out of the filter (zero fragment, +INF norm), into the exact list
            const unsigned opos = atomicAdd(&out[3], 1u);
            if (opos < ocap)
                olist[opos] = (unsigned)i;
#pragma unroll
            for (int part = 0; part < 4; ++part)
                v[part] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
            vmax = 0.0f;
            nrm = 0.0f;
        }
        if (k <= KNN_NIF_MAX_K) {   // the norm's two fp16 halves in the free K-slots 30, 31 (see cell_tile_step_nif); a row that is out: a score no threshold passes
            const unsigned nw = real ? pack_norm22(nrm) : 0x00007BFFu;   // (65504, 0)
            v[3][6] = __builtin_bit_cast(_Float16, (unsigned short)(nw & 0xFFFFu));
            v[3][7] = __builtin_bit_cast(_Float16, (unsigned short)(nw >> 16));
        }
        const unsigned c = code[i];
        const size_t pos = (size_t)tile_start[c] * 32 + atomicAdd(&fill[c], 1u);
        h8 *__restrict__ tile = frag + (pos >> 5) * 128;   // two K-steps of 64 lanes
#pragma unroll
        for (int part = 0; part < 4; ++part)
            tile[(part >> 1) * 64 + (part & 1) * 32 + (pos & 31)] = v[part];
        norms[pos] = real ? nrm : INFINITY;
        norms2[pos] = pack_norm22(real ? nrm : INFINITY);
        perm[pos] = (unsigned)i;
    }
    vmax = wave_max_f(vmax);
    nrm = wave_max_f(nrm);
    __shared__ float s_v[4], s_n[4];
    if ((threadIdx.x & 63) == 0) {
        s_v[threadIdx.x >> 6] = vmax;
        s_n[threadIdx.x >> 6] = nrm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        guarded_atomic_max(&out[0], __float_as_uint(fmaxf(fmaxf(s_v[0], s_v[1]), fmaxf(s_v[2], s_v[3]))));
        guarded_atomic_max(&out[1], __float_as_uint(fmaxf(fmaxf(s_n[0], s_n[1]), fmaxf(s_n[2], s_n[3]))));
    }
}

// Scatter + layout in one pass (k <= 16): row i, read in row order, goes to the next free position of its
// cell as an fp16 fragment + norm — what knn_frag_kernel would write there (same arithmetic, same outlier
// rule), without a second pass that gathers 64-byte rows in cell order (8.8 ms for 2^24 rows, against 1.1).
// (the padding positions of every cell are written afterwards by knn_cells_pad_kernel); out[] as in
// knn_frag_kernel.
__global__ __launch_bounds__(256) void knn_cells_scatter_frag_kernel(
    const float *__restrict__ R, long long n, int k, const unsigned *__restrict__ code,
    const unsigned *__restrict__ tile_start, unsigned *__restrict__ fill, const float *__restrict__ center, float sigma,
    h8 *__restrict__ frag, float *__restrict__ norms, unsigned *__restrict__ norms2, unsigned *__restrict__ perm,
    unsigned *__restrict__ out, unsigned *__restrict__ olist, unsigned ocap)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float vmax = 0.0f, nrm = 0.0f;
    if (i < n) {
        const float *__restrict__ x = R + (size_t)i * k;
        bool real = true;
        h8 v[2];
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = half * 8 + j;
                const float sc = d < k ? (x[d] - center[d]) * sigma : 0.0f;  // fp32 subtract, exact power-of-two scale
                const _Float16 hval = (_Float16)sc;                          // round to nearest even
                const float back = (float)hval;
                real = real && fabsf(back) <= 1.0f;                          // outside the robust box, NaN included
                vmax = fmaxf(vmax, fabsf(back));
                nrm = nrm + back * back;                                     // exact products, fp32 sum
                v[half][j] = hval;
            }
        if (!real) {   // out of the filter (zero fragment, +INF norm), into the exact list
            const unsigned opos = atomicAdd(&out[3], 1u);
            if (opos < ocap)
                olist[opos] = (unsigned)i;
            v[0] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
            v[1] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
            vmax = 0.0f;
            nrm = 0.0f;
        }
        const unsigned c = code[i];
        const size_t pos = (size_t)tile_start[c] * 32 + atomicAdd(&fill[c], 1u);
        frag[(pos >> 5) * 64 + (pos & 31)] = v[0];
        frag[(pos >> 5) * 64 + 32 + (pos & 31)] = v[1];
        norms[pos] = real ? nrm : INFINITY;
        norms2[pos] = pack_norm22(real ? nrm : INFINITY);
        perm[pos] = (unsigned)i;
    }
    vmax = wave_max_f(vmax);
    nrm = wave_max_f(nrm);
    __shared__ float s_v[4], s_n[4];
    if ((threadIdx.x & 63) == 0) {
        s_v[threadIdx.x >> 6] = vmax;
        s_n[threadIdx.x >> 6] = nrm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        guarded_atomic_max(&out[0], __float_as_uint(fmaxf(fmaxf(s_v[0], s_v[1]), fmaxf(s_v[2], s_v[3]))));
        guarded_atomic_max(&out[1], __float_as_uint(fmaxf(fmaxf(s_n[0], s_n[1]), fmaxf(s_n[2], s_n[3]))));
    }
    // (out[2], knn_frag_kernel's count of values beyond fp16's range, stays 0 here: such rows fail the |x| <= 1
    // test above and are on the exact list)
}

// ---- the two-pass build (round 3) ---------------------------------------------------------------------------
// The one-pass placement above reads the rows once and writes every row's 44 bytes as five pieces to a random place of a
// 0.6 GB layout: 3.2 GB of HBM writes for 0.73 GB of output at n = 2^24 (partial lines), and 16 M atomics on 2^16
// counters in front of it.  Two-pass: the rows are first grouped by the TOP 8 BITS of their cell code — 256 buckets of
// consecutive cells, 256 write streams of whole 64-byte lines — and every bucket is then placed by blocks that share
// one XCD's L2: its 2-3 MB of layout are written piecewise but leave the cache as whole lines.  Cell counts come from
// per-block LDS histograms of a bucket's records instead of one global atomic per row.
#define CELL_BUCKETS 256
#define CELL_PLACE_PARTS 16   // blocks per bucket in the placement pass

#define CELL_BUILD_ROWS 4096   // rows (records) a block of the build's passes handles: 16 per thread

// the cell code of a row held in registers
__device__ __forceinline__ unsigned cell_code_of(const float (&x)[16], const CellGeom &g, const float *__restrict__ bounds)
{
    unsigned c = 0u;
#pragma unroll
    for (int d = 0; d < 16; ++d)
        if (d < g.k && g.nb[d])
            c |= cell_bin(bounds + d * (CELL_MAX_BINS - 1), 1 << g.nb[d], x[d]) << g.shift[d];
    return c;
}

// A: code of every row + how many rows each bucket gets.  (A lane reads its 64-byte row as four 16-byte loads when k = 16:
// sixteen 4-byte loads at a 64-byte stride kept the texture path busy for 0.9 ms of a pass that moves 1 GB.)
__global__ __launch_bounds__(256) void knn_cells_bucket_count_kernel(const float *__restrict__ R, long long n, CellGeom g,
                                                                     const float *__restrict__ bounds, int bshift,
                                                                     unsigned *__restrict__ code,
                                                                     unsigned *__restrict__ bucket_counts,
                                                                     unsigned *__restrict__ bad)
{
    __shared__ unsigned s_h[CELL_BUCKETS];
    __shared__ float s_bnd[16 * (CELL_MAX_BINS - 1)];
    s_h[threadIdx.x] = 0u;
    if (threadIdx.x < 16 * (CELL_MAX_BINS - 1))
        s_bnd[threadIdx.x] = bounds[threadIdx.x];
    __syncthreads();
    const bool vec = g.k == 16 && ((uintptr_t)R & 15u) == 0;
    const long long i0 = (long long)blockIdx.x * CELL_BUILD_ROWS;
    for (int it = 0; it < CELL_BUILD_ROWS / 256; ++it) {
        const long long i = i0 + it * 256 + threadIdx.x;
        if (i < n) {
            float x[16];
            if (vec) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f4v v = __builtin_nontemporal_load(&((const f4v *)(R + (size_t)i * 16))[j]);
                    x[4 * j] = v[0];
                    x[4 * j + 1] = v[1];
                    x[4 * j + 2] = v[2];
                    x[4 * j + 3] = v[3];
                }
            } else {
#pragma unroll
                for (int d = 0; d < 16; ++d)
                    x[d] = d < g.k ? R[(size_t)i * g.k + d] : 0.0f;
            }
            const unsigned c = cell_local(cell_code_of(x, g, s_bnd), g, bad);
            code[i] = c;
            atomicAdd(&s_h[c >> bshift], 1u);
        }
    }
    __syncthreads();
    if (s_h[threadIdx.x] != 0u)
        atomicAdd(&bucket_counts[threadIdx.x], s_h[threadIdx.x]);
}

// Block-wide helper of the build's passes: s_h[256] holds a histogram of the block's 4096 items over 256 bins; on return
// s_start[bin] = first slot of the bin in the block's sorted order (exclusive prefix), s_h unchanged.  256 threads.
__device__ __forceinline__ void block_prefix_256(const unsigned *__restrict__ s_h, unsigned *__restrict__ s_start,
                                                 unsigned *__restrict__ s_wsum)
{
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const unsigned v = s_h[threadIdx.x];
    unsigned inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(inc, off, KNN_WAVE);
        if (lane >= (unsigned)off)
            inc += o;
    }
    if (lane == 63u)
        s_wsum[w] = inc;
    __syncthreads();
    unsigned base = 0u;
    for (unsigned j = 0; j < w; ++j)
        base += s_wsum[j];
    s_start[threadIdx.x] = base + inc - v;
    __syncthreads();
}

// B: every row to the next free record of its bucket: 64 bytes of coordinates (k of them used) + (code << 32 | row).  A block
// counts its 4096 rows per bucket first and reserves its room in every bucket with ONE atomic (one per row's block of 256
// was 16 M returning atomics on 256 words: 1.6 ms), then writes in BUCKET order — a counting sort of the rows' numbers in
// LDS — so that the ~16 records it adds to a bucket leave as one 1 KiB run (in row order they were 64-byte pieces spread
// over time: 0.94 ms against 0.6).
__global__ __launch_bounds__(256) void knn_cells_bucket_scatter_kernel(const float *__restrict__ R, long long n, int k, int bshift,
                                                                       const unsigned *__restrict__ code,
                                                                       const unsigned *__restrict__ bucket_start,
                                                                       unsigned *__restrict__ bucket_fill,
                                                                       float *__restrict__ trows, u64 *__restrict__ tmeta)
{
    __shared__ unsigned s_h[CELL_BUCKETS], s_start[CELL_BUCKETS], s_cur[CELL_BUCKETS], s_base[CELL_BUCKETS], s_wsum[4];
    __shared__ unsigned short s_order[CELL_BUILD_ROWS];
    s_h[threadIdx.x] = 0u;
    __syncthreads();
    const long long i0 = (long long)blockIdx.x * CELL_BUILD_ROWS;
    unsigned cs[CELL_BUILD_ROWS / 256];
#pragma unroll
    for (int it = 0; it < CELL_BUILD_ROWS / 256; ++it) {
        const long long i = i0 + it * 256 + threadIdx.x;
        cs[it] = i < n ? code[i] : 0u;
        if (i < n)
            atomicAdd(&s_h[cs[it] >> bshift], 1u);
    }
    __syncthreads();
    block_prefix_256(s_h, s_start, s_wsum);
    {
        const unsigned cnt = s_h[threadIdx.x];   // thread t reserves the block's room in bucket t
        s_base[threadIdx.x] = cnt != 0u ? bucket_start[threadIdx.x] + atomicAdd(&bucket_fill[threadIdx.x], cnt) : 0u;
        s_cur[threadIdx.x] = s_start[threadIdx.x];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < CELL_BUILD_ROWS / 256; ++it)
        if (i0 + it * 256 + threadIdx.x < n)
            s_order[atomicAdd(&s_cur[cs[it] >> bshift], 1u)] = (unsigned short)(it * 256 + threadIdx.x);
    __syncthreads();
    const unsigned nrows = (unsigned)min((long long)CELL_BUILD_ROWS, n - i0);
    const bool vec = k == 16 && ((uintptr_t)R & 15u) == 0;
    for (unsigned slot = threadIdx.x; slot < nrows; slot += 256u) {
        const unsigned src = s_order[slot];
        const long long i = i0 + src;
        const unsigned c = code[i];
        const unsigned bk = c >> bshift;
        const size_t pos = (size_t)s_base[bk] + (slot - s_start[bk]);
        const float *__restrict__ x = R + (size_t)i * k;
        float *__restrict__ t = trows + pos * 16;
        if (vec) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                ((f4v *)t)[j] = ((const f4v *)x)[j];
        } else {
            for (int d = 0; d < k; ++d)
                t[d] = x[d];
        }
        tmeta[pos] = ((u64)c << 32) | (u64)(unsigned)i;
    }
}

// ---- the fast build (round 5): no pass over the rows for counts, no host round trip between the passes ---------------
// B': A and B in one kernel, for buckets of FIXED room (`cap` records each: the caller sizes it for rows that spread evenly
// over the buckets + 1/8, which is what the quantile cuts give on data they fit; a bucket that outgrows its room raises
// `overflow` and the caller builds again with the counted passes above).  The block reads its 4096 rows ONCE from memory —
// codes worked out from the rows on the fly and parked in LDS, no `code` array — and copies them in bucket order four lanes
// to a row: every load and store of the copy is a whole 64-byte line (one lane per row was four instructions touching 64
// different lines each).  The second read of a row comes out of the L2 (the block's rows are 256 KiB).
// R / n: the rows [row0, row0 + n) of the shard (a chunk of an ingest, or all of it): tmeta carries row0 + the row's number in R.
__global__ __launch_bounds__(256) void knn_cells_bucket_scatter_fixed_kernel(const float *__restrict__ R, long long n, long long row0, CellGeom g,
                                                                             const float *__restrict__ bounds, int bshift, unsigned cap,
                                                                             unsigned *__restrict__ bucket_fill,
                                                                             float *__restrict__ trows, u64 *__restrict__ tmeta,
                                                                             unsigned *__restrict__ overflow)
{
    __shared__ unsigned s_h[CELL_BUCKETS], s_start[CELL_BUCKETS], s_cur[CELL_BUCKETS], s_base[CELL_BUCKETS], s_wsum[4];
    __shared__ unsigned short s_order[CELL_BUILD_ROWS], s_code[CELL_BUILD_ROWS];
    __shared__ float s_bnd[16 * (CELL_MAX_BINS - 1)];
    s_h[threadIdx.x] = 0u;
    if (threadIdx.x < 16 * (CELL_MAX_BINS - 1))
        s_bnd[threadIdx.x] = bounds[threadIdx.x];
    __syncthreads();
    const long long i0 = (long long)blockIdx.x * CELL_BUILD_ROWS;
    const bool vec = g.k == 16 && ((uintptr_t)R & 15u) == 0;
    if (vec) {
        // four lanes to a row, 16 bytes each: the block reads its 256 KiB as one linear stream (a lane per row was four
        // instructions, each touching 64 different lines).  A lane bins its four dimensions, the quad ORs the pieces.
        const unsigned q = threadIdx.x & 3u;
        unsigned nbq[4], shq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            nbq[e] = g.nb[4 * q + e];      // (a dependent load from the kernel arguments per lane: once, outside the loop)
            shq[e] = g.shift[4 * q + e];
        }
#pragma unroll 4
        for (int it = 0; it < CELL_BUILD_ROWS / 64; ++it) {
            const unsigned row = (unsigned)it * 64u + (threadIdx.x >> 2);
            const long long i = i0 + row;
            unsigned c = 0u;
            if (i < n) {
                const f4v v = ((const f4v *)(R + (size_t)i * 16))[q];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (nbq[e])
                        c |= cell_bin(s_bnd + (4 * q + e) * (CELL_MAX_BINS - 1), 1 << nbq[e], v[e]) << shq[e];
            }
            c |= (unsigned)__shfl_xor((int)c, 1, KNN_WAVE);
            c |= (unsigned)__shfl_xor((int)c, 2, KNN_WAVE);
            if (q == 0u && i < n) {
                s_code[row] = (unsigned short)c;
                atomicAdd(&s_h[c >> bshift], 1u);
            }
        }
    } else {
#pragma unroll 4
        for (int it = 0; it < CELL_BUILD_ROWS / 256; ++it) {
            const long long i = i0 + it * 256 + threadIdx.x;
            if (i < n) {
                float x[16];
#pragma unroll
                for (int d = 0; d < 16; ++d)
                    x[d] = d < g.k ? R[(size_t)i * g.k + d] : 0.0f;
                const unsigned c = cell_code_of(x, g, s_bnd);   // (no shard geometry on this path: local = global code, < 2^16)
                s_code[it * 256 + threadIdx.x] = (unsigned short)c;
                atomicAdd(&s_h[c >> bshift], 1u);
            }
        }
    }
    __syncthreads();
    block_prefix_256(s_h, s_start, s_wsum);
    {
        const unsigned cnt = s_h[threadIdx.x];   // thread t reserves the block's room in bucket t
        unsigned at = 0u;
        if (cnt != 0u) {
            at = atomicAdd(&bucket_fill[threadIdx.x], cnt);
            if (at + cnt > cap)
                atomicOr(overflow, 1u);
        }
        s_base[threadIdx.x] = at;   // position INSIDE the bucket; records beyond `cap` are not written
        s_cur[threadIdx.x] = s_start[threadIdx.x];
    }
    __syncthreads();
    const unsigned nrows = (unsigned)min((long long)CELL_BUILD_ROWS, n - i0);
    for (unsigned j = threadIdx.x; j < nrows; j += 256u)
        s_order[atomicAdd(&s_cur[(unsigned)s_code[j] >> bshift], 1u)] = (unsigned short)j;
    __syncthreads();
    if (vec) {
        const unsigned q = threadIdx.x & 3u;
        for (unsigned slot = threadIdx.x >> 2; slot < nrows; slot += 64u) {
            const unsigned src = s_order[slot];
            const unsigned c = s_code[src];
            const unsigned bk = c >> bshift;
            const unsigned inb = s_base[bk] + (slot - s_start[bk]);
            if (inb < cap) {
                const size_t pos = (size_t)bk * cap + inb;
                // (streaming stores: the records are read next by another kernel; the block's own rows should stay cached for
                // this second read of them)
                __builtin_nontemporal_store(((const f4v *)(R + (size_t)(i0 + src) * 16))[q], &((f4v *)(trows + pos * 16))[q]);
                if (q == 0u)
                    __builtin_nontemporal_store(((u64)c << 32) | (u64)(unsigned)(row0 + i0 + src), &tmeta[pos]);
            }
        }
    } else {
        for (unsigned slot = threadIdx.x; slot < nrows; slot += 256u) {
            const unsigned src = s_order[slot];
            const unsigned c = s_code[src];
            const unsigned bk = c >> bshift;
            const unsigned inb = s_base[bk] + (slot - s_start[bk]);
            if (inb < cap) {
                const size_t pos = (size_t)bk * cap + inb;
                const float *__restrict__ x = R + (size_t)(i0 + src) * g.k;
                float *__restrict__ t = trows + pos * 16;
                for (int d = 0; d < g.k; ++d)
                    t[d] = x[d];
                tmeta[pos] = ((u64)c << 32) | (u64)(unsigned)(row0 + i0 + src);
            }
        }
    }
}

// C1': the cells' tile ranges and the scan's items from the cell counts, on the device (rounds 2-4: counts to the host, prefix
// there, tile ranges and items back).  One cell per thread, 1024 cells per block: block b first adds up what the cells in
// front of its own need (every block reads the counts before its range again — 8 MB out of the L2 in all at 2^16 cells — instead
// of waiting for its neighbours), then scans its own 1024.  Everything is read and written with consecutive lanes on
// consecutive words (a first form — one block, 64 consecutive cells per thread — took 97-176 us: 64 lines per instruction
// from ONE compute unit).  res = {tiles, items, rows of the largest cell}; the caller zeroes counts[] afterwards (it
// becomes the placement's fill counters).
__global__ __launch_bounds__(1024) void knn_cells_prefix_kernel(const unsigned *__restrict__ counts, unsigned ncells,
                                                                unsigned *__restrict__ tile_start, u64 *__restrict__ items,
                                                                unsigned *__restrict__ res)
{
    __shared__ unsigned s_t[1024], s_i[1024], s_red[3][16];
    const unsigned first = blockIdx.x * 1024u, c = first + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    unsigned bt = 0u, bi = 0u;   // what the cells in front of this block need
    for (unsigned j = threadIdx.x; j < first; j += 1024u) {
        const unsigned t = (counts[j] + 31u) / 32u;
        bt += t;
        bi += (t + KNN_CELL_ITEM_TILES - 1u) / KNN_CELL_ITEM_TILES;
    }
    const unsigned rows = c < ncells ? counts[c] : 0u;
    const unsigned tiles = (rows + 31u) / 32u, nit = (tiles + KNN_CELL_ITEM_TILES - 1u) / KNN_CELL_ITEM_TILES;
    unsigned big = rows;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        bt += (unsigned)__shfl_xor((int)bt, off, KNN_WAVE);
        bi += (unsigned)__shfl_xor((int)bi, off, KNN_WAVE);
        big = max(big, (unsigned)__shfl_xor((int)big, off, KNN_WAVE));
    }
    if (lane == 0u) {
        s_red[0][w] = bt;
        s_red[1][w] = bi;
        s_red[2][w] = big;
    }
    s_t[threadIdx.x] = tiles;
    s_i[threadIdx.x] = nit;
    __syncthreads();
    bt = bi = big = 0u;
    for (int j = 0; j < 16; ++j) {
        bt += s_red[0][j];
        bi += s_red[1][j];
        big = max(big, s_red[2][j]);
    }
    for (unsigned off = 1u; off < 1024u; off <<= 1) {   // inclusive scans of the block's own cells (Hillis-Steele, 10 steps)
        const unsigned a = threadIdx.x >= off ? s_t[threadIdx.x - off] : 0u, b = threadIdx.x >= off ? s_i[threadIdx.x - off] : 0u;
        __syncthreads();
        s_t[threadIdx.x] += a;
        s_i[threadIdx.x] += b;
        __syncthreads();
    }
    if (c < ncells) {
        const unsigned tb = bt + s_t[threadIdx.x] - tiles;
        unsigned ib = bi + s_i[threadIdx.x] - nit;
        tile_start[c] = tb;
        for (unsigned t0 = 0u; t0 < tiles; t0 += KNN_CELL_ITEM_TILES)   // (one item per cell on data the cuts fit)
            items[ib++] = ((u64)c << 48) | ((u64)min((unsigned)KNN_CELL_ITEM_TILES, tiles - t0) << 40) | (u64)(tb + t0);
    }
    if (threadIdx.x == 0u)
        atomicMax(&res[2], big);
    if (c == ncells - 1u) {   // the last cell's thread has the totals
        tile_start[ncells] = bt + s_t[threadIdx.x];
        res[0] = bt + s_t[threadIdx.x];
        res[1] = bi + s_i[threadIdx.x];
    }
}

// C1: rows per cell from the buckets' records (a bucket's cells are consecutive: an LDS histogram per block).
// (bucket_fill, nullable: the fast build's buckets have fixed room — bucket b holds records [bucket_start[b], + bucket_fill[b]))
__device__ __forceinline__ unsigned bucket_end(const unsigned *__restrict__ bucket_start, const unsigned *__restrict__ bucket_fill, unsigned b)
{
    const unsigned r0 = bucket_start[b], room = bucket_start[b + 1] - r0;
    return bucket_fill ? r0 + min(bucket_fill[b], room) : r0 + room;
}

__global__ __launch_bounds__(256) void knn_cells_bucket_cellcount_kernel(const u64 *__restrict__ tmeta,
                                                                         const unsigned *__restrict__ bucket_start, int bshift,
                                                                         unsigned *__restrict__ counts,
                                                                         const unsigned *__restrict__ bucket_fill)
{
    __shared__ unsigned s_h[256];   // cells of one bucket (2^bshift <= 256: bits <= 16)
    const unsigned b = blockIdx.x / CELL_PLACE_PARTS, part = blockIdx.x % CELL_PLACE_PARTS;
    s_h[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned r0 = bucket_start[b], r1 = bucket_end(bucket_start, bucket_fill, b);
    const unsigned len = (r1 - r0 + CELL_PLACE_PARTS - 1u) / CELL_PLACE_PARTS;
    const unsigned a = min(r0 + part * len, r1), e = min(a + len, r1);
    for (unsigned i = a + threadIdx.x; i < e; i += 256u)
        atomicAdd(&s_h[(unsigned)(tmeta[i] >> 32) & ((1u << bshift) - 1u)], 1u);
    __syncthreads();
    if (threadIdx.x < (1u << bshift) && s_h[threadIdx.x] != 0u)
        atomicAdd(&counts[((size_t)b << bshift) + threadIdx.x], s_h[threadIdx.x]);
}

// C2: a bucket's records to their places in the layout: what knn_cells_scatter_frag_kernel writes for a row (same
// arithmetic, same outlier rule).  All CELL_PLACE_PARTS blocks of a bucket run on ONE XCD (blocks go to the XCDs round robin),
// 32 buckets per XCD one after the other.  A block walks its records in runs of 4096: cells counted in LDS, the run's room
// in every cell reserved with one atomic per cell, ranks from LDS.
__global__ __launch_bounds__(256) void knn_cells_place_kernel(
    const float *__restrict__ trows, const u64 *__restrict__ tmeta, const unsigned *__restrict__ bucket_start, int k, int bshift,
    const unsigned *__restrict__ tile_start, unsigned *__restrict__ fill, const float *__restrict__ center, float sigma,
    h8 *__restrict__ frag, float *__restrict__ norms, unsigned *__restrict__ norms2, unsigned *__restrict__ perm,
    unsigned *__restrict__ out, unsigned *__restrict__ olist, unsigned ocap, const unsigned *__restrict__ bucket_fill)
{
    __shared__ unsigned s_h[256], s_start[256], s_cur[256], s_pos[256], s_wsum[4];
    __shared__ unsigned short s_order[CELL_BUILD_ROWS];
    __shared__ float s_c[16];
    const unsigned xcd = blockIdx.x & 7u, jb = blockIdx.x >> 3;
    const unsigned b = (jb / CELL_PLACE_PARTS) * 8u + xcd, part = jb % CELL_PLACE_PARTS;
    const unsigned r0 = bucket_start[b], r1 = bucket_end(bucket_start, bucket_fill, b);
    const unsigned len = (r1 - r0 + CELL_PLACE_PARTS - 1u) / CELL_PLACE_PARTS;
    const unsigned a = min(r0 + part * len, r1), e = min(a + len, r1);
    const unsigned cmask = (1u << bshift) - 1u;
    if (threadIdx.x < 16)
        s_c[threadIdx.x] = threadIdx.x < (unsigned)k ? center[threadIdx.x] : 0.0f;
    float vmax = 0.0f, nmaxv = 0.0f;
    for (unsigned run = a; run < e; run += CELL_BUILD_ROWS) {   // (block-uniform)
        __syncthreads();   // the previous run's tables are done with
        s_h[threadIdx.x] = 0u;
        __syncthreads();
        unsigned cl[CELL_BUILD_ROWS / 256];
#pragma unroll
        for (int it = 0; it < CELL_BUILD_ROWS / 256; ++it) {
            const unsigned i = run + it * 256 + threadIdx.x;
            cl[it] = i < e ? (unsigned)(tmeta[i] >> 32) & cmask : 0u;
            if (i < e)
                atomicAdd(&s_h[cl[it]], 1u);
        }
        __syncthreads();
        block_prefix_256(s_h, s_start, s_wsum);
        {
            const unsigned cnt = s_h[threadIdx.x];   // thread t: cell t of the bucket
            const unsigned c = (b << bshift) + threadIdx.x;
            s_pos[threadIdx.x] = cnt != 0u ? tile_start[c] * 32u + atomicAdd(&fill[c], cnt) : 0u;
            s_cur[threadIdx.x] = s_start[threadIdx.x];
        }
        __syncthreads();
        // the run's records in CELL order (counting sort of their numbers): the ~16 a cell gets go to consecutive places
#pragma unroll
        for (int it = 0; it < CELL_BUILD_ROWS / 256; ++it)
            if (run + it * 256 + threadIdx.x < e)
                s_order[atomicAdd(&s_cur[cl[it]], 1u)] = (unsigned short)(it * 256 + threadIdx.x);
        __syncthreads();
        const unsigned nrec = min((unsigned)CELL_BUILD_ROWS, e - run);
        for (unsigned slot = threadIdx.x; slot < nrec; slot += 256u) {
            const unsigned i = run + s_order[slot];
            const u64 meta = tmeta[i];
            const unsigned c = (unsigned)(meta >> 32), row = (unsigned)meta;
            const f4v *__restrict__ x4 = (const f4v *)(trows + (size_t)i * 16);
            bool real = true;
            float nrm = 0.0f, vm = 0.0f;
            h8 v[2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const f4v xa = x4[2 * half], xb = x4[2 * half + 1];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int d = half * 8 + jj;
                    const float xv = jj < 4 ? xa[jj] : xb[jj - 4];
                    const float sc = d < k ? (xv - s_c[d]) * sigma : 0.0f;   // fp32 subtract, exact power-of-two scale
                    const _Float16 hval = (_Float16)sc;                      // round to nearest even
                    const float back = (float)hval;
                    real = real && fabsf(back) <= 1.0f;                      // outside the robust box, NaN included
                    vm = fmaxf(vm, fabsf(back));
                    nrm = nrm + back * back;                                 // exact products, fp32 sum
                    v[half][jj] = hval;
                }
            }
            if (!real) {   // out of the filter (zero fragment, +INF norm), into the exact list
                const unsigned opos = atomicAdd(&out[3], 1u);
                if (opos < ocap)
                    olist[opos] = row;
                v[0] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
                v[1] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
                vm = 0.0f;
                nrm = 0.0f;
            }
            const size_t pos = (size_t)s_pos[c & cmask] + (slot - s_start[c & cmask]);
            frag[(pos >> 5) * 64 + (pos & 31)] = v[0];
            frag[(pos >> 5) * 64 + 32 + (pos & 31)] = v[1];
            norms[pos] = real ? nrm : INFINITY;
            norms2[pos] = pack_norm22(real ? nrm : INFINITY);
            perm[pos] = row;
            vmax = fmaxf(vmax, vm);
            nmaxv = fmaxf(nmaxv, nrm);
        }
    }
    vmax = wave_max_f(vmax);
    nmaxv = wave_max_f(nmaxv);
    __shared__ float s_v[4], s_n[4];
    if ((threadIdx.x & 63) == 0) {
        s_v[threadIdx.x >> 6] = vmax;
        s_n[threadIdx.x >> 6] = nmaxv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        guarded_atomic_max(&out[0], __float_as_uint(fmaxf(fmaxf(s_v[0], s_v[1]), fmaxf(s_v[2], s_v[3]))));
        guarded_atomic_max(&out[1], __float_as_uint(fmaxf(fmaxf(s_n[0], s_n[1]), fmaxf(s_n[2], s_n[3]))));
    }
}

// The padding of every cell's last tile: positions [rows of the cell, its tiles x 32) get what the scan must see there — zero
// fragment, +INF norm (never a score under a threshold), no row.  32 threads per cell, after the placement (fill[c] = rows
// placed).  Rounds 2-3 memset the whole layout first: 0.76 GB of writes for 46 MB of padding at C3.
__global__ __launch_bounds__(256) void knn_cells_pad_kernel(const unsigned *__restrict__ tile_start, const unsigned *__restrict__ fill,
                                                            unsigned ncells, h8 *__restrict__ frag, float *__restrict__ norms,
                                                            unsigned *__restrict__ norms2, unsigned *__restrict__ perm, int kt, int nif)
{
    const unsigned c = (blockIdx.x * blockDim.x + threadIdx.x) >> 5, j = threadIdx.x & 31u;   // 32 threads per cell: at most 31 pads
    if (c >= ncells)
        return;
    const size_t pos = (size_t)tile_start[c] * 32 + fill[c] + j, end = (size_t)tile_start[c + 1u] * 32;
    if (pos < end) {
        for (int t = 0; t < 2 * kt; ++t) {   // (kt K-steps x two halves of 32 lanes)
            h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            if (nif && t == 3)   // (kt = 2, k <= 30: K-slot 30 carries the norm — a padding row's: 65504, above every threshold)
                z[6] = __builtin_bit_cast(_Float16, (unsigned short)0x7BFFu);
            frag[(pos >> 5) * 64 * kt + t * 32 + (pos & 31)] = z;
        }
        norms[pos] = INFINITY;
        norms2[pos] = 0x00007C00u;
        perm[pos] = 0xFFFFFFFFu;
    }
}

__device__ __forceinline__ float min_tree16(const f16v &x, float seed)
{
    const float m0 = min3f(x[0], x[1], x[2]);
    const float m1 = min3f(x[3], x[4], x[5]);
    const float m2 = min3f(x[6], x[7], x[8]);
    const float m3 = min3f(x[9], x[10], x[11]);
    const float m4 = min3f(x[12], x[13], x[14]);
    const float m5 = min3f(m0, m1, m2);
    const float m6 = min3f(m3, m4, x[15]);
    return min3f(m5, m6, seed);
}

// Cell-major matching: a block of 8 waves owns cells 64b .. 64b+63 (one high-table entry, 64 consecutive
// low-table entries).  Pass 1 (queries on the lanes, an eighth of the batch per wave): which queries get past
// the high table alone — about a third for uniform data — compacted into an LDS queue.  Pass 2 (cells on
// the lanes, the queue cut into one run per wave): the low-table entry of each queued query, 16 loads in
// flight per wave; survivors go to the cell's list at offsets prefix-summed over the waves.  cell_counts[c] = queries
// that could not rule cell c out, lists[c][0..) = their numbers.  No global atomics: per-cell
// appends with returning atomics ran at 22 per ns, 0.13 ms for this batch (tools/atomic_probe).
#define CELL_MATCH_STAGED_CAP 640u
template <int CELL_MATCH_WAVES>   // 8, or 16 for shards of few cells (one block per 64 cells: 128 blocks at 2^13 cells)
__global__ __launch_bounds__(64 * CELL_MATCH_WAVES) void knn_cells_match_kernel(
    const float *__restrict__ lo_tab, const float *__restrict__ hi_tab, const float *__restrict__ dup, int m,
    int m_padded, CellGeom g, unsigned ncells, unsigned cap, unsigned short *__restrict__ lists,
    unsigned *__restrict__ cell_counts, unsigned *__restrict__ ctl,
    unsigned stage)   // entries of a list assembled in LDS (<= cap; 0: none): the launch's dynamic LDS holds 64 x (stage + 2) of them
{
    __shared__ unsigned short s_q[1024];
    __shared__ float s_hv[1024], s_dq[1024];
    __shared__ unsigned s_npass, s_flag, s_wcnt[CELL_MATCH_WAVES][64];
    // the first `stage` entries of every list are put together in LDS (dynamic: 64 rows of stage + 2 entries — a row stride of an
    // odd number of dwords, 65 at 128: lanes appending at the same depth hit different banks) and written out as whole rows;
    // what lies beyond goes straight to memory (k <= 16: stage = 128, lists are 25-125 entries long on uniform data).  (16 < k <= 32, lists of 384 / 640: written straight to memory —
    // ten million two-byte stores 768 bytes apart at k = 20 — the launch took 75 us at C3's shape; staged in LDS 0.326 -> 0.304 ms per step at k = 20, 0.810 -> 0.709 at k = 24)
    extern __shared__ __attribute__((aligned(16))) unsigned short s_list[];
    const unsigned lstride = stage + 2u;
    // (the flag is read ONCE per block: other blocks of this launch may raise it, and threads of one block must not
    // disagree about leaving before the barriers below)
    if (threadIdx.x == 0)
        s_flag = ctl[KNN_CTL_FALLBACK];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned c0 = blockIdx.x * 64u;
    const unsigned cell = c0 + (unsigned)lane;
    const int nl = 1 << g.sa;
    const unsigned h = c0 >> g.sa;                      // block-uniform: nl >= 64
    const unsigned l = cell & (unsigned)(nl - 1);
    const float *__restrict__ hrow = hi_tab + (size_t)h * m_padded;
    if (threadIdx.x == 0)
        s_npass = 0u;
    __syncthreads();
    if (s_flag != 0u)
        return;
    {   // pass 1: this wave's share of the batch (m <= 1024: 16 chunks of 64 queries over the waves)
        constexpr int U = 16 / CELL_MATCH_WAVES;
        float hv[U], dq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = (u * CELL_MATCH_WAVES + wib) * 64 + lane;
            hv[u] = q < m ? hrow[q] : INFINITY;
            dq[u] = q < m ? dup[q] : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = (u * CELL_MATCH_WAVES + wib) * 64 + lane;
            const bool pass = q < m && !(hv[u] > dq[u]);
            const u64 mask = __ballot(pass);
            if (mask != 0ull) {   // wave-uniform
                unsigned base = 0u;
                if (lane == 0)
                    base = atomicAdd(&s_npass, (unsigned)__popcll(mask));
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                if (pass) {
                    const unsigned pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                          __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    s_q[pos] = (unsigned short)q;
                    s_hv[pos] = hv[u];
                    s_dq[pos] = dq[u];
                }
            }
        }
    }
    __syncthreads();
    const unsigned npass = s_npass;
    unsigned short *__restrict__ my_lds = s_list + (size_t)lane * lstride, *__restrict__ my_mem = lists + (size_t)cell * cap;
    // pass 2: the queue in contiguous runs, one per wave (<= 1024 / WAVES entries); a lane keeps "my cell could not
    // rule entry j out" as bit j of a mask, the waves' counts are prefix-summed through LDS, and the second sweep
    // writes every survivor to its final place — no LDS atomics (16 waves adding to the same 64 counters cost
    // more than the table loads at 2^13 cells, where a list holds ~125 of 1024 queries), lists in query order.
    // (64 loads outstanding — a wave's whole run of the queue in one round trip instead of three — measured in round 4: 76-80
    // registers instead of 55-58, C3 one batch at a time 0.150 -> 0.155 ms, pipelined 0.1195 -> 0.1235.  16 stays.)
    constexpr int EPW = 1024 / CELL_MATCH_WAVES, INFLIGHT = 16;   // entries per wave; low-table loads outstanding
    const unsigned per = (npass + CELL_MATCH_WAVES - 1u) / CELL_MATCH_WAVES;
    const unsigned e_begin = min((unsigned)wib * per, npass), e_end = min(e_begin + per, npass);
    u64 keep[EPW / 64];
#pragma unroll
    for (int w64 = 0; w64 < EPW / 64; ++w64)
        keep[w64] = 0ull;
    for (unsigned e0 = e_begin; e0 < e_end; e0 += INFLIGHT) {
        float lo[INFLIGHT];
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) {
            const unsigned e = min(e0 + (unsigned)u, e_end - 1u);
            lo[u] = lo_tab[(size_t)s_q[e] * nl + l];
        }
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) {
            const unsigned e = e0 + (unsigned)u;
            if (e < e_end) {
                const float lb = lo[u] + s_hv[e];
                const unsigned j = e - e_begin;
                if (!(lb > s_dq[e])) {
#pragma unroll
                    for (int w64 = 0; w64 < EPW / 64; ++w64)
                        if ((int)(j >> 6) == w64)
                            keep[w64] |= 1ull << (j & 63u);
                }
            }
        }
    }
    unsigned mine = 0u;
#pragma unroll
    for (int w64 = 0; w64 < EPW / 64; ++w64)
        mine += (unsigned)__popcll(keep[w64]);
    s_wcnt[wib][lane] = mine;
    __syncthreads();
    unsigned pos = 0u, total = 0u;
    for (int w = 0; w < CELL_MATCH_WAVES; ++w) {
        const unsigned v = s_wcnt[w][lane];
        pos += w < wib ? v : 0u;
        total += v;
    }
#pragma unroll
    for (int w64 = 0; w64 < EPW / 64; ++w64)
        for (u64 bits = keep[w64]; bits != 0ull; bits &= bits - 1ull) {
            const unsigned e = e_begin + (unsigned)w64 * 64u + (unsigned)__builtin_ctzll(bits);
            if (pos < stage)
                my_lds[pos] = s_q[e];
            else if (pos < cap)
                my_mem[pos] = s_q[e];   // (beyond the staged part: rare where the lists are short, the rule where nothing is staged)
            ++pos;
        }
    if (wib == 0) {
        cell_counts[cell] = total;   // (> cap: the list is cut short and the scan scores the cell against the whole batch)
    }
    __syncthreads();
    if (stage != 0u) {
        for (int j = wib; j < 64; j += CELL_MATCH_WAVES) {
            unsigned cj = 0u;
            for (int w = 0; w < CELL_MATCH_WAVES; ++w)
                cj += s_wcnt[w][j];
            cj = min(cj, stage);
            for (unsigned d = (unsigned)lane; d * 2u < cj; d += 64u)
                ((unsigned *)(lists + (size_t)(c0 + (unsigned)j) * cap))[d] = ((const unsigned *)(s_list + (size_t)j * lstride))[d];
        }
    }
}

// The scan.  A block's waves share the batch's B operands and thresholds in LDS.  Per item a wave issues every load it
// needs at once (tiles, norms, list: one round trip), and the kernel is written lean (~80 VGPRs, 12 waves per block) so
// that 5-6 waves per SIMD each have an item in flight.  The C tile (norms) goes through a per-wave LDS window, four
// broadcast ds_read_b128 per tile and block of queries.  (Round 3 measured the C tile out of one extra MFMA on split norms
// instead: 115-138 VGPRs, slower at every size — tools/arms/README.md.)
#define CELL_SCAN_WAVES 12
#define CELL_SCAN_WAVES_KT2 16   // 16 < k <= 32: one block per CU (74 KiB of operands), 4 waves per SIMD, 128 registers each
#define CELL_SCAN_CHUNK 256   // slots of a block's share whose tile ranges and list lengths sit in LDS at a time (DYN)
#define CELL_INLINE_RERANK_MAX 64u   // records a scan wave re-ranks itself; a longer list is left to the tail kernel
#define CELL_PUBLISH_STEP 64u             // records between two publications of a wave's count to the batch's counter
#define CELL_BATCH_RECORD_LIMIT (1u << 19)   // records of a batch beyond which it goes to the exact evaluation of its listed pairs
#define CELL_SCAN_RUN 16      // DYN: consecutive items a block takes at a time; its next run lies gridDim.x runs further on

// One (tile, block of 32 listed queries) step: scores + min tree + threshold test -> hit mask.
template <int KT>
__device__ __forceinline__ u64 cell_tile_step(const h8 (&a)[KT], const f4v *__restrict__ my_nrm, int p, int half, const h8 (&b)[KT], float th)
{
    f16v d;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const f4v v = my_nrm[p * 8 + 2 * gq + half];
        d[4 * gq + 0] = v[0];
        d[4 * gq + 1] = v[1];
        d[4 * gq + 2] = v[2];
        d[4 * gq + 3] = v[3];
    }
#pragma unroll
    for (int kk = 0; kk < KT; ++kk)
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], b[kk], d, 0, 0, 0);
    const float mn = min_tree16(d, th);
    return __ballot(mn < th);
}

// 16 < k <= 30 (round 5): the norm rides in the fragment — K-slots 30 and 31 of a row hold its norm's two fp16 halves
// (pack_norm22: hi + mid 2^-11 = N to 2^-22 N, in rho since round 3), the scan's B operands 1 and 2^-11 there — so the two
// MFMAs of a step give N - 2 a.b out of a ZERO C tile: no norm window, no LDS reads for it.  Every (tile, block of 32 queries)
// step read 4 KiB of LDS for its C tile; at the 5-10 blocks per tile of these dimensions that was the scan's bound (k 20:
// 2.9 M steps x 4.2 KiB = 176 us of the LDS pipe's time under a 165 us HBM floor).
template <int KT>
__device__ __forceinline__ u64 cell_tile_step_nif(const h8 (&a)[KT], const h8 (&b)[KT], float th)
{
    f16v d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], zero_acc(), 0, 0, 0);
#pragma unroll
    for (int kk = 1; kk < KT; ++kk)
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], b[kk], d, 0, 0, 0);
    const float mn = min_tree16(d, th);
    return __ballot(mn < th);
}

// |coordinate| a query may have in a cell's frame: twice that must stay an fp16 number (the B operand is -2 x the query); the
// bounds do not mind the size — every error term is relative to the pair's own amax (64 box half-widths at the largest scale).
#define CELL_FRAME_AMAX 16384.0f

// The centred scan's B operand and threshold for one (query, cell) pair (per-cell frames, knn_cells_recentre): lane = (query
// of the block, half of the dimensions).  The query is rounded in the CELL's frame exactly as knn_cells_prep_kernel rounds it
// in the shard's — fp32 subtract, exact power-of-two scale, fp16 to nearest even, times -2 —; amax and the computed norm mq
// are the pair's own.  The threshold is knn_threshold's last line with the constants of knn_bound_consts(k, 1, scale_c, amax,
// bmax_c, nmax_c), evaluated in fp32 with everything rounded towards "pass":
//     thr = Dup + 2 eta sqrt(Dup) + eta^2 + rho - mq (1 - gamma)          (Dup in the cell's units: Dup_q x ratio^2)
// P = the positive terms: eight fp32 operations on positive numbers, constants rounded up — within 2^-20 of exact, taken
// 10^-5 larger; mq x 0.999996 <= mq (1 - gamma)(1 - 10^-6)(1 - 2^-22) (gamma = 18 x 2^-24 = 1.07 x 10^-6; the 10^-6 is the slack
// knn_threshold gives its own double arithmetic: the fp32 value is never below the double one — tests/test_cells_logic.py checks
// 200 000 random pairs); the last subtraction rounds by at most 2^-24 (P + mq), 2.4 x 10^-7 (P + mq) is added.  A pair whose query does not fit the cell's frame: see the end.
__device__ __forceinline__ void cell_centred_operand(const float *__restrict__ s_q32, int k, unsigned qid, int half, bool valid,
                                                     const float (&cc)[8], float scale, float ratio, float bmaxc, float nmaxc,
                                                     float dupq, float sqdq, h8 &b, float &th)
{
#pragma clang fp contract(off)
    // the query's fp32 row, padded to 16 dimensions with zeros, from LDS (the centre is zero there too: no masks)
    const f4v x0 = *(const f4v *)(s_q32 + (size_t)qid * 16 + 8 * half), x1 = *(const f4v *)(s_q32 + (size_t)qid * 16 + 8 * half + 4);
    const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    float sc[8], a = 0.0f, mq = 0.0f;
    h8 hv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = (x[j] - cc[j]) * scale;
        hv[j] = (_Float16)sc[j];
        const float back = (float)hv[j];
        a = fmaxf(a, fabsf(back));   // (+INF where a coordinate left the fp16 range; a NaN — a query that is not finite: the prep kernel has sent its batch to the exact scan — is skipped)
        mq = mq + back * back;
    }
    a = fmaxf(a, __shfl_xor(a, 32, KNN_WAVE));
    mq = mq + __shfl_xor(mq, 32, KNN_WAVE);
    b = hv * (_Float16)-2.0f;   // exact: |coordinate| <= CELL_FRAME_AMAX = 2^14 below
    const float kf = (float)k;
    const float emax = 4.8865e-4f * (a + bmaxc) + 1.2220e-4f;          // theta' (amax + bmax) + 2 nu0
    const float sqk = k <= 1 ? 1.0f : k <= 4 ? 2.0f : k <= 9 ? 3.0f : 4.0f;   // >= sqrt(k), k <= 16 (no square root here: the exact kernels' ISA is checked for FMAs)
    const float eta = sqk * emax, eta2 = kf * emax * emax;
    const float rho = 1.1921e-5f * (nmaxc + 16.0f * a * a) + 1.79e-7f + 4.77e-7f * nmaxc;
    const float dupc = dupq * ratio * ratio, sqdc = sqdq * ratio;
    const float P = dupc + 2.0f * eta * sqdc + eta2 + rho;
    th = (P * 1.00001f + (P + mq) * 2.4e-7f + 1e-30f) - mq * 0.999996f;
    if (!(a <= CELL_FRAME_AMAX)) {   // (both lanes of a query take this together: `a` is the pair's)
        // The query does not fit the cell's frame (a coordinate beyond CELL_FRAME_AMAX cell units): along that coordinate it is
        // at least a32 (1 - 2^-22) - bmax_c (1 + 2^-10) - nu0 away from every row of the cell, in exact arithmetic.  Further than
        // sqrt(Dup): no row of this cell can be its answer (a `dense` cell is scored against the whole batch, queries that never
        // listed it included).  Else every real row passes — a ZERO operand leaves the rows' norms as scores (finite; an operand
        // of fp16 infinities would make them NaN, which no threshold passes) — and is re-ranked exactly, as any candidate is.
        float a32 = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            a32 = fmaxf(a32, fabsf(sc[j]));
        a32 = fmaxf(a32, __shfl_xor(a32, 32, KNN_WAVE));
        const float lb = a32 * 0.999f - bmaxc * 1.001f - 0.001f;
        const bool beyond = lb > 0.0f && lb * lb > dupc * 1.001f;   // (a32 = +INF: beyond any finite Dup)
        b = (h8){0, 0, 0, 0, 0, 0, 0, 0};
        th = dupq > -INFINITY && !beyond ? INFINITY : -INFINITY;
    }
    if (!valid)
        th = -INFINITY;
}

// ------------------------------------------------------------------------------------------
// Round 3: everything a batch needs before its cells can be matched, in ONE kernel — one block of 4 waves per query.
// Replaces knn_frag_kernel (queries) + knn_cells_seed_kernel + the keys-fill launch of round 2 (three launches, 26 us
// back to back at 1024 queries; this one 13).  Every wave rounds the query to its fp16 B operand (cheap, and each needs
// it), the block tabulates the squared gaps and the two pruning tables with all 256 threads, wave w scores seed cell w
// (own cell + the cells across the two nearest cuts) with all its tiles in flight at once, and thread 0 turns the best
// score into thr_q and Dup_q.  (One WAVE per query, the first form of this kernel, took 24 us: 34 seed tiles and 288
// table entries in one serial stream per query.)
// ------------------------------------------------------------------------------------------
#define CELL_PREP_WAVES 4
#define CELL_PREP_TILES 9     // seed tiles a wave requests at once (a cell of 144 .. 288 rows: one round trip)
#define CELL_SEED_MAX_TILES 36u   // tiles of one seed cell a query looks at (a larger cell: every stride-th tile)
#define CELL_SEED_MAX_TILES_CTR 144u   // the same with per-cell frames (clustered data: cells of thousands of tiles)
#define CELL_OUTER_SEED_TILES 2   // cell-range shards: tiles an outer seed cell of this rank contributes (the layer's depth by default)

// Where the seed tiles of cells OUTSIDE this index's range come from (cell-range shards; all zero otherwise): every rank's
// part of the replicated seed layer, `tiles` tiles per cell (ShardGeom::part_bytes).
#define KNN_MAX_RANKS 64
struct SeedLayer {
    const unsigned char *base;
    unsigned cpr, tiles;             // cells a part has room for (the largest rank's), tiles per cell
    unsigned nranks;
    unsigned long long part_bytes;
    unsigned first[KNN_MAX_RANKS + 1];   // first cell code of every rank's range (first[nranks] = the grid's cells)
};

// PW: waves per query, 4 or 2 (half the registers held while batches are in flight side by side, see knn_cells_query).
// SD: seed dimensions — the seeds are the query's own cell and every combination of moves across its SD nearest cuts:
//     2 = 4 cells (what ships; a cell of this index WHOLE, a cell of another rank through the seed layer's few tiles).  The
//     kernel is generic in SD: 4 = 16 cells leaves fewer survivors (profiles/r04_shard_sim.txt: 236 cells per query and
//     rank at N = 8 against 399, 199 with the bound one GPU would have) but measured slower end to end (knn_cells_query).
// KT: K-steps of a tile, 1 (k <= 16) or 2 (16 < k <= 32, round 5: the cuts, the gaps and the tables cover the first 16
//     dimensions; the fragments and the seed scores all of them)
// CTR: per-cell frames (knn_cells_recentre; KT = 1, no seed layer): every seed cell is scored with the query rounded in THAT
//     cell's frame, each gives its own bound Dup (frame-free: a squared distance) and the smallest stands; thr[q] = an upper
//     bound of sqrt(Dup_q) — what the centred scan's per-pair thresholds start from — instead of a score threshold
template <int PW, int SD, int KT = 1, bool CTR = false>
__global__ __launch_bounds__(64 * PW, KT == 1 && !CTR ? 4 : 3) void knn_cells_prep_kernel(   // (4 waves per SIMD: a batch of 1024 queries is resident at once)
    const float *__restrict__ Q, int m, int m_padded, CellGeom g, const float *__restrict__ bounds, double sigma2,
    const float *__restrict__ center, float sigma, const unsigned *__restrict__ tile_start, long long ntiles,
    const h8 *__restrict__ rf, const unsigned *__restrict__ rn2, SeedLayer layer, h8 *__restrict__ qfg, float *__restrict__ lo_tab,
    float *__restrict__ hi_tab, float bmax, float nmax, float amax_limit, float *__restrict__ thr,
    float *__restrict__ dup_out, unsigned *__restrict__ ctl, unsigned *__restrict__ ctl_next,
    unsigned *__restrict__ counts, unsigned nlists, u64 *__restrict__ keys_init,
    int lo_by_entry,   // != 0: the low table as [entry][query] (what the self-listing scan reads: a cell's row is contiguous)
    const float *__restrict__ frame, const unsigned *__restrict__ tile_cell)   // CTR only
{
#pragma clang fp contract(off)
    constexpr int SEEDS = 1 << SD, NS = SEEDS / PW;   // seed cells in all, per wave
    constexpr int PREP_TILES = KT == 1 ? CELL_PREP_TILES : 6;   // seed tiles a wave requests at once (KT KiB each)
    __shared__ float s_gap[16][CELL_MAX_BINS];
    __shared__ float s_red[PW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = blockIdx.x;
    // housekeeping folded in here to save launches: the record counters of the scan, the control words of the NEXT
    // batch on this slot (calls on a slot are stream-ordered; this batch's own words were cleared by the previous one)
    for (unsigned i = blockIdx.x * (64u * PW) + (unsigned)tid; i < nlists; i += gridDim.x * (64u * PW))
        counts[i] = 0u;
    if (blockIdx.x == 0 && tid == 0) {
        ctl_next[KNN_CTL_FALLBACK] = 0u;
        ctl_next[KNN_CTL_RECORDS] = 0u;
        ctl_next[KNN_CTL_WIDE_SEEDS] = 0u;
        ctl_next[KNN_CTL_DENSE_CELLS] = 0u;
        ctl_next[KNN_CTL_EXACT_CELLS] = 0u;
        ctl_next[KNN_CTL_SCAN_DONE] = 0u;
        ctl_next[KNN_CTL_DEFERRED] = 0u;
        ctl_next[KNN_CTL_TAIL_DONE] = 0u;
        ctl_next[KNN_CTL_TOTAL] = 0u;
    }
    const int half = lane >> 5;
    const size_t frag_at = (size_t)(qi >> 5) * 64 * KT + (size_t)half * 32 + (size_t)(qi & 31);   // (+ 64 per K-step)
    if (qi >= m) {   // padding query of the last tile (block-uniform): never listed, never passes
        if (wib == 0 && (lane & 31) == 0) {
#pragma unroll
            for (int t = 0; t < KT; ++t)
                qfg[frag_at + 64 * t] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
        }
        if (tid == 0) {
            thr[qi] = -INFINITY;
            dup_out[qi] = -INFINITY;
        }
        return;
    }
    if (keys_init && tid == 0)
        keys_init[qi] = kKeyInit;

    // ---- the query as an fp16 B operand (what knn_frag_kernel writes for a query row: centred, scaled, rounded,
    // times -2), its norm and largest coordinate.  Every lane does the whole row: the loads are wave-uniform.
    const float *__restrict__ qrow = Q + (size_t)qi * g.k;
    float nrm = 0.0f, amax = 0.0f;
    bool qbad = false;
    h8 bq[KT];
#pragma unroll
    for (int d = 0; d < 16 * KT; ++d) {
        float sc = 0.0f;
        if (d < g.k)
            sc = (qrow[d] - center[d]) * sigma;   // fp32 subtract, exact power-of-two scale
        const _Float16 hval = (_Float16)sc;       // round to nearest even
        const float back = (float)hval;
        qbad = qbad || !(fabsf(back) < INFINITY);
        amax = fmaxf(amax, fabsf(back));
        nrm = nrm + back * back;                  // exact products, fp32 sum in dimension order
        const _Float16 v = (_Float16)(back * -2.0f);
        qbad = qbad || !(fabsf((float)v) < INFINITY);
        if (((d >> 3) & 1) == half)
            bq[d >> 4][d & 7] = v;
    }
    if (wib == 0 && (lane & 31) == 0) {
#pragma unroll
        for (int t = 0; t < KT; ++t)
        {
            h8 o = bq[t];
            if (KT == 2 && t == 1 && half == 1 && g.k <= KNN_NIF_MAX_K) {   // K-slots 30, 31 of the scan's B operand: 1, 2^-11 (x the norm's halves)
                o[6] = (_Float16)1.0f;
                o[7] = __builtin_bit_cast(_Float16, (unsigned short)0x1000u);
            }
            qfg[frag_at + 64 * t] = o;   // for the scan (lanes 0 and 32 hold the two halves of every K-step)
        }
    }

    // ---- squared gaps to every bin of every dimension (scaled units, rounded down): 256 entries over the block's threads
    for (int e = tid; e < 256; e += 64 * PW) {
        const int d = e >> 4, b = e & 15;
        float v = 0.0f;
        if (d < g.k && g.nb[d] && b < (1 << g.nb[d])) {   // (d < 16: e < 256)
            const int nbins = 1 << g.nb[d];
            const float *__restrict__ bnd = bounds + d * (CELL_MAX_BINS - 1);
            const double q = (double)qrow[d];
            double gap = 0.0;
            if (b > 0 && (double)bnd[b - 1] > q)
                gap = (double)bnd[b - 1] - q;        // rows of the bin have x >= bnd[b-1] > q
            if (b < nbins - 1 && q > (double)bnd[b])
                gap = q - (double)bnd[b];            // rows of the bin have x < bnd[b] < q
            v = __double2float_rd(gap * gap * sigma2);
        }
        s_gap[d][b] = v;
    }
    // ---- seed cells (every wave works them out; wave w then takes cells w, w + PW, ...): dimensions on the lanes — the
    // query's own bin and the neighbouring bin nearest to it
    unsigned bin = 0u, alt = 0xFFFFFFFFu, nbl = 0u, shl = 0u;
    float ag = INFINITY;
    if (lane < g.k && lane < 16) {   // (the cells cut the first 16 dimensions)
        nbl = g.nb[lane];
        shl = g.shift[lane];
    }
    if (nbl) {
        const int nbins = 1 << nbl;
        const float *__restrict__ bnd = bounds + lane * (CELL_MAX_BINS - 1);
        const float q = qrow[lane];
        bin = cell_bin(bnd, nbins, q);
        if (bin > 0u) {
            alt = bin - 1u;
            ag = q - bnd[bin - 1];
        }
        if (bin + 1u < (unsigned)nbins && !(bnd[bin] - q >= ag)) {
            alt = bin + 1u;
            ag = bnd[bin] - q;
        }
        if (!(ag >= 0.0f))
            ag = 0.0f;
    }
    unsigned own = bin << shl;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1)
        own |= (unsigned)__shfl_xor((int)own, off, KNN_WAVE);
    own = (unsigned)__shfl((int)own, 0, KNN_WAVE);
    int pick[SD];
    u64 key = alt != 0xFFFFFFFFu ? ((u64)__float_as_uint(ag) << 32) | (u64)lane : ~0ull;
#pragma unroll
    for (int j = 0; j < SD; ++j) {
        u64 best = key;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const u64 o = __shfl_xor(best, off, KNN_WAVE);
            best = o < best ? o : best;
        }
        pick[j] = best == ~0ull ? -1 : (int)(best & 0xFFFFFFFFull);
        if (lane == pick[j])
            key = ~0ull;
    }
    unsigned code = own;   // lane s < SEEDS: the code of seed cell s (bit j of s = across the j-th nearest cut)
    bool ok = lane < SEEDS;
#pragma unroll
    for (int j = 0; j < SD; ++j) {
        const int pj = pick[j] < 0 ? 0 : pick[j];
        const unsigned pa = (unsigned)__shfl((int)alt, pj, KNN_WAVE), pn = (unsigned)__shfl((int)nbl, pj, KNN_WAVE),
                       ps = (unsigned)__shfl((int)shl, pj, KNN_WAVE);
        if ((lane >> j) & 1) {
            if (pick[j] < 0)
                ok = false;
            else
                code = (code & ~(((1u << pn) - 1u) << ps)) | (pa << ps);
        }
    }
    // the tiles of seed cell `lane` (requested now, used after the tables): its first fragment, its first norm word, how many.
    // A cell of this index: all its tiles, out of the layout; a cell of another rank (cell-range shards): the few tiles of
    // the replicated seed layer.
    unsigned long long v_fa = 0ull, v_na = 0ull;
    unsigned v_nt = 0u, v_cell = 0u;
    if (ok) {
        const unsigned l = code - g.cell_base;
        if (code >= g.cell_base && l < g.ncells) {
            const unsigned tb = tile_start[l];
            v_cell = l;
            v_nt = tile_start[l + 1u] - tb;
            // (a shard's OUTER seeds — beyond the own cell and the cells across the two nearest cuts — give what the layer
            // would: their first tiles.  Whole, the 16 local seed cells of a query that lives on this rank were 136 tiles
            // against the 32 of everybody else's, and the launch lasted as long as those blocks: 36 us against 15)
            if (SD > 2 && lane >= 4)
                v_nt = min(v_nt, (unsigned)CELL_OUTER_SEED_TILES);
            v_fa = (unsigned long long)(rf + (size_t)tb * 64 * KT);
            v_na = (unsigned long long)(rn2 + (size_t)tb * 32);
        } else if (KT == 1 && layer.base) {
            unsigned part = 0u;   // the rank whose range holds the cell (a table walk: no 64-bit divisions in here)
            for (unsigned r = 1u; r < layer.nranks; ++r)
                part += code >= layer.first[r] ? 1u : 0u;
            const unsigned cl = code - layer.first[part];
            const unsigned char *pb = layer.base + (size_t)part * layer.part_bytes + KNN_SEED_HEADER_BYTES;
            v_nt = layer.tiles;
            v_fa = (unsigned long long)(pb + (size_t)cl * layer.tiles * 1024u);
            v_na = (unsigned long long)(pb + (size_t)layer.cpr * layer.tiles * 1024u + (size_t)cl * layer.tiles * 128u);
        }
    }
    float fv_seed[NS];   // CTR: the frames of this wave's seed cells, word w on lane w (in flight while the tables are made)
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        fv_seed[c] = 0.0f;
        if constexpr (CTR) {
            const unsigned cell = (unsigned)__builtin_amdgcn_readlane((int)v_cell, wib + PW * c);
            if (lane < KNN_CELL_FRAME_WORDS)
                fv_seed[c] = frame[(size_t)cell * KNN_CELL_FRAME_WORDS + lane];
        }
    }
    __syncthreads();   // s_gap is complete
    // ---- the tables: double sums of the rounded-down gaps, rounded down again.  (Bits and positions of the dimensions
    // come from the lanes that hold them — nbl, shl above — as wave-uniform values: indexing the geometry struct with a
    // run-time d is a dependent scalar load from the kernel arguments per dimension and entry.)
    // Entry e of the high table belongs to local cells [e 2^sa, (e + 1) 2^sa): codes cell_base + that (cell_base is a
    // multiple of 2^sa; 0 unless the index is a cell-range shard).
    const int nl = 1 << g.sa, nh = (int)g.nh;   // nl >= 64: a wave's entries are all low or all high
    for (int e0 = 64 * wib; e0 < nl + nh; e0 += 64 * PW) {
        const int e = e0 + lane;
        const bool low = e0 < nl;   // wave-uniform
        const unsigned ecode = low ? (unsigned)e : ((g.cell_base >> g.sa) + (unsigned)(e - nl)) << g.sa;
        double sum = 0.0;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const unsigned nbd = (unsigned)__builtin_amdgcn_readlane((int)nbl, d);
            const unsigned shd = (unsigned)__builtin_amdgcn_readlane((int)shl, d);
            if (nbd != 0u && ((int)shd < g.sa) == low)   // wave-uniform
                sum += (double)s_gap[d][(ecode >> shd) & ((1u << nbd) - 1u)];
        }
        const float v = __double2float_rd(sum);
        if (e < nl + nh) {
            if (low)
                lo_tab[lo_by_entry ? (size_t)e * m_padded + qi : (size_t)qi * nl + e] = v;
            else
                hi_tab[(size_t)(e - nl) * m_padded + qi] = v;
        }
    }
    // ---- seed scores.  A wave's seed tiles — those of its NS cells, a cell of many tiles sampled (every stride-th,
    // at most CELL_SEED_MAX_TILES: any real row's score bounds the answer, and one query per MFMA against the thousands of
    // tiles of a cluster would cost more than the scan it prepares) — are ONE list, requested CELL_PREP_TILES at a time.
    float um = INFINITY;
    // (all wave-uniform) run c: `cnt[c]` tiles fa[c] + v stride[c] KiB, norm words na[c] + v stride[c] 128 B
    auto score_runs = [&](const unsigned long long (&fa)[NS], const unsigned long long (&na)[NS], const unsigned (&cnt)[NS],
                          const unsigned (&stride)[NS], const h8 (&bqx)[KT]) __attribute__((always_inline)) {
        unsigned start[NS + 1];   // run c holds positions [start[c], start[c + 1]) of the list (constant indices only: these
        start[0] = 0u;            // arrays must stay in registers — indexed by a run-time c they went to scratch memory)
#pragma unroll
        for (int c = 0; c < NS; ++c)
            start[c + 1] = start[c] + cnt[c];
        const unsigned total = start[NS];
        for (unsigned v0 = 0u; v0 < total; v0 += PREP_TILES) {
            h8 ar[PREP_TILES][KT];
            unsigned nw[PREP_TILES];
#pragma unroll
            for (int p = 0; p < PREP_TILES; ++p) {
                const unsigned v = v0 + (unsigned)p;   // position in the list -> (run, tile of the run)
                nw[p] = 0u;
                if (v < total) {
                    unsigned long long f = fa[0], nn = na[0];
                    unsigned st = stride[0], vv = v;
#pragma unroll
                    for (int c = 1; c < NS; ++c)
                        if (v >= start[c]) {   // (start[] ascends: the last run that matches is the one)
                            f = fa[c];
                            nn = na[c];
                            st = stride[c];
                            vv = v - start[c];
                        }
                    const size_t t = (size_t)vv * st;
#pragma unroll
                    for (int kk = 0; kk < KT; ++kk)
                        ar[p][kk] = ((const h8 *)f)[(t * KT + kk) * 64 + lane];
                    if (lane < 32)
                        nw[p] = ((const unsigned *)nn)[t * 32 + lane];
                }
            }
#pragma unroll
            for (int p = 0; p < PREP_TILES; ++p)
                if (v0 + (unsigned)p < total) {
                    f16v d = __builtin_amdgcn_mfma_f32_32x32x16_f16(norm_a_operand(nw[p]), norm_b_operand(), zero_acc(), 0, 0, 0);
#pragma unroll
                    for (int kk = 0; kk < KT; ++kk)
                        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[p][kk], bqx[kk], d, 0, 0, 0);
                    um = min_tree16(d, um);
                }
        }
    };
    if constexpr (CTR) {
        // One seed cell (or sampled tile) at a time: the query rounded in the cell's frame — what knn_frag_kernel would write
        // for it with (centre_c, scale_c) —, the cell's tiles scored against it, the bound on the answer's distance they give
        // in the SHARD's scaled units (the two frames differ by the power of two frame[17]).  All lanes do all of it.
        // fv: the cell's frame, word w on lane w (requested early — before the tables — for the seed cells)
        auto cell_bound = [&](float fv, unsigned long long f0, unsigned long long n0, unsigned cnt0,
                              unsigned stride0) __attribute__((always_inline)) -> float {
            float fr[KNN_CELL_FRAME_WORDS];
#pragma unroll
            for (int w_ = 0; w_ < KNN_CELL_FRAME_WORDS; ++w_)
                fr[w_] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fv), w_));
            const float scale = fr[16], ratio = fr[17];
            float nrmc = 0.0f, amaxc = 0.0f, n32 = 0.0f;
            bool badc = false;
            h8 bqc[KT];
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                float sc = 0.0f;
                if (d < g.k)
                    sc = (qrow[d] - fr[d]) * scale;
                n32 = n32 + sc * sc;
                const _Float16 hval = (_Float16)sc;
                const float back = (float)hval;
                badc = badc || !(fabsf(back) < INFINITY);
                amaxc = fmaxf(amaxc, fabsf(back));
                nrmc = nrmc + back * back;
                const _Float16 v = (_Float16)(back * -2.0f);
                badc = badc || !(fabsf((float)v) < INFINITY);
                if (((d >> 3) & 1) == half)
                    bqc[0][d & 7] = v;
            }
            // A query that does not fit this cell's frame (beyond CELL_FRAME_AMAX cell units: far from a tight cell): no fp16 scores — a
            // zero B operand leaves the rows' norms, finite iff the tiles hold a real row — and the bound is the triangle
            // inequality's: every row of the cell is within sqrt(k) bmax_c (1 + 2^-10) of its centre.
            const bool far = badc || !(amaxc <= CELL_FRAME_AMAX);
            if (far)
                bqc[0] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
            unsigned long long fa[NS], na[NS];
            unsigned cnt[NS], stride[NS];
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                fa[c] = na[c] = 0ull;
                cnt[c] = 0u;
                stride[c] = 1u;
            }
            fa[0] = f0;
            na[0] = n0;
            cnt[0] = cnt0;
            stride[0] = stride0;
            um = INFINITY;
            score_runs(fa, na, cnt, stride, bqc);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                um = fminf(um, __shfl_xor(um, off, KNN_WAVE));
            if (!(um < INFINITY))
                return INFINITY;
            const BoundConsts cst = knn_bound_consts(g.k, 1, scale, far ? 0.0f : amaxc, fr[18], fr[19]);
            double dup = 0.0;
            if (far) {
                const double reach = sqrt((double)n32) * (1.0 + 1e-6) + sqrt((double)g.k) * (double)fr[18] * 1.001 + 0.001;
                dup = reach * reach * (1.0 + 1e-5) * (1.0 + cst.g2) * (1.0 + cst.g2) + cst.sigma2 * cst.tau;
                if (!(dup < 1e300))
                    return INFINITY;
            } else {
                const float t = knn_threshold(cst, um, nrmc, &dup);
                if (!(t < INFINITY))
                    return INFINITY;
            }
            dup = dup / ((double)ratio * (double)ratio) * (1.0 + 1e-6);
            float df = (float)dup;
            if ((double)df < dup)
                df = nextafterf(df, INFINITY);
            return df;
        };
        float best = INFINITY;
#pragma unroll
        for (int c = 0; c < NS; ++c) {   // this wave's seed cells
            const int sl = wib + PW * c;
            const unsigned nt = (unsigned)__builtin_amdgcn_readlane((int)v_nt, sl);
            if (nt != 0u) {   // wave-uniform
                const unsigned long long f0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v_fa >> 32), sl) << 32) |
                                              (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)v_fa, sl);
                const unsigned long long n0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v_na >> 32), sl) << 32) |
                                              (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)v_na, sl);
                // (a fat cell — a whole cluster — leaves about (its rows / the rows sampled here) candidates per query: 1/56 of its
                // tiles, between 36 and 144, keeps a batch of 1024 near 16 records per scan wave.  n 2^24, 64 clusters: 36 tiles
                // 0.318 ms per step, 248 k records; 144 tiles 0.208, 60 k.  n 2^22: 36 tiles 0.105; 144 tiles 0.121 — the prep
                // kernel's extra 16 us buy nothing there)
                const unsigned scap = min(CELL_SEED_MAX_TILES_CTR, max(CELL_SEED_MAX_TILES, (nt * 9u) >> 9));
                const unsigned st0 = (nt + scap - 1u) / scap;
                best = fminf(best, cell_bound(fv_seed[c], f0, n0, (nt + st0 - 1u) / st0, st0));
            }
        }
        if (lane == 0)
            s_red[wib] = best;
        __syncthreads();
        float u = s_red[0];
#pragma unroll
        for (int i = 1; i < PW; ++i)
            u = fminf(u, s_red[i]);
        if (!(u < INFINITY) && ntiles > 0) {   // block-uniform: nothing in the seed cells — 64 tiles spread over the layout, each in its cell's frame
            __syncthreads();
            const unsigned total = (unsigned)(ntiles > 64 ? 64 : ntiles);
            const unsigned wstride = (unsigned)(ntiles > 64 ? ntiles / 64 : 1);
            best = INFINITY;
            for (unsigned i = (unsigned)wib * (64u / PW); i < min(((unsigned)wib + 1u) * (64u / PW), total); ++i) {
                const size_t t = (size_t)i * wstride;
                const float fv = lane < KNN_CELL_FRAME_WORDS ? frame[(size_t)tile_cell[t] * KNN_CELL_FRAME_WORDS + lane] : 0.0f;
                best = fminf(best, cell_bound(fv, (unsigned long long)(rf + t * 64), (unsigned long long)(rn2 + t * 32), 1u, 1u));
            }
            if (lane == 0)
                s_red[wib] = best;
            __syncthreads();
            u = s_red[0];
#pragma unroll
            for (int i = 1; i < PW; ++i)
                u = fminf(u, s_red[i]);
            if (tid == 0)
                atomicAdd(&ctl[KNN_CTL_WIDE_SEEDS], 1u);   // rare; statistics only
        }
        if (tid == 0) {
            const bool bad = qbad || !(amax <= amax_limit) || !(u < INFINITY);
            float sq = sqrtf(u);
            sq = nextafterf(nextafterf(sq, INFINITY), INFINITY);
            thr[qi] = bad ? -INFINITY : sq;
            dup_out[qi] = bad ? -INFINITY : u;
            if (bad)
                ctl[KNN_CTL_FALLBACK] = 1u;  // benign race: every writer stores 1
        }
        return;
    }
    {
        unsigned long long fa[NS], na[NS];
        unsigned cnt[NS], stride[NS];
#pragma unroll
        for (int c = 0; c < NS; ++c) {   // this wave's seed cells
            const int sl = wib + PW * c;
            const unsigned nt = (unsigned)__builtin_amdgcn_readlane((int)v_nt, sl);
            fa[c] = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v_fa >> 32), sl) << 32) |
                    (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)v_fa, sl);
            na[c] = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v_na >> 32), sl) << 32) |
                    (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)v_na, sl);
            stride[c] = (nt + CELL_SEED_MAX_TILES - 1u) / CELL_SEED_MAX_TILES;   // 1 up to the cap
            cnt[c] = nt == 0u ? 0u : (nt + stride[c] - 1u) / stride[c];
        }
        score_runs(fa, na, cnt, stride, bq);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)   // (every column is this query; the halves hold different rows)
        um = fminf(um, __shfl_xor(um, off, KNN_WAVE));
    if (lane == 0)
        s_red[wib] = um;
    __syncthreads();
    float u = s_red[0];
#pragma unroll
    for (int i = 1; i < PW; ++i)
        u = fminf(u, s_red[i]);
    if (!(u < INFINITY) && ntiles > 0) {   // block-uniform
        // nothing in the seed cells (a query in an empty corner of a clustered set): any real row gives a valid, if
        // loose, bound — look at 64 tiles spread over the whole layout, 64 / PW per wave
        __syncthreads();   // s_red has been read by everybody
        const unsigned total = (unsigned)(ntiles > 64 ? 64 : ntiles);
        const unsigned wstride = (unsigned)(ntiles > 64 ? ntiles / 64 : 1);
        const unsigned mine_first = (unsigned)wib * (64u / PW);
        um = INFINITY;
        if (mine_first < total) {
            unsigned long long fa[NS], na[NS];
            unsigned cnt[NS], stride[NS];
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                fa[c] = na[c] = 0ull;
                cnt[c] = 0u;
                stride[c] = 1u;
            }
            fa[0] = (unsigned long long)(rf + (size_t)mine_first * wstride * 64 * KT);
            na[0] = (unsigned long long)(rn2 + (size_t)mine_first * wstride * 32);
            cnt[0] = min(64u / PW, total - mine_first);
            stride[0] = wstride;
            score_runs(fa, na, cnt, stride, bq);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            um = fminf(um, __shfl_xor(um, off, KNN_WAVE));
        if (lane == 0)
            s_red[wib] = um;
        __syncthreads();
        u = s_red[0];
#pragma unroll
        for (int i = 1; i < PW; ++i)
            u = fminf(u, s_red[i]);
        if (tid == 0)
            atomicAdd(&ctl[KNN_CTL_WIDE_SEEDS], 1u);   // rare; statistics only
    }
    if (tid == 0) {
        bool bad = qbad || !(amax <= amax_limit);
        float t = -INFINITY, dupf = -INFINITY;
        if (!bad && !(u < INFINITY))
            bad = true;        // no row of the filter seen: cannot bound
        if (!bad) {
            const BoundConsts cst = knn_bound_consts(g.k, KT, sigma, amax, bmax, nmax);
            double dup = 0.0;
            t = knn_threshold(cst, u, nrm, &dup);
            if (!(t < INFINITY))
                bad = true;
            else {
                dup *= 1.0 + 1e-6;
                dupf = (float)dup;
                if ((double)dupf < dup)
                    dupf = nextafterf(dupf, INFINITY);
            }
        }
        thr[qi] = bad ? -INFINITY : t;
        dup_out[qi] = bad ? -INFINITY : dupf;
        if (bad)
            ctl[KNN_CTL_FALLBACK] = 1u;  // benign race: every writer stores 1
    }
}

// (<= 80 VGPRs: registers are handed out in eights and 6 waves x 88 do not fit a SIMD's 512)
// (HIP's second launch bound is waves per SIMD: 6 = at most 80 registers.  Left at 2 the allocator settled at 96-98 once the
// dense-cell and overflow paths were in — 4 waves per SIMD, and the kernel alone went from 0.035 to 0.042 ms at 2^21 rows)
// What ends a batch (the last block of the scan on a clean batch, else of the tail kernel): local row -> global id of the
// keys' index half when the shard's rows carry their own global numbers (cell-range shards, knn_index_create_sharded),
// and the int32 indices when the caller asked for them (knn_index_query: no separate unpack launch).
struct CellFinal {
    const unsigned *gids;   // nullable: keys hold base + row already
    int *out_idx;           // nullable
    int defer;              // != 0: more launches fold into the keys behind the scan (rows outside the robust box): the tail finalises
};

__device__ __forceinline__ void cells_finalize(u64 *__restrict__ keys, int m, const CellFinal &fin, unsigned tid, unsigned nthreads)
{
    if (!fin.gids && !fin.out_idx)
        return;
    for (unsigned i = tid; i < (unsigned)m; i += nthreads) {
        // (agent scope: the keys were folded by atomics of blocks on other XCDs; a plain load may see this XCD's stale L2 line)
        u64 key = __hip_atomic_load(&keys[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (fin.gids && (unsigned)(key >> 32) != 0x7F800000u) {   // (+INF, 0) = nothing found: v0's index 0 stays
            key = (key & 0xFFFFFFFF00000000ull) | (u64)fin.gids[(unsigned)key];
            keys[i] = key;
        }
        if (fin.out_idx)
            fin.out_idx[i] = (int)(unsigned)key;
    }
}

#ifdef KNN_SCAN_TIMELINE
// development only (tools/build_timeline_lib.sh builds a private copy of the library with this macro, tools/scan_timeline.py
// reads it): five stamps per wave of the last scan launch — entry, LDS filled, items done, own records re-ranked, exit — of
// the 100 MHz wall clock.  The product build compiles none of it.
__device__ unsigned long long g_scan_stamps[8192 * 5];
#define SCAN_STAMP(i)                                                                                  \
    do {                                                                                               \
        if ((threadIdx.x & 63) == 0 && blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) < 8192u)      \
            g_scan_stamps[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 5 + (i)] = wall_clock64(); \
    } while (0)
extern "C" int knn_debug_scan_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scan_stamps), sizeof(unsigned long long) * 8192 * 5);
}
#else
#define SCAN_STAMP(i) do { } while (0)
#endif
// K: 16 = compile-time dimension of the inline re-rank, 0 = run-time k <= 16
// SELF: the waves make the lists of their own items (cell_self_list, knn_exact_dev.h) — no match launch in front of the scan;
//       cell_counts / lists are unused and `cap` is CELL_SELF_CAP
// KT:   K-steps of a tile: 1 (k <= 16), 2 (16 < k <= 32, round 5: tiles of 2 KiB, two MFMAs per step, 64-byte B operands — 74 KiB
//       of LDS for a batch, so a CU holds ONE block: it has 16 waves of up to 128 registers, windows of nine tiles like KT = 1;
//       K = 0 there: the inline re-rank walks k in chunks of 16 dimensions)
// CTR:  per-cell frames (knn_cells_recentre; KT = 1, lists from the match launch): the B operand and the threshold of a (query,
//       cell) pair are made here, from the fp32 query, the cell's frame and Dup_q (cell_centred_operand) — s_thr holds the
//       batch's sqrt(Dup) bounds, s_dup the Dup values, the B operands' room in LDS stays unused
// NIF:  16 < k <= 30: the norms ride in the fragments' K-slots 30, 31 (cell_tile_step_nif): no norm window
template <bool DYN, int K, bool SELF, int KT = 1, bool CTR = false, bool NIF = false>
__global__ __launch_bounds__(64 * (KT == 1 && !CTR ? CELL_SCAN_WAVES : CELL_SCAN_WAVES_KT2), CTR ? 4 : KT == 1 ? 6 : 4) void knn_cells_scan_kernel(
    const h8 *__restrict__ rf, const float *__restrict__ rn, const u64 *__restrict__ items, unsigned nitems,
    const h8 *__restrict__ qfg, const float *__restrict__ thrg, int m, int m_padded,
    const unsigned *__restrict__ cell_counts, const unsigned short *__restrict__ lists, unsigned cap,
    u64 *__restrict__ rec, unsigned *__restrict__ counts, unsigned *__restrict__ ctl, unsigned slice,
    unsigned ovf_base, unsigned ovf_cap,
    // the exact re-rank of this wave's own records (v0 arithmetic on the fp32 rows, through perm) and the end of the batch
    const float *__restrict__ Q, const float *__restrict__ R, int krt, const unsigned *__restrict__ perm, long long npos,
    long long base, u64 *__restrict__ keys, CellFinal fin, CellSelf self)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(128))) unsigned char s_dyn[];   // (aligned: static LDS of the kernel sits in front of it, and the b128 reads below want 16-byte addresses)
    constexpr int TPP = CELL_TILES_PER_PASS;                         // reference tiles a wave holds in registers at a time
    constexpr int SW = KT == 1 && !CTR ? CELL_SCAN_WAVES : CELL_SCAN_WAVES_KT2;   // waves of a block
    constexpr int QB = CTR ? 64 : 32 * KT;                              // bytes of a query in LDS (CTR: its fp32 row, 16 dimensions)
    h8 *s_qf = (h8 *)s_dyn;                                             // [m_padded / 32][KT][64]; CTR: float s_q32[m_padded][16]
    float *s_thr = (float *)(s_dyn + (size_t)m_padded * QB);            // [m_padded]
    f4v *s_nrm = (f4v *)(s_dyn + (size_t)m_padded * (QB + 4));          // [waves][TPP * 8]
    // SELF: the batch's Dup values and one list room per wave behind the norm windows (knn_cells_scan_plan sizes it)
    float *s_dup = (float *)(s_dyn + (size_t)m_padded * (QB + 4) + (size_t)SW * TPP * 8 * sizeof(f4v));   // [m_padded]
    unsigned short *s_lists = (unsigned short *)(s_dup + m_padded);     // [waves][CELL_SELF_CAP]
    __shared__ unsigned s_flag;
    SCAN_STAMP(0);
    if (threadIdx.x == 0)
        s_flag = ctl[KNN_CTL_FALLBACK];   // read once per block: see knn_cells_match_kernel
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned wave = blockIdx.x * (unsigned)SW + (unsigned)wib, nwaves = gridDim.x * (unsigned)SW;
    __shared__ unsigned s_next, s_imeta[DYN ? CELL_SCAN_CHUNK : 1], s_itb[DYN ? CELL_SCAN_CHUNK : 1], s_inq[DYN ? CELL_SCAN_CHUNK : 1];
    const unsigned per_wave = (nitems + nwaves - 1u) / nwaves;
    // DYN: a block's share is one RUN of CELL_SCAN_RUN consecutive items out of every stripe of gridDim.x runs, not one
    // contiguous stretch of the item order.  Round 4, from the per-wave stamps of tools/scan_timeline.py at C3
    // (profiles/r04_scan_timeline.txt): with contiguous shares the blocks finished their items 71 to 109 us after the launch
    // began, later the further along the cell order their stretch lay (+14 us from the first to the last block of either half
    // of the grid: the work per cell drifts along the order — the side of a cut that got a percent more rows has more cells
    // that spill into a ninth tile).  A share that samples the whole order carries the average: the launch alone 0.1133 ->
    // 0.1085 ms, one batch at a time 0.1500 -> 0.1450 (three A/B pairs on one box); batches in flight fill the gaps either
    // way (0.1186-0.1237 / 0.1179-0.1204 per step).  All blocks still move through the layout together.
    // Which run of a stripe: rotated from stripe to stripe by a golden-ratio step — with the b-th run of every stripe a
    // block's cells would all share the code bits that number the run inside a stripe, and the drift follows the code bits.
    // Slot s of a share = item (stripe * gridDim.x + (blockIdx.x + stripe * rot) % gridDim.x) * RUN + s % RUN, stripe = s / RUN;
    // slots past the last item are holes.
    // (What is left after this: the 256 blocks placed second on their CUs finish 15 us behind the 256 placed first whatever
    // they are given — 87 against 102 us — and the XCDs differ by +-6 %.  A pool of items behind the shares that waves drain
    // through ONE counter in memory was tried for that: an agent-scope atomic on one address costs ~26 ns and they queue up —
    // 10 % of the items in the pool doubled the launch, 0.107 -> 0.190 ms.  tools/arms/scan_item_pool.patch.)
    // (everything derived from the arguments is recomputed where it is used — the fills, twice a launch: values kept across
    // the item loop for them went to scratch under the 80-register cap)
    const unsigned i0 = 0u;
    const unsigned i1 = DYN ? (((nitems + (unsigned)CELL_SCAN_RUN - 1u) / (unsigned)CELL_SCAN_RUN + gridDim.x - 1u) / gridDim.x) *
                                  (unsigned)CELL_SCAN_RUN
                            : 0u;   // the share, in slots
    auto slot_item = [&](unsigned s) {
        const unsigned nruns = (nitems + (unsigned)CELL_SCAN_RUN - 1u) / (unsigned)CELL_SCAN_RUN;
        const unsigned rot = __umulhi(gridDim.x, 2654435769u) | 1u;   // gridDim.x * 0.618...
        const unsigned stripe = s / (unsigned)CELL_SCAN_RUN;
        const unsigned run = stripe * gridDim.x + (blockIdx.x + stripe * rot) % gridDim.x;
        const unsigned it = run * (unsigned)CELL_SCAN_RUN + s % (unsigned)CELL_SCAN_RUN;
        return run < nruns && it < nitems ? it : 0xFFFFFFFFu;
    };
    // DYN: the first chunk's descriptors — two dependent round trips (item -> its cell's list length) — are requested BEFORE
    // the block fills its LDS (36 KiB of B operands and thresholds, one more round trip and a barrier): they overlap instead
    // of queueing up in front of the first tile load, and one barrier pair goes (round 4: 114.4 -> 112.5 us at C3, rocprofv3).
    // (The same for the fixed deal kept two more values alive across the fill: 16 bytes of scratch under the 80-register
    // cap, and a kernel that uses scratch at all ran 7 % slower — 0.1197 -> 0.1283 ms on one box.  Left as it was.)
    if constexpr (DYN) {
        const unsigned nc0 = min((unsigned)CELL_SCAN_CHUNK, i1 - i0);
        for (unsigned i = threadIdx.x; i < nc0; i += 64 * SW) {
            const unsigned it = slot_item(i0 + i);
            const u64 item = it != 0xFFFFFFFFu ? items[it] : 0ull;
            s_imeta[i] = (unsigned)(item >> 40);
            s_itb[i] = (unsigned)item;
            if constexpr (SELF)
                s_inq[i] = it != 0xFFFFFFFFu ? 1u : 0u;   // (the list is made when the item is taken)
            else
                s_inq[i] = it != 0xFFFFFFFFu ? cell_counts[(unsigned)(item >> 48)] : 0u;   // (a hole: nobody lists it)
        }
        if (threadIdx.x == 0)
            s_next = (unsigned)SW;   // the first item of every wave is its own number
    }
    // (LDS-DMA for this fill — no staging registers — measured: C3 -0.5 %, a rank of 8 +5 % per pipelined step.  Not taken.)
    if constexpr (CTR) {   // the batch's fp32 rows, padded to 16 dimensions (and to m_padded queries) with zeros
        const int kq = K > 0 ? K : krt;
        for (int i = threadIdx.x; i < m_padded * 16; i += 64 * SW)
            ((float *)s_dyn)[i] = (i >> 4) < m && (i & 15) < kq ? Q[(size_t)(i >> 4) * kq + (i & 15)] : 0.0f;
    } else {
        for (int i = threadIdx.x; i < m_padded * 2 * KT; i += 64 * SW)
            s_qf[i] = qfg[i];
    }
    for (int i = threadIdx.x; i < m_padded; i += 64 * SW)
        s_thr[i] = thrg[i];
    if constexpr (SELF || CTR)
        for (int i = threadIdx.x; i < m_padded; i += 64 * SW)
            s_dup[i] = self.dup[i];
    __syncthreads();
    if (s_flag != 0u)
        return;
    SCAN_STAMP(1);
    f4v *my_nrm = s_nrm + wib * (TPP * 8);
    unsigned short *my_list = s_lists + (SELF ? wib * (int)CELL_SELF_CAP : 0);

    u64 *__restrict__ my_rec = rec + (size_t)wave * slice;
    unsigned cnt = 0u;
    bool dead = false;                  // the shared area is over-full: stop scanning (wave-uniform)
    const int col = lane & 31, half = lane >> 5;
    // DYN = false: wave w takes items w, w + W, ... (an item = a run of tiles of one cell; uniform data: one item per cell):
    // all waves read one moving window of the layout (contiguous ranges per wave: +6 %).
    // DYN = true: the block owns a share of the items (see above) and its waves take them
    // one by one from a counter in LDS — with the fixed deal the busiest wave of a 2^21-row shard had 115 tile steps against
    // 48.5 on average (lists of 76..160 queries, cells of 5..9 tiles) and the launch lasted as long as that wave.
    // DYN: the chunk of the block's run whose tables are in LDS (the first one was filled above), this wave's item in it
    unsigned c0 = i0, nc = min((unsigned)CELL_SCAN_CHUNK, i1 - i0), mine_dyn = (unsigned)wib;
    for (unsigned g0 = 0u;; g0 += 64u) {
        unsigned v_meta = 0u, v_tb = 0u, v_nq = 0u;
        if constexpr (DYN) {
            if (g0 == 0u && c0 >= i1)
                break;
            if (g0 != 0u && mine_dyn >= nc) {   // block-uniform in effect: every wave runs dry before the barrier lets anyone on
                c0 += CELL_SCAN_CHUNK;
                if (c0 >= i1)
                    break;
                nc = min((unsigned)CELL_SCAN_CHUNK, i1 - c0);
                __syncthreads();   // everybody is done with the previous chunk's tables
                for (unsigned i = threadIdx.x; i < nc; i += 64 * SW) {
                    const unsigned it = slot_item(c0 + i);
                    const u64 item = it != 0xFFFFFFFFu ? items[it] : 0ull;
                    s_imeta[i] = (unsigned)(item >> 40);
                    s_itb[i] = (unsigned)item;
                    if constexpr (SELF)
                        s_inq[i] = it != 0xFFFFFFFFu ? 1u : 0u;
                    else
                        s_inq[i] = it != 0xFFFFFFFFu ? cell_counts[(unsigned)(item >> 48)] : 0u;
                }
                if (threadIdx.x == 0)
                    s_next = (unsigned)SW;   // the first item of every wave is its own number
                __syncthreads();
                mine_dyn = (unsigned)wib;
            }
            if (mine_dyn < nc && lane == 0 && !dead) {
                v_meta = s_imeta[mine_dyn];
                v_tb = s_itb[mine_dyn];
                v_nq = s_inq[mine_dyn];
            }
        } else {
            if (g0 >= per_wave || dead)
                break;
            // cells, list lengths and tile ranges of up to 64 items, one per lane
            // (measured and not kept, round 3: handing the cells out through an odd multiplier — cells w, w + W, ... share their
            // low bits and with them the queries that list them, the busiest wave has twice the average number of tile steps —
            // left the 2^21-row shard where it was and cost C3 4 %: the moving window over the layout is worth more than the balance)
            const unsigned mine = (g0 + (unsigned)lane) * nwaves + wave;
            const bool in = g0 + (unsigned)lane < per_wave && mine < nitems;
            const u64 item = in ? items[mine] : 0ull;
            v_meta = (unsigned)(item >> 40);   // cell << 8 | tiles
            v_tb = (unsigned)item;
            if constexpr (SELF)
                v_nq = in ? 1u : 0u;   // (the list is made when the item is taken)
            else
                v_nq = in ? cell_counts[v_meta >> 8] : 0u;
        }
        for (u64 todo = __ballot(v_nq != 0u); todo != 0ull && !dead; todo &= todo - 1ull) {
            const int j = (int)__builtin_ctzll(todo);
            const unsigned tb = (unsigned)__builtin_amdgcn_readlane((int)v_tb, j);
            const unsigned meta = (unsigned)__builtin_amdgcn_readlane((int)v_meta, j);
            const unsigned te = tb + (meta & 0xFFu);
            const unsigned cellj = meta >> 8;
            unsigned nq;
            if constexpr (SELF) {
                __builtin_amdgcn_wave_barrier();   // the previous item's reads of the list room are done
                nq = cell_self_list(self.lo_t, self.hi, self.sa, m_padded, cellj, m, s_dup, my_list, lane);
                wave_lds_sync();
                if (nq == 0u)   // (wave-uniform) nobody wants this cell
                    continue;
            } else {
                nq = (unsigned)__builtin_amdgcn_readlane((int)v_nq, j);
            }
            // a list longer than its room (a thousand copies of one query all want the same cells) is cut short by the
            // match kernel: the cell is then scored `dense`, against every query of the batch — what a list that long
            // asks for anyway — instead of sending the batch to the exact scan as round 2 did
            const bool dense = nq > cap;
            if (dense) {
                nq = (unsigned)m;
                if (lane == 0)
                    atomicAdd(&ctl[KNN_CTL_DENSE_CELLS], 1u);   // rare; statistics only
            }
            const unsigned short *__restrict__ list = SELF ? my_list : lists + (size_t)cellj * cap;
            // the first two blocks of the list travel with the tiles (one round trip per cell)
            // (SELF: the list is in LDS — every block of 32 is read from there)
            const unsigned l0 = SELF ? 0u : dense ? (unsigned)lane : (unsigned)list[min((unsigned)lane, nq - 1u)];
            // (round 3: blocks three and four of the list too — lists average 120 entries on the 2^21-row shards of an
            // 8-GPU run, and every block beyond the second was a dependent read from memory)
            const unsigned l1 = SELF ? 0u : dense ? 64u + (unsigned)lane : (unsigned)list[min(64u + (unsigned)lane, nq - 1u)];
            float ccv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, c_scale = 0.f, c_ratio = 0.f, c_bmax = 0.f, c_nmax = 0.f, th_kept = 0.f;
            h8 b_kept = {0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (CTR) {   // the cell's frame: this lane's half of the centre, the scale, the cell's bounds
                const float *__restrict__ fr = self.frame + (size_t)cellj * KNN_CELL_FRAME_WORDS;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    ccv[j] = fr[8 * half + j];
                c_scale = fr[16];
                c_ratio = fr[17];
                c_bmax = fr[18];
                c_nmax = fr[19];
            }
            for (unsigned t0 = tb; t0 < te && !dead; t0 += TPP) {
                const int nt = (int)min((unsigned)TPP, te - t0);   // wave-uniform
                h8 ar[TPP][KT];
#pragma unroll
                for (int p = 0; p < TPP; ++p)
                    if (p < nt) {
#pragma unroll
                        for (int kk = 0; kk < KT; ++kk)
                            ar[p][kk] = __builtin_nontemporal_load(&rf[((size_t)(t0 + (unsigned)p) * KT + kk) * 64 + lane]);
                    }
                if constexpr (!NIF) {
                    const f4v *__restrict__ rn4 = (const f4v *)rn + (size_t)t0 * 8;
                    const f4v n0 = lane < nt * 8 ? __builtin_nontemporal_load(&rn4[lane]) : (f4v){0.f, 0.f, 0.f, 0.f};
                    f4v n1 = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (TPP * 8 > 64)
                        n1 = 64 + lane < nt * 8 ? __builtin_nontemporal_load(&rn4[64 + lane]) : (f4v){0.f, 0.f, 0.f, 0.f};
                    __builtin_amdgcn_wave_barrier();   // the previous pass's reads of the window are done
                    if (TPP * 8 >= 64 || lane < TPP * 8)
                        my_nrm[lane] = n0;
                    if constexpr (TPP * 8 > 64)
                        if (lane < TPP * 8 - 64)
                            my_nrm[64 + lane] = n1;
                    wave_lds_sync();
                }
                for (unsigned q0 = 0u; q0 < nq && !dead; q0 += 32u) {
                    const unsigned idx = q0 + (unsigned)col;
                    const bool valid = idx < nq;
                    unsigned qid;
                    if constexpr (SELF) {
                        qid = dense ? (valid ? idx : 0u) : (unsigned)my_list[valid ? idx : 0u];
                    } else if (q0 < 64u) {
                        const unsigned from = __shfl(l0, (int)idx, KNN_WAVE);
                        qid = valid ? from : __shfl(l0, 0, KNN_WAVE);
                    } else if (q0 < 128u) {
                        const unsigned from = __shfl(l1, (int)(idx - 64u), KNN_WAVE);
                        qid = valid ? from : __shfl(l0, 0, KNN_WAVE);
                    } else {
                        qid = dense ? (valid ? idx : 0u) : (unsigned)list[valid ? idx : 0u];
                    }
                    h8 b[KT];
                    float th;
                    if constexpr (CTR) {
                        // (an item of several passes whose list is one block of queries — 16 queries per cluster cell on 64 tight
                        // clusters — keeps the block's operand from its first pass)
                        if (t0 == tb || nq > 32u)
                            cell_centred_operand((const float *)s_dyn, K > 0 ? K : krt, qid, half, valid, ccv, c_scale, c_ratio, c_bmax, c_nmax,
                                                 s_dup[qid], s_thr[qid], b_kept, th_kept);
                        b[0] = b_kept;
                        th = th_kept;
                    } else {
#pragma unroll
                        for (int kk = 0; kk < KT; ++kk)
                            b[kk] = s_qf[((qid >> 5) * KT + (unsigned)kk) * 64u + (unsigned)half * 32u + (qid & 31u)];
                        th = valid ? s_thr[qid] : -INFINITY;
                    }
                    // (a hit is recorded right behind its tile: parking the nine masks of a pass until its end, as round 2
                    // did, kept 18 registers busy with them — the allocator put the mask pairs in VGPRs)
#pragma unroll
                    for (int p = 0; p < TPP; ++p) {
                        if (p < nt) {
                            const u64 mask = NIF ? cell_tile_step_nif<KT>(ar[p], b, th) : cell_tile_step<KT>(ar[p], my_nrm, p, half, b, th);
                            if (__builtin_expect(mask != 0ull, 0)) {
                                const bool hit = (mask >> lane) & 1ull;
                                const unsigned pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                                     __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                                const u64 r = ((u64)qid << 32) | ((u64)(t0 + (unsigned)p) << 1) | (u64)half;
                                if (hit && pos < slice)
                                    my_rec[pos] = r;
                                const unsigned total = cnt + (unsigned)__popcll(mask);
                                if (total > slice && ovf_cap != 0u) {   // wave-uniform
                                    // this wave's slice is full (many queries of the batch want the same tile — copies of one
                                    // query): what does not fit goes to the area all waves share, ONE atomic per step for the
                                    // lanes that need room there, none once the area is over-full
                                    const unsigned first_over = max(cnt, slice);
                                    unsigned obase = ovf_cap;
                                    if (lane == 0 && !dead)
                                        obase = atomicAdd(&ctl[KNN_CTL_RECORDS], total - first_over);
                                    obase = (unsigned)__builtin_amdgcn_readfirstlane((int)obase);
                                    if (hit && pos >= slice) {
                                        const unsigned op = obase + (pos - first_over);
                                        if (op < ovf_cap)
                                            rec[(size_t)ovf_base + op] = r;   // (beyond: the re-rank sees the count and falls back)
                                    }
                                    // Slice full AND the shared area over-full (what the atomic returned says so; a plain read
                                    // of a word other XCDs are adding to can stay stale in this XCD's L2): the fp16 scores do not
                                    // separate this batch's rows (a cluster tighter than the fp16 step — every row of a query's
                                    // cells is a candidate).  The re-rank will see the count and hand the batch's listed pairs to
                                    // knn_cells_exact_kernel; nothing this wave still finds is needed.  The loops around the
                                    // steps look at `dead`; a jump out of the unrolled steps cost every step of every batch six
                                    // instructions of exec bookkeeping.
                                    dead = obase + (total - first_over) > ovf_cap;
                                }
                                const unsigned steps_new = (total / CELL_PUBLISH_STEP - cnt / CELL_PUBLISH_STEP) * CELL_PUBLISH_STEP;
                                cnt = total;
                                // Is this a batch the fp16 scores cannot separate (rows of a cluster tighter than the fp16 step:
                                // millions of candidates)?  Then it goes to the exact evaluation of its listed pairs, which costs
                                // the same whatever was recorded, and every record and tile step from here on is wasted — round 4
                                // noticed only when a wave's own slice AND the shared area were full, i.e. when all 6144 slices
                                // were (64 tight clusters: 3.6 M records, scan 69 -> 183 us).  Now: every 64th wave publishes its
                                // count in steps of CELL_PUBLISH_STEP records (a SAMPLE of the batch's total: a returning atomic on
                                // one word costs ~26 ns and they queue — all waves publishing took the scan to 0.4 ms); when the
                                // sample says CELL_BATCH_RECORD_LIMIT is passed, the publisher marks the shared area over-full, and
                                // every wave looks at that word (an agent-scope load, no atomic) whenever its own count crosses a
                                // step.  A clean batch — a few records per wave — never gets here.
                                // (not in the self-listing variant: its registers are all taken — 8 bytes of scratch with this in —
                                // and it serves shards of <= 2^13 cells one batch at a time, where the slices are large)
                                if (!SELF && steps_new != 0u && ovf_cap != 0u) {   // wave-uniform
                                    if ((wave & 63u) == 0u) {
                                        unsigned seen = 0u;
                                        if (lane == 0)
                                            seen = atomicAdd(&ctl[KNN_CTL_TOTAL], steps_new) + steps_new;
                                        seen = (unsigned)__builtin_amdgcn_readfirstlane((int)seen);
                                        if (seen > CELL_BATCH_RECORD_LIMIT / 64u) {
                                            if (lane == 0)
                                                atomicMax(&ctl[KNN_CTL_RECORDS], ovf_cap + 1u);   // what the tail kernel reads as "over-full"
                                            dead = true;
                                        }
                                    } else if (__hip_atomic_load(&ctl[KNN_CTL_RECORDS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > ovf_cap) {
                                        dead = true;
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
        if constexpr (DYN) {   // the next item of the chunk (an index past its end: the chunk is done for this wave)
            unsigned nx = 0u;
            if (lane == 0)
                nx = dead ? nc : atomicAdd(&s_next, 1u);
            mine_dyn = (unsigned)__builtin_amdgcn_readfirstlane((int)nx);
        }
    }
    SCAN_STAMP(2);
    const unsigned nrec = min(cnt, slice);   // what is IN the slice; the rest went to the shared area (wave-uniform)
    if (lane == 0)
        counts[wave] = nrec;                  // (statistics: knn_index_last_stats sums them)
    // ---- this wave's records, re-ranked on the spot (round 4; rounds 1-3 launched knn_rerank_kernel behind the scan: one
    // wave per list, 8-9 us for ~3000 records spread over 6144 lists).  16 lanes per record: its 16 rows with v0's
    // arithmetic on the fp32 rows, min-folded, ONE guarded atomic per record.  The loop above is over: its registers are free.
    // A LONG list is not re-ranked here: it is left where it is and the tail kernel, which sees the whole batch, either
    // re-ranks it with every CU or — when the shared area ended up over-full — drops it for the exact evaluation of the
    // batch's listed pairs.  (Round 4's first form re-ranked whatever a wave had: on 64 tight clusters (n 2^22) every wave
    // filled its slice of 586 records while the shared area was still filling, re-ranked them for 270 us on average — and the
    // batch then went to the exact evaluation anyway: scan 69 -> 566 us, step 0.32 -> 0.78 ms, profiles/r04_distribution_check.txt.)
    if (nrec > CELL_INLINE_RERANK_MAX && !dead && lane == 0)
        __hip_atomic_store(&ctl[KNN_CTL_DEFERRED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (every writer stores 1)
    if (nrec != 0u && nrec <= CELL_INLINE_RERANK_MAX && !dead) {
        // the records were stored by other lanes of this wave: wait for the stores, no more — workgroup scope is this CU's own
        // cache.  (An agent-scope fence here, __threadfence(), is a write-back AND an invalidate of the XCD's whole L2 — one per
        // wave with records: the scan took 261 us instead of 110 at C3 with it.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int k = K > 0 ? K : krt;
        for (unsigned c0 = 0u; c0 < nrec * 16u; c0 += 64u) {
            const unsigned c = c0 + (unsigned)lane;
            const bool live = c < nrec * 16u;
            // one dependent chain per pair — record -> position -> row number -> row — with everything that does not hang on
            // it (the query's row, its current key) requested up front
            const u64 e = live ? __hip_atomic_load(&my_rec[c >> 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ull;
            const unsigned qi = (unsigned)(e >> 32), lo = (unsigned)e, reg = c & 15u;
            const long long pos = (long long)(lo >> 1) * 32 + 8 * (reg >> 2) + 4 * (lo & 1u) + (reg & 3u);
            const unsigned row = live && pos < npos ? perm[pos] : 0xFFFFFFFFu;   // ~0u: padding position
            const float *__restrict__ qp = Q + (size_t)qi * k;
            constexpr int KD = K > 0 ? K : 16;
            const u64 cur = keys[qi];   // (may be stale: keys[] only ever decreases, a stale read costs a spare atomic)
            const float *__restrict__ rp = R + (size_t)(row != 0xFFFFFFFFu ? row : 0u) * k;
            float acc = 0.0f;
#pragma unroll
            for (int ch = 0; ch < KT; ++ch) {   // (KT = 2: dimensions 16 .. 31 behind 0 .. 15, the same registers again)
                float qv[KD], rv[KD];
#pragma unroll
                for (int d = 0; d < KD; ++d)
                    qv[d] = 16 * ch + d < k ? qp[16 * ch + d] : 0.0f;
#pragma unroll
                for (int d = 0; d < KD; ++d)
                    rv[d] = 16 * ch + d < k ? rp[16 * ch + d] : 0.0f;
#pragma unroll
                for (int d = 0; d < KD; ++d)
                    if (16 * ch + d < k) {   // v0's order and operations: diff, square, add (no contraction)
                        const float diff = qv[d] - rv[d];
                        const float sq = diff * diff;
                        acc = acc + sq;
                    }
            }
            u64 key = row != 0xFFFFFFFFu && acc < INFINITY ? pack_key(acc, (unsigned)(base + (long long)row)) : ~0ull;
            // (ds_swizzle's xor mode, not __shfl_xor: that one wants every lane's number in a register, computed at the top of
            // the kernel and kept across the item loop — 4 bytes of scratch under the 80-register cap)
#define KNN_SWZ_MIN(OFF)                                                                                   \
            {                                                                                              \
                const unsigned olo = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)key, ((OFF) << 10) | 0x1F);         \
                const unsigned ohi = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)(key >> 32), ((OFF) << 10) | 0x1F); \
                const u64 o = ((u64)ohi << 32) | (u64)olo;                                                 \
                key = o < key ? o : key;                                                                   \
            }
            KNN_SWZ_MIN(8)
            KNN_SWZ_MIN(4)
            KNN_SWZ_MIN(2)
            KNN_SWZ_MIN(1)
#undef KNN_SWZ_MIN
            if ((lane & 15) == 0 && key < cur)   // (key == ~0: never below a key)
                key_atomic_min(&keys[qi], key);
        }
    }
    // ---- end of the batch: the block that finishes last finalises it, unless something is still to fold into the keys
    // (records in the shared area, an over-full area, rows outside the box) — then the tail kernel does, which sees the
    // same two words and returns at once in the case handled here.
    // (What the finaliser reads from other blocks are words they changed with agent-scope ATOMICS — the keys, the record
    // counter, the DEFERRED flag: those are performed at the memory side, no cache to write back.  Every wave waits for its
    // own atomics to have been PERFORMED before the block counts itself done — cells_wait_own_atomics: the workgroup-scope
    // release fence alone compiles to `s_waitcnt lgkmcnt(0)` on gfx950 and leaves the no-return atomics in flight (ADVICE
    // r04; tests/test_host_logic.py reads the ISA for the vmcnt(0)) — and the finaliser reads with agent-scope loads.)
    __shared__ unsigned s_last;
    SCAN_STAMP(3);
    cells_wait_own_atomics();
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(&ctl[KNN_CTL_SCAN_DONE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    if (s_last != 0u && !fin.defer) {   // block-uniform
        const unsigned have = __hip_atomic_load(&ctl[KNN_CTL_RECORDS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) |
                              __hip_atomic_load(&ctl[KNN_CTL_DEFERRED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (have == 0u)
            cells_finalize(keys, m, fin, threadIdx.x, 64u * SW);
    }
    SCAN_STAMP(4);
}

// What is left of a batch behind the scan, in ONE gated launch (rounds 2-3: a re-rank launch and two gated ones):
//   nothing (the usual case: the scan's last block has finalised the batch)     -> every block returns at once
//   records in the shared overflow area                                        -> re-ranked here, all blocks striding
//   the area over-full (a cluster tighter than the fp16 step: the fp16 scores do not separate the batch's rows)
//                                                                              -> the batch's listed (item, query) pairs
//                                                                                 with the exact arithmetic (cells_exact_items)
//   FALLBACK (a query nothing bounds)                                          -> K = 16: the exact scan of the whole shard,
//                                                                                 here (exact_qreg_body); other k: the gated
//                                                                                 exact launch in front of this one has run
//   fin.defer                                                                  -> only the finalisation
// KT (with K = 0): run-time k <= 16 KT
template <int K, int KT = 1>
__global__ __launch_bounds__(KNN_BLOCK) void knn_cells_tail_kernel(
    const float *__restrict__ Q, const float *__restrict__ R, int krt, int m, long long nrows, long long npos, long long base,
    const u64 *__restrict__ items, unsigned nitems, const unsigned *__restrict__ cell_counts,
    const unsigned short *__restrict__ lists, unsigned cap, const unsigned *__restrict__ perm,
    const u64 *__restrict__ rec, unsigned ovf_base, unsigned ovf_cap, unsigned *__restrict__ ctl, u64 *__restrict__ keys,
    CellFinal fin,
    // the scan's record lists (counts[nlists], `slice` records each): lists longer than CELL_INLINE_RERANK_MAX are re-ranked here
    const unsigned *__restrict__ counts, unsigned nlists, unsigned slice,
    CellSelf self)   // lo_t != null: the scan made its own lists; the exact evaluation of the listed pairs makes them again
{
#pragma clang fp contract(off)
    __shared__ unsigned short s_tail_list[KNN_WAVES][CELL_SELF_CAP];
    const unsigned fb = ctl[KNN_CTL_FALLBACK], have = ctl[KNN_CTL_RECORDS];   // final: prep and the scan are complete
    const unsigned deferred = ctl[KNN_CTL_DEFERRED];
    if (!fin.defer && fb == 0u && have == 0u && deferred == 0u)
        return;
    if (fb != 0u) {
        if constexpr (K == 16) {
            // a query nothing bounds (not finite, far outside the box): v0's arithmetic over the whole shard, 512 queries per
            // block (two per lane), slices of 1024 rows strided over the blocks
            const unsigned qgroups = ((unsigned)m + 511u) / 512u, gx = gridDim.x / qgroups;
            if (gx != 0u && blockIdx.x < gx * qgroups)
                exact_qreg_body<16, 1>(Q, R, m, nrows, base, keys, 1024ll, blockIdx.x / qgroups, gx, blockIdx.x % qgroups);
        }
    } else {
        if (have > ovf_cap) {
            if (blockIdx.x == 0 && threadIdx.x == 0)
                ctl[KNN_CTL_EXACT_CELLS] = 1u;   // (statistics: knn_index_last_stats[2] = 2)
            cells_exact_items<K, KT>(Q, R, krt, m, base, items, nitems, cell_counts, lists, cap, perm, keys,
                                 blockIdx.x * (unsigned)KNN_WAVES + (threadIdx.x >> 6), gridDim.x * (unsigned)KNN_WAVES, self,
                                 &s_tail_list[threadIdx.x >> 6][0]);
        } else {
          const int k = K > 0 ? K : krt;
          if (deferred != 0u) {   // the long lists the scan's waves left alone: a block per list, 16 lanes per record
            for (unsigned l = blockIdx.x; l < nlists; l += gridDim.x) {
                const unsigned n = min(counts[l], slice);   // block-uniform
                if (n <= CELL_INLINE_RERANK_MAX)
                    continue;
                const u64 *__restrict__ lr = rec + (size_t)l * slice;
                const unsigned pairs = n * 16u;
                const unsigned padded = (pairs + KNN_BLOCK - 1) / KNN_BLOCK * KNN_BLOCK;
                for (unsigned c = threadIdx.x; c < padded; c += KNN_BLOCK) {
                    u64 key = ~0ull;
                    unsigned qi = 0u;
                    if (c < pairs)
                        key = rerank_pair<K>(Q, R, k, npos, base, lr[c >> 4], c & 15u, 0xFFFFu, 0u, perm, qi);
#pragma unroll
                    for (int off = 8; off > 0; off >>= 1) {
                        const u64 o = __shfl_xor(key, off, KNN_WAVE);
                        key = o < key ? o : key;
                    }
                    if ((threadIdx.x & 15u) == 0u && key != ~0ull && key < keys[qi])
                        key_atomic_min(&keys[qi], key);
                }
            }
          }
          if (have != 0u) {
            const u64 *__restrict__ ovf = rec + ovf_base;
            const unsigned pairs = have * 16u;
            const unsigned padded = (pairs + KNN_BLOCK - 1) / KNN_BLOCK * KNN_BLOCK;
            for (unsigned c = blockIdx.x * KNN_BLOCK + threadIdx.x; c < padded; c += gridDim.x * KNN_BLOCK) {
                u64 key = ~0ull;
                unsigned qi = 0u;
                if (c < pairs)
                    key = rerank_pair<K>(Q, R, k, npos, base, ovf[c >> 4], c & 15u, 0xFFFFu, 0u, perm, qi);
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) {
                    const u64 o = __shfl_xor(key, off, KNN_WAVE);
                    key = o < key ? o : key;
                }
                if ((threadIdx.x & 15u) == 0u && key != ~0ull && key < keys[qi])
                    key_atomic_min(&keys[qi], key);
            }
          }
        }
    }
    if (!fin.gids && !fin.out_idx)
        return;
    __shared__ unsigned s_last;
    cells_wait_own_atomics();   // (as in the scan: this wave's atomics have been performed)
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(&ctl[KNN_CTL_TAIL_DONE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    if (s_last != 0u)
        cells_finalize(keys, m, fin, threadIdx.x, KNN_BLOCK);
}

// ------------------------------------------------------------------------------------------
// Cell-range shards (round 4): the caller's partition pass, the replicated seed layer, the check of the rows' global numbers.
// ------------------------------------------------------------------------------------------
// owner[i] = the rank whose range of the global grid holds row i's cell
__global__ __launch_bounds__(256) void knn_geom_assign_kernel(const float *__restrict__ R, long long n, CellGeom g,
                                                              const float *__restrict__ bounds_arg, unsigned nranks,
                                                              int *__restrict__ owner)
{
    __shared__ float s_bnd[16 * (CELL_MAX_BINS - 1)];
    if (threadIdx.x < 16 * (CELL_MAX_BINS - 1))
        s_bnd[threadIdx.x] = bounds_arg[threadIdx.x];
    __syncthreads();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    unsigned c = 0u;
    for (int d = 0; d < g.k; ++d)
        if (g.nb[d])
            c |= cell_bin(s_bnd + d * (CELL_MAX_BINS - 1), 1 << g.nb[d], R[(size_t)i * g.k + d]) << g.shift[d];
    owner[i] = (int)knn_shard_owner(c, g.ncells, nranks, 1u << g.sa);   // (one 64-bit division per row: a build-time pass)
}

// This rank's part of the seed layer: for every cell of its range the first T tiles of the cell-sorted layout (fragments
// and split norms; a cell of fewer tiles is padded with zero fragments and +INF norms, which never give a score).
// One block per cell.
__global__ __launch_bounds__(256) void knn_cells_seed_export_kernel(const unsigned *__restrict__ tile_start, unsigned ncells,
                                                                    unsigned cpr, unsigned T, const h8 *__restrict__ rf,
                                                                    const unsigned *__restrict__ rn2,
                                                                    unsigned char *__restrict__ part, float bmax, float nmax)
{
    const unsigned cell = blockIdx.x;   // < cpr
    h8 *__restrict__ of = (h8 *)(part + KNN_SEED_HEADER_BYTES + (size_t)cell * T * 1024u);
    unsigned *__restrict__ on = (unsigned *)(part + KNN_SEED_HEADER_BYTES + (size_t)cpr * T * 1024u + (size_t)cell * T * 128u);
    unsigned tb = 0u, nt = 0u;
    if (cell < ncells) {
        tb = tile_start[cell];
        nt = min(T, tile_start[cell + 1u] - tb);
    }
    for (unsigned i = threadIdx.x; i < T * 64u; i += 256u)
        of[i] = i < nt * 64u ? rf[(size_t)tb * 64 + i] : (h8){0, 0, 0, 0, 0, 0, 0, 0};
    for (unsigned i = threadIdx.x; i < T * 32u; i += 256u)
        on[i] = i < nt * 32u ? rn2[(size_t)tb * 32 + i] : 0x00007C00u;
    if (cell == 0u && threadIdx.x == 0) {   // header: what the bound constants of every rank must cover
        ((float *)part)[0] = bmax;
        ((float *)part)[1] = nmax;
    }
}

__global__ __launch_bounds__(256) void knn_gids_check_kernel(const unsigned *__restrict__ gids, long long n, unsigned *__restrict__ bad)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i + 1 < n; i += stride)
        if (!(gids[i] < gids[i + 1]))
            atomicAdd(bad, 1u);
}

hipError_t knn_geom_assign_launch(const ShardGeom &sg, const float *rows_dev, long long n, int *owner_dev, hipStream_t s)
{
    if (n <= 0)
        return hipSuccess;
    CellGeom g;
    memset(&g, 0, sizeof g);
    g.k = sg.k;
    g.bits = sg.bits;
    g.sa = sg.sa;
    memcpy(g.nb, sg.nb, 16);
    memcpy(g.shift, sg.shift, 16);
    g.ncells = sg.ncells;
    float *bnd = nullptr;
    FTRY(KNN_DEV_ALLOC((void **)&bnd, sizeof sg.bounds));
    hipError_t e = hipMemcpyAsync(bnd, sg.bounds, sizeof sg.bounds, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(knn_geom_assign_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rows_dev, n, g, bnd,
                           (unsigned)sg.nranks, owner_dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);   // (the cuts' staging buffer goes back to the pool)
    (void)KNN_DEV_FREE(bnd);
    return e;
}

hipError_t knn_cells_seed_export(const FilterState &st, int rank, unsigned char *layer_dev, hipStream_t s)
{
    if (!st.cells || !st.cells->geom || !layer_dev)
        return hipErrorInvalidValue;
    const ShardGeom &sg = *st.cells->geom;
    hipLaunchKernelGGL(knn_cells_seed_export_kernel, dim3(sg.cells_per_rank), dim3(256), 0, s, st.cells->tile_start, st.cells->ncells,
                       sg.cells_per_rank, (unsigned)sg.seed_tiles, (const h8 *)st.ref_frags, st.ref_norms2,
                       layer_dev + (size_t)rank * sg.part_bytes(), st.bmax, st.nmax);
    return hipGetLastError();
}

hipError_t knn_gids_check(const unsigned *gids_dev, long long n, unsigned *bad_out, hipStream_t s)
{
    *bad_out = 0u;
    if (n < 2)
        return hipSuccess;
    unsigned *bad = nullptr;
    FTRY(KNN_DEV_ALLOC((void **)&bad, sizeof(unsigned)));
    hipError_t e = hipMemsetAsync(bad, 0, sizeof(unsigned), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(knn_gids_check_kernel, dim3(1024), dim3(256), 0, s, gids_dev, n, bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(bad_out, bad, sizeof(unsigned), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);
    (void)KNN_DEV_FREE(bad);
    return e;
}

// The global grid of a cell-range sharded set.  bits: as many as the rows allow (>= 144 rows per cell on average, the
// rule of a single index), at most 4 per dimension, and no more than lets every rank hold <= 2^16 cells; every rank's range
// starts at a multiple of 2^sa codes (whole entries of the high pruning table; 2^sa >= 64: whole blocks of the match pass).
bool knn_geom_cells(ShardGeom &g, int k, long long n_global, int nranks, const float *sample, long long samples, int seed_tiles)
{
    if (k < 1 || k > 16 || nranks < 1 || nranks > 64 || samples < 64 || seed_tiles < 1 || seed_tiles > 8 || n_global < 1)
        return false;
    int bits = std::min(cell_bits_for_rows(n_global), 4 * k);
    while (bits > 0 && ((1ull << bits) + (unsigned long long)nranks - 1ull) / (unsigned long long)nranks > 65536ull)
        --bits;
    if (bits < 9)
        return false;
    g.k = k;
    g.bits = bits;
    g.nranks = nranks;
    g.seed_tiles = seed_tiles;
    g.n_global = n_global;
    g.ncells = 1u << bits;
    cell_grid_shape(k, bits, g.nb, g.shift, &g.sa);
    if (g.sa < 6)
        return false;
    g.cells_per_rank = 0u;
    for (int r = 0; r < nranks; ++r) {   // every rank needs a layout of its own, of at most 2^16 cells
        const unsigned cr = g.cells_of(r);
        if (cr < 512u || cr > 65536u || cr % 64u != 0u)
            return false;
        g.cells_per_rank = std::max(g.cells_per_rank, cr);
    }
    cell_quantile_cuts(k, g.nb, sample, samples, g.bounds);
    return true;
}


// ------------------------------------------------------------------------------------------
// Per-cell frames (round 5; VERDICT r03 / r04: "centre fragments on their cell's box").  The shard's ONE frame rounds a
// coordinate to fp16 with an error of 2^-12 of the box; where the rows of a cell sit in a thousandth of the box (clustered
// data: 64 clusters of width 10^-3 — tools/distribution_check.py) that error is a quarter of the cluster's width, the
// thresholds admit most of the cluster (2 200 candidates per query) and the batch ends in the exact evaluation of its listed
// pairs.  In a frame of the CELL — fragment = fp16((row - centre_c) x scale_c), scale_c = sigma 2^e <= 2^8 sigma so that the
// cell's rows fill [-1, 1] — the same rounding is 2^-12 of the cell, and everything knn_filter_dev.h derives for (centre,
// sigma) holds for (centre_c, scale_c) word for word: the B operand of a (query, cell) pair is the query in THAT frame
// (built by the scan's lanes from the fp32 query, which the block keeps in LDS: knn_cells_scan_kernel<CTR>), its threshold comes from Dup_q — a squared
// DISTANCE, frame-free up to the power-of-two ratio — and the pair's own amax, the cell's bmax and nmax.
// Built as a pass over the finished cell-sorted layout (rows gathered through perm: ~3 ms per 2^24 rows — taken only when the
// build's sample says the data is clustered, or on request: knn_cells_recentre).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_cells_tile_cell_kernel(const unsigned *__restrict__ tile_start, unsigned ncells,
                                                                  unsigned *__restrict__ tile_cell)
{
    // 32 threads per cell (a cell of a clustered set can hold thousands of tiles)
    const unsigned c = (blockIdx.x * blockDim.x + threadIdx.x) >> 5, j = threadIdx.x & 31u;
    if (c >= ncells)
        return;
    for (unsigned t = tile_start[c] + j; t < tile_start[c + 1u]; t += 32u)
        tile_cell[t] = c;
}

// box[c][0..15] = min, box[c][16..31] = max of the cell's rows (ordered-uint images; rows outside the robust box — +INF norm —
// and padding positions do not count).  One wave per run of CELL_BOX_RUN consecutive tiles, lane = (row, half of the
// dimensions): the lanes keep their own minima while the run stays in one cell (tiles are in cell order) and fold them — 5
// shuffle steps, 16 guarded atomics per half — when the cell changes or the run ends.  (One wave per tile, folding every tile:
// 4.0 ms for the 131 072 tiles of 64 clusters — 32 atomics per tile on 64 cells' words; this form 0.2 ms.)
#define CELL_BOX_RUN 32u
__global__ __launch_bounds__(256) void knn_cells_box_kernel(const float *__restrict__ R, int k, const unsigned *__restrict__ perm,
                                                            const float *__restrict__ norms, unsigned ntiles,
                                                            const unsigned *__restrict__ tile_cell, unsigned *__restrict__ box)
{
    const unsigned t0 = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * CELL_BOX_RUN;
    const int lane = threadIdx.x & 63, half = lane >> 5;
    if (t0 >= ntiles)
        return;
    const unsigned t1 = min(t0 + CELL_BOX_RUN, ntiles);
    float lo[8], hi[8];
    auto reset = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            lo[j] = INFINITY;
            hi[j] = -INFINITY;
        }
    };
    auto fold = [&](unsigned cell) {
#pragma unroll
        for (int off = 16; off > 0; off >>= 1)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                lo[j] = fminf(lo[j], __shfl_xor(lo[j], off, KNN_WAVE));
                hi[j] = fmaxf(hi[j], __shfl_xor(hi[j], off, KNN_WAVE));
            }
        if ((lane & 31) == 0) {
            unsigned *__restrict__ b = box + (size_t)cell * 32;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (lo[j] < INFINITY)
                    guarded_atomic_min(&b[8 * half + j], f2ord(lo[j]));
                if (hi[j] > -INFINITY)
                    guarded_atomic_max(&b[16 + 8 * half + j], f2ord(hi[j]));
            }
        }
    };
    reset();
    unsigned cur = tile_cell[t0];
    for (unsigned t = t0; t < t1; ++t) {
        const unsigned cell = tile_cell[t];   // wave-uniform
        if (cell != cur) {
            fold(cur);
            reset();
            cur = cell;
        }
        const size_t pos = (size_t)t * 32 + (lane & 31);
        const unsigned row = perm[pos];
        if (row != 0xFFFFFFFFu && norms[pos] < INFINITY) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (8 * half + j < k) {
                    const float x = R[(size_t)row * k + 8 * half + j];
                    lo[j] = fminf(lo[j], x);
                    hi[j] = fmaxf(hi[j], x);
                }
        }
    }
    fold(cur);
}

// frame[c] = { centre[16], scale, 2^e, 0 (bmax), 0 (nmax) }: the centre is the middle of the cell's box, e the largest
// exponent <= 8 for which the box, scaled, stays inside [-1, 1]
__global__ __launch_bounds__(256) void knn_cells_frame_kernel(const unsigned *__restrict__ box, unsigned ncells, int k, float sigma,
                                                              float *__restrict__ frame)
{
    const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncells)
        return;
    float *__restrict__ f = frame + (size_t)c * KNN_CELL_FRAME_WORDS;
    float hw = 0.0f;
    for (int d = 0; d < 16; ++d) {
        float ctr = 0.0f;
        if (d < k && box[(size_t)c * 32 + d] <= box[(size_t)c * 32 + 16 + d]) {   // (min <= max: the cell has a row)
            const float lo = ord2f(box[(size_t)c * 32 + d]), hi = ord2f(box[(size_t)c * 32 + 16 + d]);
            ctr = 0.5f * lo + 0.5f * hi;
            hw = fmaxf(hw, fmaxf(hi - ctr, ctr - lo));
        }
        f[d] = ctr;
    }
    float ratio = 1.0f;
    for (int e = 0; e < 8 && hw * sigma * (2.0f * ratio) <= 0.999f; ++e)
        ratio *= 2.0f;
    f[16] = sigma * ratio;
    f[17] = ratio;
    f[18] = 0.0f;
    f[19] = 0.0f;
}

// The fragments, norms and norm halves of every tile again, in its cell's frame; frame[c][18] / [19] = the cell's largest
// |coordinate| and norm.  Same arithmetic as the placement (fp32 subtract, exact power-of-two scale, round to nearest even,
// exact products summed in fp32), same rule for rows outside the shard's robust box (they stay out: zero fragment, +INF norm).
__global__ __launch_bounds__(256) void knn_cells_recentre_kernel(const float *__restrict__ R, int k, const unsigned *__restrict__ perm,
                                                                 unsigned ntiles, const unsigned *__restrict__ tile_cell,
                                                                 float *__restrict__ frame, h8 *__restrict__ frag,
                                                                 float *__restrict__ norms, unsigned *__restrict__ norms2)
{
    const unsigned t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63, half = lane >> 5;
    if (t >= ntiles)
        return;
    const size_t pos = (size_t)t * 32 + (lane & 31);
    const unsigned row = perm[pos];
    const bool real = row != 0xFFFFFFFFu && norms[pos] < INFINITY;
    float *__restrict__ f = frame + (size_t)tile_cell[t] * KNN_CELL_FRAME_WORDS;
    const float scale = f[16];
    h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    float vmax = 0.0f, part = 0.0f;
    if (real) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = 8 * half + j;
            const float sc = d < k ? (R[(size_t)row * k + d] - f[d]) * scale : 0.0f;
            const _Float16 hval = (_Float16)sc;
            const float back = (float)hval;
            vmax = fmaxf(vmax, fabsf(back));
            part = part + back * back;
            v[j] = hval;
        }
    }
    const float other = __shfl_xor(part, 32, KNN_WAVE);
    const float nrm = half == 0 ? part + other : other + part;   // (dimensions 0..7) + (dimensions 8..15) on both lanes of a row
    frag[(size_t)t * 64 + lane] = v;
    if (half == 0) {
        norms[pos] = real ? nrm : INFINITY;
        norms2[pos] = pack_norm22(real ? nrm : INFINITY);
    }
    vmax = wave_max_f(vmax);
    const float nmaxv = wave_max_f(real ? nrm : 0.0f);
    if (lane == 0) {
        guarded_atomic_max((unsigned *)&f[18], __float_as_uint(vmax));
        guarded_atomic_max((unsigned *)&f[19], __float_as_uint(nmaxv));
    }
}

// ------------------------------------------------------------------------------------------
// Host side.
// ------------------------------------------------------------------------------------------
void knn_cells_free(CellIndex *&c)
{
    if (!c)
        return;
    (void)KNN_DEV_FREE(c->bounds);
    (void)KNN_DEV_FREE(c->tile_start);
    (void)KNN_DEV_FREE(c->perm);
    (void)KNN_DEV_FREE(c->items);
    (void)KNN_DEV_FREE(c->tmp_rows);
    (void)KNN_DEV_FREE(c->tmp_meta);
    (void)KNN_DEV_FREE(c->bucket_start);
    (void)KNN_DEV_FREE(c->cell_frame);
    (void)KNN_DEV_FREE(c->tile_cell);
    delete c;
    c = nullptr;
}

std::atomic<int> g_knn_cells_centre{0};                 // option `cells_centre`
std::atomic<long long> g_knn_cells_centred_builds{0};   // read-only option `cells_centred_builds`

// Moves a finished cell-sorted layout (k <= 16, no shard geometry) into per-cell frames — see the kernels.  Enqueues on `s`;
// the caller's next synchronisation covers it.
hipError_t knn_cells_recentre(FilterState &st, const float *r, hipStream_t s)
{
    if (!st.cells || st.kt != 1 || st.cells->geom || st.cells->centred)
        return hipSuccess;
    CellIndex &c = *st.cells;
    const unsigned ntiles = (unsigned)st.ntiles;
    unsigned *box = nullptr;
    FTRY(KNN_DEV_ALLOC((void **)&c.cell_frame, (size_t)c.ncells * KNN_CELL_FRAME_WORDS * sizeof(float)));
    FTRY(KNN_DEV_ALLOC((void **)&c.tile_cell, (size_t)std::max(1u, ntiles) * sizeof(unsigned)));
    FTRY(KNN_DEV_ALLOC((void **)&box, (size_t)c.ncells * 32 * sizeof(unsigned)));
    hipError_t e = hipMemsetAsync(box, 0xFF, (size_t)c.ncells * 32 * sizeof(unsigned), s);   // min = ~0; max: cleared below
    if (e == hipSuccess)
        e = hipMemset2DAsync(box + 16, 32 * sizeof(unsigned), 0, 16 * sizeof(unsigned), c.ncells, s);
    if (e == hipSuccess && ntiles != 0u) {
        hipLaunchKernelGGL(knn_cells_tile_cell_kernel, dim3((c.ncells * 32u + 255u) / 256u), dim3(256), 0, s, c.tile_start, c.ncells, c.tile_cell);
        hipLaunchKernelGGL(knn_cells_box_kernel, dim3(((ntiles + CELL_BOX_RUN - 1u) / CELL_BOX_RUN + 3u) / 4u), dim3(256), 0, s, r, st.k, c.perm, st.ref_norms,
                           ntiles, c.tile_cell, box);
        hipLaunchKernelGGL(knn_cells_frame_kernel, dim3((c.ncells + 255u) / 256u), dim3(256), 0, s, box, c.ncells, st.k, st.sigma, c.cell_frame);
        hipLaunchKernelGGL(knn_cells_recentre_kernel, dim3((ntiles + 3u) / 4u), dim3(256), 0, s, r, st.k, c.perm, ntiles, c.tile_cell,
                           c.cell_frame, (h8 *)st.ref_frags, st.ref_norms, st.ref_norms2);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);   // (box is scratch)
    (void)KNN_DEV_FREE(box);
    FTRY(e);
    c.centred = true;
    ++g_knn_cells_centred_builds;
    return hipSuccess;
}

// What the builds call once a cell-sorted layout stands: per-cell frames when option `cells_centre` asks for them (1), or
// (0) when the build's host sample looks clustered.  samp may be null (no sample at hand: only on request).
hipError_t knn_cells_maybe_recentre(FilterState &st, const float *r, const float *samp, long long samples, hipStream_t s)
{
    if (!st.usable || !st.cells || st.kt != 1 || st.cells->geom)
        return hipSuccess;
    const int opt = g_knn_cells_centre;
    if (opt == 2 || (opt == 0 && !(samp && knn_cells_sample_is_clustered(samp, samples, st.k, st.sigma))))
        return hipSuccess;
    return knn_cells_recentre(st, r, s);
}

// Does the build's host sample look clustered?  Median, over 64 sample rows, of the distance (largest coordinate difference)
// to the nearest of up to 256 other sample rows, in units of the frame's box (scaled: the box is [-1, 1]): uniform or gaussian
// data in 16 dimensions ~0.5-1, 1000 blobs of 0.05 of the box likewise (the sample's neighbours are in other blobs), 64
// clusters of 10^-3 of the box 0.007.  Below 1/16 the per-cell frames pay for their pass over the rows.  (0.05 ms of host
// arithmetic in every build of a cell-sorted layout.)
bool knn_cells_sample_is_clustered(const float *samp, long long samples, int k, float sigma)
{
    const long long cand = std::min<long long>(samples, 256), step = cand > 0 ? samples / cand : 1, probes = std::min<long long>(cand, 64);
    if (cand < 32)
        return false;
    std::vector<float> nn((size_t)probes, INFINITY);
    for (long long i = 0; i < probes; ++i) {
        const float *a = samp + (size_t)(i * (cand / probes) * step) * k;
        for (long long j = 0; j < cand; ++j) {
            const float *b = samp + (size_t)(j * step) * k;
            if (a == b)
                continue;
            float dmax = 0.0f;
            for (int d = 0; d < k; ++d)
                dmax = std::max(dmax, fabsf(a[d] - b[d]));
            if (!(dmax == dmax))
                return false;
            nn[(size_t)i] = std::min(nn[(size_t)i], dmax);
        }
    }
    std::nth_element(nn.begin(), nn.begin() + probes / 2, nn.end());
    return nn[(size_t)(probes / 2)] * sigma < 1.0f / 16.0f;
}

// The fast build in stages (an ingest runs the scatter chunk by chunk under the copy): rows [row0, row1) of the shard, at `r`
// = the shard's first row on the device.
hipError_t knn_cells_fast_scatter(CellIndex &c, int k, const float *r, long long row0, long long row1, hipStream_t s)
{
    if (row1 <= row0)
        return hipSuccess;
    const CellGeom g = cell_geom_of(c, k);
    const long long rows = row1 - row0;
    hipLaunchKernelGGL(knn_cells_bucket_scatter_fixed_kernel, dim3((unsigned)((rows + CELL_BUILD_ROWS - 1) / CELL_BUILD_ROWS)), dim3(256), 0,
                       s, r + (size_t)row0 * k, rows, row0, g, c.bounds, c.lbits - 8, c.bucket_cap, c.bucket_fill, c.tmp_rows, c.tmp_meta,
                       c.build_res + 3);
    return hipGetLastError();
}

// rows per cell, tile ranges + items (device prefix), fill counters zeroed for the placement
hipError_t knn_cells_fast_finish(CellIndex &c, unsigned *counts, hipStream_t s)
{
    hipLaunchKernelGGL(knn_cells_bucket_cellcount_kernel, dim3(CELL_BUCKETS * CELL_PLACE_PARTS), dim3(256), 0, s, c.tmp_meta, c.bucket_start,
                       c.lbits - 8, counts, c.bucket_fill);
    hipLaunchKernelGGL(knn_cells_prefix_kernel, dim3((c.ncells + 1023u) / 1024u), dim3(1024), 0, s, counts, c.ncells, c.tile_start, c.items,
                       c.build_res);
    FTRY(hipGetLastError());
    return hipMemsetAsync(counts, 0, (size_t)c.ncells * sizeof(unsigned), s);   // the counts become the placement's fill counters
}

// Sorts the shard into cells (see the head of this file).  *out stays null when the shard is too small, the
// dimension does not suit, or the cuts leave the cells badly unbalanced.  samp: the strided host sample
// of the build (samples x k).  Synchronous.
hipError_t knn_cells_build(CellIndex **out, int k, long long n, const float *r, const std::vector<float> &samp,
                           long long samples, hipStream_t s, long long *ntiles_out, unsigned **code_out,
                           unsigned **fill_out, bool one_pass, const ShardGeom *geom, int rank, unsigned *bad_rows_out, bool fast,
                           bool defer_scatter)
{
    *out = nullptr;
    *code_out = nullptr;
    *fill_out = nullptr;
    if (bad_rows_out)
        *bad_rows_out = 0u;
    // (16 < k <= 32, round 5: the cells cut the first 16 dimensions — a lower bound over some dimensions is one over all —
    // and the layout has two K-steps per tile; one-pass placement, no shard geometry)
    if (k > 32 || (k > 16 && geom) || n > 0x7FFFFFFFll || (!geom && (n < (1ll << 17) || samples < 64)))
        return hipSuccess;
    const int kc = std::min(k, 16);   // dimensions the cells may cut
    if (k > 16) {
        one_pass = true;
        fast = false;
    }
    CellIndex *c = new CellIndex();
    std::vector<float> bounds((size_t)16 * (CELL_MAX_BINS - 1), INFINITY);
    if (geom) {
        // cell-range shard: the global grid's shape and cuts, this rank's range of its codes
        c->bits = geom->bits;
        c->sa = geom->sa;
        memcpy(c->nb, geom->nb, 16);
        memcpy(c->shift, geom->shift, 16);
        c->cell_base = geom->first_cell(rank);
        c->ncells = geom->cells_of(rank);
        c->geom = geom;
        memcpy(bounds.data(), geom->bounds, sizeof geom->bounds);
        if (c->ncells < 512u || c->ncells % 64u != 0u || c->ncells > 65536u) {   // (knn_geom_from_sample sizes the ranges so)
            delete c;
            return hipSuccess;
        }
    } else {
        const int bits = std::min(cell_bits_for_rows(n), std::min(16, 4 * kc));
        if (bits < 9) {
            delete c;
            return hipSuccess;
        }
        c->bits = bits;
        c->ncells = 1u << bits;
        cell_grid_shape(kc, bits, c->nb, c->shift, &c->sa);
        if (c->sa < 6) {   // a wave of the match pass covers 64 consecutive low-table entries
            delete c;
            return hipSuccess;
        }
        cell_quantile_cuts(k, c->nb, samp.data(), samples, bounds.data());
    }
    // list capacity per cell and batch: 384 queries per cell at >= 2^15 cells (48 MiB of lists per slot at 2^16 cells; uniform
    // data in 16 dimensions keeps 25 of 1024 on average, 53 at most), every query of a batch at <= 2^13 cells
    // (round 5: 384 entries at >= 2^15 cells, was 128 — heavy-tailed rows with gaussian queries, n 2^24: lists of 121 .. 213
    // queries, nearly every cell `dense` (scored against all 1024), 1.16 ms per step; 0.36 with room for the lists.  The match
    // kernel assembles the first 128 entries of a list in LDS as before and writes the rest straight to memory)
    c->cap = std::min(1024u, std::max(384u, (1u << 23) / c->ncells));
    if (k > 16)   // (more dimensions, fewer cells ruled out: lists of ~100-200 of 1024 queries at k = 20 — 128 entries made most cells dense)
        c->cap = std::min(1024u, std::max(k > 20 ? 640u : 384u, (1u << 24) / c->ncells));
    // (k = 24, n 2^24: lists of 384 left most cells dense, 1.19 ms per step; 640 or 1024 entries 0.75 — the full scan takes 0.97)
    const CellGeom g = cell_geom_of(*c, k);
    int lbits = 0;   // bits of a LOCAL cell number (= bits without a shard geometry)
    while ((1u << lbits) < c->ncells)
        ++lbits;

    // ---- the fast build (round 5): no counting pass over the rows, no host round trip (see the kernels).  Taken for whole
    // indexes (no shard geometry: their rows' codes are local = global and fit 16 bits) whose scratch fits; everything it
    // launches is asynchronous — the caller reads c->build_res behind the placement and, if a bucket outgrew its fixed room
    // (data the quantile cuts do not spread evenly over the 256 buckets), builds again with fast = false.
    if (defer_scatter && !(fast && !geom && !one_pass && (size_t)n * 96 <= ((size_t)2 << 30) && c->ncells >= 512u)) {
        delete c;
        return hipSuccess;
    }
    if (fast && !geom && !one_pass && (size_t)n * 96 <= ((size_t)2 << 30) && c->ncells >= 512u) {
        // room per bucket: an even spread + 1/2.  The cuts are medians of a 1024-row sample: each is off by ~1.6 % of the rows
        // (1 sigma), a bucket is the product of 8 such halves — 1 sigma 9 %, the fullest of 256 buckets ~27 % over the mean on
        // uniform data (measured: + 1/8 overflowed at C3)
        const unsigned cap_rows = (unsigned)((n / CELL_BUCKETS) + (n / CELL_BUCKETS) / 2 + CELL_BUILD_ROWS);
        const size_t recs = (size_t)cap_rows * CELL_BUCKETS;
        const long long tiles_ub = n / 32 + (long long)c->ncells;                         // every cell wastes less than a tile
        const size_t items_ub = (size_t)c->ncells + (size_t)(tiles_ub / KNN_CELL_ITEM_TILES) + 1u;
        unsigned *counts_f = nullptr;
        hipError_t a = KNN_DEV_ALLOC((void **)&c->bounds, (bounds.size() + 1) * sizeof(float));
        if (a == hipSuccess)
            a = KNN_DEV_ALLOC((void **)&c->tile_start, ((size_t)c->ncells + 1) * sizeof(unsigned));
        if (a == hipSuccess)
            a = KNN_DEV_ALLOC((void **)&counts_f, (size_t)c->ncells * sizeof(unsigned));
        if (a == hipSuccess)
            a = KNN_DEV_ALLOC((void **)&c->tmp_rows, recs * 16 * sizeof(float));
        if (a == hipSuccess)
            a = KNN_DEV_ALLOC((void **)&c->tmp_meta, recs * sizeof(u64));
        if (a == hipSuccess)   // [257] bucket starts | [256] fills | [4] results
            a = KNN_DEV_ALLOC((void **)&c->bucket_start, (CELL_BUCKETS + 1 + CELL_BUCKETS + 4) * sizeof(unsigned));
        if (a == hipSuccess)
            a = KNN_DEV_ALLOC((void **)&c->items, items_ub * sizeof(u64));
        if (a == hipSuccess)
            a = KNN_DEV_ALLOC((void **)&c->perm, (size_t)tiles_ub * 32 * sizeof(unsigned));
        if (a == hipSuccess) {
            c->bucket_fill = c->bucket_start + CELL_BUCKETS + 1;
            c->build_res = c->bucket_fill + CELL_BUCKETS;
            static_assert(sizeof c->h_bounds == 16 * (CELL_MAX_BINS - 1) * sizeof(float), "cuts");
            for (int b = 0; b <= CELL_BUCKETS; ++b)
                c->h_bucket_start[b] = (unsigned)b * cap_rows;
            memcpy(c->h_bounds, bounds.data(), sizeof c->h_bounds);
            a = hipMemcpyAsync(c->bounds, c->h_bounds, sizeof c->h_bounds, hipMemcpyHostToDevice, s);
            if (a == hipSuccess)
                a = hipMemcpyAsync(c->bucket_start, c->h_bucket_start, sizeof c->h_bucket_start, hipMemcpyHostToDevice, s);
            if (a == hipSuccess)
                a = hipMemsetAsync(c->bucket_fill, 0, (CELL_BUCKETS + 4) * sizeof(unsigned), s);
            if (a == hipSuccess)
                a = hipMemsetAsync(counts_f, 0, (size_t)c->ncells * sizeof(unsigned), s);
            c->lbits = lbits;
            c->bucket_cap = cap_rows;
            if (a == hipSuccess && !defer_scatter) {   // (an ingest scatters chunk by chunk as the rows land, then finishes)
                a = knn_cells_fast_scatter(*c, k, r, 0, n, s);
                if (a == hipSuccess)
                    a = knn_cells_fast_finish(*c, counts_f, s);
            }
        }
        if (a == hipSuccess) {
            c->nitems = 0u;   // (the caller fills nitems / max_cell_rows in from build_res)
            *code_out = nullptr;
            *fill_out = counts_f;
            *ntiles_out = tiles_ub;
            *out = c;
            return hipSuccess;
        }
        // no room (or a launch failed): the counted build below starts from scratch
        (void)hipGetLastError();
        (void)KNN_DEV_FREE(counts_f);
        knn_cells_free(c);
        if (defer_scatter)   // (an ingest has no rows on the device yet: nothing to count — the caller copies, then builds)
            return hipSuccess;
        c = new CellIndex();
        c->bits = lbits;
        c->ncells = 1u << lbits;
        cell_grid_shape(kc, lbits, c->nb, c->shift, &c->sa);
        c->cap = std::min(1024u, std::max(384u, (1u << 23) / c->ncells));
    }
    unsigned *code = nullptr, *counts = nullptr;
    std::vector<unsigned> hcounts((size_t)c->ncells), hstart((size_t)c->ncells + 1);
    std::vector<u64> hitems;
    // (one word behind the cuts counts rows outside the index's cell range)
    hipError_t e = KNN_DEV_ALLOC((void **)&c->bounds, (bounds.size() + 1) * sizeof(float));
    unsigned *bad_dev = (unsigned *)(c->bounds + bounds.size());
    if (e == hipSuccess)
        e = hipMemsetAsync(bad_dev, 0, sizeof(unsigned), s);
    if (e == hipSuccess)
        e = KNN_DEV_ALLOC((void **)&c->tile_start, hstart.size() * sizeof(unsigned));
    if (e == hipSuccess)
        e = KNN_DEV_ALLOC((void **)&code, (size_t)n * sizeof(unsigned));
    if (e == hipSuccess)
        e = KNN_DEV_ALLOC((void **)&counts, hcounts.size() * sizeof(unsigned));
    if (e == hipSuccess)
        e = hipMemcpyAsync(c->bounds, bounds.data(), bounds.size() * sizeof(float), hipMemcpyHostToDevice, s);
    if (e == hipSuccess)
        e = hipMemsetAsync(counts, 0, hcounts.size() * sizeof(unsigned), s);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    // Two-pass build (see the kernels): needs n x 72 bytes of scratch; without it (or with option `cells_build` = 1, for A/B
    // timing and tests) the one-pass placement serves.
    const bool one_pass_env = one_pass;
    const int bshift = lbits - 8;   // >= 512 cells: a bucket = 2^bshift consecutive (local) cells
    const unsigned bblocks = (unsigned)((n + CELL_BUILD_ROWS - 1) / CELL_BUILD_ROWS);
    unsigned *bucket_counts = nullptr, *bucket_fill = nullptr;
    // (shards of up to 2^25 rows: beyond, the scratch is gigabytes that the buffer pool does not keep between builds, and one
    // hipMalloc / hipFree pair of that size in eight took 2.2 SECONDS on a 2^27-row shard — tools/build_repeat.py: 30 20 19 19
    // 19 19 2208 21 ms — where the one-pass placement's 33 ms are steady)
    bool two_pass = e == hipSuccess && !one_pass_env && (size_t)n * 64 <= ((size_t)2 << 30);
    if (two_pass) {
        hipError_t a = KNN_DEV_ALLOC((void **)&c->tmp_rows, (size_t)n * 16 * sizeof(float));
        if (a == hipSuccess)
            a = KNN_DEV_ALLOC((void **)&c->tmp_meta, (size_t)n * sizeof(u64));
        if (a == hipSuccess)
            a = KNN_DEV_ALLOC((void **)&c->bucket_start, (CELL_BUCKETS + 1) * sizeof(unsigned));
        if (a == hipSuccess)
            a = KNN_DEV_ALLOC((void **)&bucket_counts, 2 * CELL_BUCKETS * sizeof(unsigned));
        if (a != hipSuccess) {   // no room for the scratch: one pass
            (void)hipGetLastError();
            (void)KNN_DEV_FREE(c->tmp_rows);
            (void)KNN_DEV_FREE(c->tmp_meta);
            (void)KNN_DEV_FREE(c->bucket_start);
            (void)KNN_DEV_FREE(bucket_counts);
            c->tmp_rows = nullptr;
            c->tmp_meta = nullptr;
            c->bucket_start = nullptr;
            bucket_counts = nullptr;
            two_pass = false;
        } else
            bucket_fill = bucket_counts + CELL_BUCKETS;
    }
    if (two_pass) {
        unsigned hb[CELL_BUCKETS], hbs[CELL_BUCKETS + 1];
        e = hipMemsetAsync(bucket_counts, 0, 2 * CELL_BUCKETS * sizeof(unsigned), s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(knn_cells_bucket_count_kernel, dim3(bblocks), dim3(256), 0, s, r, n, g, c->bounds, bshift, code,
                               bucket_counts, bad_dev);
            e = hipGetLastError();
        }
        if (e == hipSuccess)
            e = hipMemcpyAsync(hb, bucket_counts, sizeof hb, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess)
            e = hipStreamSynchronize(s);
        if (e == hipSuccess) {
            unsigned acc = 0u;
            for (int b = 0; b < CELL_BUCKETS; ++b) {
                hbs[b] = acc;
                acc += hb[b];
            }
            hbs[CELL_BUCKETS] = acc;
            e = hipMemcpyAsync(c->bucket_start, hbs, sizeof hbs, hipMemcpyHostToDevice, s);
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(knn_cells_bucket_scatter_kernel, dim3(bblocks), dim3(256), 0, s, r, n, k, bshift, code, c->bucket_start,
                               bucket_fill, c->tmp_rows, c->tmp_meta);
            hipLaunchKernelGGL(knn_cells_bucket_cellcount_kernel, dim3(CELL_BUCKETS * CELL_PLACE_PARTS), dim3(256), 0, s, c->tmp_meta,
                               c->bucket_start, bshift, counts, (const unsigned *)nullptr);
            e = hipGetLastError();
        }
        if (e == hipSuccess)
            e = hipMemcpyAsync(hcounts.data(), counts, hcounts.size() * sizeof(unsigned), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess)
            e = hipStreamSynchronize(s);   // (hbs has been copied; also keeps `bounds` alive until its copy is done)
        (void)KNN_DEV_FREE(bucket_counts);
    } else {
        if (e == hipSuccess) {
            hipLaunchKernelGGL(knn_cells_code_kernel, dim3(blocks), dim3(256), 0, s, r, n, g, c->bounds, code, counts, bad_dev);
            e = hipGetLastError();
        }
        if (e == hipSuccess)
            e = hipMemcpyAsync(hcounts.data(), counts, hcounts.size() * sizeof(unsigned), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess)
            e = hipStreamSynchronize(s);   // (also keeps `bounds` alive until its copy is done)
    }
    bool keep = e == hipSuccess;
    if (keep && geom) {   // rows outside this rank's cell range: the caller partitioned with another geometry
        unsigned hbad = 0u;
        e = hipMemcpy(&hbad, bad_dev, sizeof hbad, hipMemcpyDeviceToHost);
        if (e == hipSuccess && hbad != 0u) {
            if (bad_rows_out)
                *bad_rows_out = hbad;
            keep = false;
        }
        keep = keep && e == hipSuccess;
    }
    c->lbits = lbits;
    long long tiles = 0;
    if (keep) {
        unsigned biggest = 0u;
        for (unsigned i = 0; i < c->ncells; ++i) {
            hstart[i] = (unsigned)tiles;
            tiles += (hcounts[i] + 31u) / 32u;
            biggest = std::max(biggest, hcounts[i]);
        }
        hstart[c->ncells] = (unsigned)tiles;
        c->max_cell_rows = biggest;
        // (round 2 gave up here when the largest cell held more than 16x the average — clustered or low-rank data the
        // per-dimension quantiles do not spread.  The scan now takes ITEMS, runs of at most KNN_CELL_ITEM_TILES tiles of one
        // cell, so a fat cell is many waves' work instead of one's, and cells without rows cost nothing)
        for (unsigned i = 0; i < c->ncells; ++i)
            for (unsigned t = hstart[i]; t < hstart[i + 1]; t += KNN_CELL_ITEM_TILES)
                hitems.push_back(((u64)i << 48) | ((u64)std::min((unsigned)KNN_CELL_ITEM_TILES, hstart[i + 1] - t) << 40) | (u64)t);
        c->nitems = (unsigned)hitems.size();
        keep = c->nitems != 0u;
    }
    if (keep) {
        // the rows are placed (and turned into fragments) by knn_cells_scatter_frag_kernel once the layout buffers
        // exist: `code` and the zeroed fill counters go back to the caller
        e = hipMemcpyAsync(c->tile_start, hstart.data(), hstart.size() * sizeof(unsigned), hipMemcpyHostToDevice, s);
        if (e == hipSuccess)
            e = hipMemsetAsync(counts, 0, hcounts.size() * sizeof(unsigned), s);
        if (e == hipSuccess)
            e = KNN_DEV_ALLOC((void **)&c->items, hitems.size() * sizeof(u64));
        if (e == hipSuccess)
            e = hipMemcpyAsync(c->items, hitems.data(), hitems.size() * sizeof(u64), hipMemcpyHostToDevice, s);
        if (e == hipSuccess)
            e = KNN_DEV_ALLOC((void **)&c->perm, (size_t)tiles * 32 * sizeof(unsigned));   // (filled by the placement + padding kernels)
        if (e == hipSuccess)
            e = hipStreamSynchronize(s);   // hstart / hitems are about to go out of scope
    }
    if (e == hipErrorOutOfMemory) {   // no room for the sort: the plain layout still works
        (void)hipGetLastError();
        e = hipSuccess;
        keep = false;
    }
    if (e != hipSuccess || !keep) {
        (void)KNN_DEV_FREE(code);
        (void)KNN_DEV_FREE(counts);
        knn_cells_free(c);
        return e;
    }
    *code_out = code;
    *fill_out = counts;
    *ntiles_out = tiles;
    *out = c;
    return hipSuccess;
}

// Every row to its cell (knn_cells_place_kernel / knn_cells_scatter_frag_kernel), then the padding of the cells' last tiles.
// code / fill: what knn_cells_build handed back; out: the 4 words of knn_frag_kernel's statistics.
hipError_t knn_cells_place_rows(FilterState &st, const float *r, const unsigned *code, unsigned *fill, unsigned *out,
                                unsigned ocap, hipStream_t s)
{
    if (st.cells->tmp_rows)
        hipLaunchKernelGGL(knn_cells_place_kernel, dim3(CELL_BUCKETS * CELL_PLACE_PARTS), dim3(256), 0, s, st.cells->tmp_rows,
                           st.cells->tmp_meta, st.cells->bucket_start, st.k, st.cells->lbits - 8, st.cells->tile_start, fill, st.center, st.sigma,
                           (h8 *)st.ref_frags, st.ref_norms, st.ref_norms2, st.cells->perm, out, st.outliers, ocap,
                           (const unsigned *)st.cells->bucket_fill);
    else if (st.kt == 2)
        hipLaunchKernelGGL(knn_cells_scatter_frag2_kernel, dim3((unsigned)((st.n + 255) / 256)), dim3(256), 0, s, r, st.n, st.k, code,
                           st.cells->tile_start, fill, st.center, st.sigma, (h8 *)st.ref_frags, st.ref_norms, st.ref_norms2,
                           st.cells->perm, out, st.outliers, ocap);
    else
        hipLaunchKernelGGL(knn_cells_scatter_frag_kernel, dim3((unsigned)((st.n + 255) / 256)), dim3(256), 0, s, r, st.n, st.k, code,
                           st.cells->tile_start, fill, st.center, st.sigma, (h8 *)st.ref_frags, st.ref_norms, st.ref_norms2,
                           st.cells->perm, out, st.outliers, ocap);
    FTRY(hipGetLastError());
    hipLaunchKernelGGL(knn_cells_pad_kernel, dim3((st.cells->ncells + 7u) / 8u), dim3(256), 0, s, st.cells->tile_start, fill,
                       st.cells->ncells, (h8 *)st.ref_frags, st.ref_norms, st.ref_norms2, st.cells->perm, st.kt,
                       st.kt == 2 && st.k <= KNN_NIF_MAX_K ? 1 : 0);
    return hipGetLastError();
}

void knn_cells_workspace_free(FilterWorkspace &w)
{
    (void)KNN_DEV_FREE(w.cell_counts);
    (void)KNN_DEV_FREE(w.cell_lists);
    (void)KNN_DEV_FREE(w.dup);
    (void)KNN_DEV_FREE(w.lo_tab);
    (void)KNN_DEV_FREE(w.hi_tab);
    w.cell_counts = nullptr;
    w.cell_lists = nullptr;
    w.dup = w.lo_tab = w.hi_tab = nullptr;
    w.cell_m_cap = 0;
}

// ---- per batch ---------------------------------------------------------------------------------

static hipError_t ensure_cells_workspace(FilterState &st, FilterWorkspace &w, int m)
{
    const CellIndex &c = *st.cells;
    if (!w.cell_counts)
        FTRY(KNN_DEV_ALLOC((void **)&w.cell_counts, (size_t)c.ncells * sizeof(unsigned)));
    if (!w.cell_lists)
        FTRY(KNN_DEV_ALLOC((void **)&w.cell_lists, (size_t)c.ncells * c.cap * sizeof(unsigned short)));
    const int m_padded = (m + 31) / 32 * 32;
    if (m_padded > w.cell_m_cap) {
        (void)KNN_DEV_FREE(w.dup);
        (void)KNN_DEV_FREE(w.lo_tab);
        (void)KNN_DEV_FREE(w.hi_tab);
        w.dup = w.lo_tab = w.hi_tab = nullptr;
        w.cell_m_cap = 0;
        FTRY(KNN_DEV_ALLOC((void **)&w.dup, (size_t)m_padded * sizeof(float)));
        FTRY(KNN_DEV_ALLOC((void **)&w.lo_tab, (size_t)m_padded * ((size_t)1 << c.sa) * sizeof(float)));
        FTRY(KNN_DEV_ALLOC((void **)&w.hi_tab, (size_t)m_padded * (size_t)((c.ncells + (1u << c.sa) - 1u) >> c.sa) * sizeof(float)));
        w.cell_m_cap = m_padded;
    }
    return hipSuccess;
}

// Every size the scan launch and the re-rank behind it index with, in one place (tests/test_cells_logic.py checks the
// arithmetic on the CPU through knn_debug_scan_plan):
//   blocks   scan grid: blocks_per_cu per CU, fewer when the index has fewer items than that many waves
//   nlists   = blocks x CELL_SCAN_WAVES record lists, one per wave: counts[nlists] (the workspace holds kMaxLists = 2^16)
//   slice    records [w x slice, (w + 1) x slice) belong to wave w
//   ovf      records [ovf_base, ovf_base + ovf_cap) are the area all waves share; nlists x slice <= ovf_base
//   lds      dynamic LDS of the scan: m_padded x (32 B operand + 4 B threshold) + one norm window per wave
CellScanPlan knn_cells_scan_plan(int num_cu, int blocks_per_cu, unsigned nitems, unsigned rec_cap, int m_padded, bool self_lists, int kt,
                                 bool centred)
{
    CellScanPlan p;
    const unsigned sw = kt == 1 && !centred ? CELL_SCAN_WAVES : CELL_SCAN_WAVES_KT2;
    p.waves = sw;
    p.blocks = (unsigned)num_cu * (unsigned)(kt == 1 && !centred ? blocks_per_cu : 1);
    if (p.blocks * sw > nitems)
        p.blocks = std::max(1u, nitems / sw);
    p.nlists = p.blocks * sw;
    // the tail of the record buffer is shared by all waves: what a wave's own slice cannot hold goes there.  2^16 records:
    // room for a batch whose queries crowd into a few cells (1024 copies of one query leave ~2000 records there), and small
    // enough that a batch the fp16 scores cannot separate at all (a cluster tighter than the fp16 step: millions of
    // candidates, one atomic on ONE word per overflowing step) over-fills it — and stops scanning — within microseconds
    p.ovf_cap = std::min(rec_cap / 4u, 1u << 16);
    p.ovf_base = rec_cap - p.ovf_cap;
    p.slice = p.ovf_base / p.nlists;
    // (16 < k <= 32: 64-byte B operands, 74 KiB for 1024 queries: with windows of nine 2-KiB tiles in registers (126 VGPRs) a CU
    // holds ONE block, so that block has sixteen waves — four per SIMD; the launch raises the kernel's dynamic-LDS limit above
    // the default 64 KiB.  Round 5, one box, ms per step at k 17 / 18 / 20, n 2^24: four tiles per pass, 12 waves, 96
    // registers 0.282 / 0.315 / 0.428; five 0.269 / 0.303 / 0.399; nine, 12 waves (3 per SIMD) 0.237 / 0.242 / 0.352)
    p.lds_bytes = (size_t)m_padded * ((centred ? 64 : 32 * kt) + 4) + (size_t)sw * CELL_TILES_PER_PASS * 8 * sizeof(f4v);   // (per-cell frames: the queries' fp32 rows)
    if (self_lists)   // the self-listing scan: the batch's Dup values + one list room per wave
        p.lds_bytes += (size_t)m_padded * sizeof(float) + (size_t)sw * CELL_SELF_CAP * sizeof(unsigned short);
    else if (centred)   // per-cell frames: the batch's Dup values
        p.lds_bytes += (size_t)m_padded * sizeof(float);
    return p;
}

// Who lists the queries of a cell (option `cells_lists` 0 = this policy, 1 = the match launch, 2 = the scan's own waves).
// Measured (round 5, one MI355X, k 16, m 1024, uniform; profiles/r05_self_lists.txt): the self-listing scan saves the match
// launch and a dependent round trip per item, and pays with 16 steps of table arithmetic per item inside the hot kernel.  One
// batch at a time on a rank of 8 of C3 (2^13 cells): 0.0469 -> 0.0439 ms; with four batches in flight the match launch hides
// behind the other batches and the longer scan does not: 0.0252 -> 0.0288 per step.  2^14 cells: 0.0599 -> 0.0616 / 0.0397 ->
// 0.0433; C3: 0.1430 -> 0.1448 / 0.1176 -> 0.1280.
bool knn_cells_lists_policy(unsigned ncells, bool several_slots)
{
    return !several_slots && ncells <= 8192u;
}

// One batch of <= KNN_CELL_BATCH queries, the whole chain: prep -> match -> scan (its waves re-rank their own records) ->
// [rows outside the robust box, exactly] -> the exact scan of the shard, gated on FALLBACK -> the tail kernel (gated: records
// in the shared area, the listed pairs exactly when that area is over-full, the finalisation when the scan could not do it).
// Four launches on clean data at k = 16 (the exact scan of the shard is a branch of the tail kernel there), five otherwise.  out_idx (nullable): int32 indices of the batch, written by whichever block ends it.
hipError_t knn_cells_query(FilterState &st, FilterWorkspace &w, int m, const float *q, const float *r, long long base,
                           u64 *keys, int num_cu, bool timed, hipStream_t s, bool init_keys, int *out_idx)
{
    u64 *keys_init = init_keys ? keys : nullptr;
    FTRY(ensure_cells_workspace(st, w, m));
    const CellIndex &c = *st.cells;
    const int m_padded = (m + 31) / 32 * 32;
    const CellGeom g = cell_geom_of(c, st.k);
    w.has_rows = false;
    w.pieces = RerankPieces();
    const double sigma2 = (double)st.sigma * (double)st.sigma;
    // scan grid: two blocks of CELL_SCAN_WAVES waves per CU fill every SIMD's registers (6 waves x 80) and give the shortest
    // launch — and leave nothing for the kernels of the next batch, so with batches in flight side by side the step was the
    // SUM of the kernels' durations.  One block per CU: the scan alone takes 10-20 % longer, the step of a shard of up to 2^14
    // cells 5-7 % less (0.0465 -> 0.0444 ms at 2^21 rows; at C3 the two are within 1 %: it keeps two).
    // (2^15 cells, n = 2^23: step 0.0795 -> 0.0777 with one, three runs each on one box; 2^16 cells, C3: 0.1237 -> 0.1223, not
    // worth the 8 % the launch itself gets longer)
    const bool one_block = st.scan_blocks == 1 || (st.scan_blocks == 0 && st.several_slots && c.ncells <= 32768u);
    // Who makes the cells' lists of queries: knn_cells_match_kernel in a launch of its own (rounds 2-4), or the scan's waves
    // for the items they take (round 5, cell_self_list).  Policy in knn_cells_lists_policy.
    const bool centred = c.centred;   // per-cell frames (knn_cells_recentre): the centred prep and scan, lists from the match launch
    const bool self_lists = st.kt == 1 && !centred && (st.cells_lists == 2 || (st.cells_lists == 0 && knn_cells_lists_policy(c.ncells, st.several_slots)));
    const CellScanPlan plan = knn_cells_scan_plan(num_cu, one_block ? 1 : 2, c.nitems, w.rec_cap, m_padded, self_lists, st.kt, centred);
    const unsigned gx = plan.blocks;
    w.nlists = plan.nlists;
    w.ovf_cap = plan.ovf_cap;
    w.ovf_base = plan.ovf_base;
    w.slice = plan.slice;

    // the control words of this batch were cleared by the previous batch on this slot (or at allocation)
    const unsigned parity = w.cell_batches++ & 1u;
    w.ctl_cur = w.ctl + KNN_CTL_WORDS * (1u + parity);
    unsigned *ctl_next = w.ctl + KNN_CTL_WORDS * (2u - parity);
    // batches in flight side by side: two waves per query (two seed cells each) — half the registers the launch holds,
    // 0.0421 -> 0.0408 ms per step at n_local 2^21 for 2 us more when a batch runs alone; else four waves per query
    SeedLayer layer;
    memset(&layer, 0, sizeof layer);
    if (c.geom && c.seed_layer) {
        layer.base = c.seed_layer;
        layer.cpr = c.geom->cells_per_rank;
        layer.tiles = (unsigned)c.geom->seed_tiles;
        layer.nranks = (unsigned)c.geom->nranks;
        for (int r_ = 0; r_ <= c.geom->nranks && r_ <= KNN_MAX_RANKS; ++r_)
            layer.first[r_] = c.geom->first_cell(r_);
        layer.part_bytes = c.geom->part_bytes();
    }
#define KNN_PREP_LAUNCH(PWV, SDV, ...)                                                                                     \
    hipLaunchKernelGGL((knn_cells_prep_kernel<PWV, SDV, ##__VA_ARGS__>), dim3((unsigned)m_padded), dim3(64 * PWV), 0, s, q, m, m_padded, g, \
                       c.bounds, sigma2, st.center, st.sigma, c.tile_start, st.ntiles, (const h8 *)st.ref_frags,            \
                       st.ref_norms2, layer, (h8 *)w.qry_frags, w.lo_tab, w.hi_tab, st.bmax, st.nmax, kAmaxLimit, w.thr,    \
                       w.dup, w.ctl_cur, ctl_next, w.counts, w.nlists, keys_init, self_lists ? 1 : 0, c.cell_frame, c.tile_cell)
    // (cell-range shards take the same four seed cells — those of another rank through the seed layer.  Sixteen seed cells
    // (SD = 4) leave 20 % fewer candidates, as the simulation said, and cost more than they save: the prep kernel is a chain
    // of dependent round trips, and at a rank's size the step is made of those — emulated rank of N = 8, ms per step / one
    // batch at a time: 16 cells x 2 tiles 0.0302 / 0.0526, 8 x 2 0.0279 / 0.0493, 4 x 2 0.0275 / 0.0471, 4 x 4 0.0271 / 0.0473;
    // N = 4: 0.0432 / 0.0423 / 0.0414 / 0.0408.  profiles/r04_seed_sweep.txt)
    if (centred) {
        if (st.several_slots)
            KNN_PREP_LAUNCH(2, 2, 1, true);
        else
            KNN_PREP_LAUNCH(4, 2, 1, true);
    } else if (st.kt == 2) {
        if (st.several_slots)
            KNN_PREP_LAUNCH(2, 2, 2);
        else
            KNN_PREP_LAUNCH(4, 2, 2);
    } else if (st.several_slots)
        KNN_PREP_LAUNCH(2, 2, 1);
    else
        KNN_PREP_LAUNCH(4, 2, 1);
#undef KNN_PREP_LAUNCH
    FTRY(hipGetLastError());
    if (self_lists) {
        // no match launch: the scan's waves list their own items
    } else {
        // entries of a list assembled in LDS: k <= 16 the first 128 (16.6 KiB: four or five blocks per CU), 16 < k <= 32 the whole
        // list up to 640 entries (lists of 150-400 are the rule there)
        // (a rank of 8 of C3 — 2^13 cells, room for 1024 entries, lists of ~25 — on one box, ms per pipelined step: nothing staged (rounds
        // 2-5) 0.0290 / 0.0301, 128 entries 0.0266 / 0.0269, 256 entries 0.0274 / 0.0280)
        const unsigned stage = st.kt == 1 ? std::min(c.cap, 128u) : c.cap <= CELL_MATCH_STAGED_CAP ? c.cap : 0u;
        const size_t mlds = stage ? (size_t)64 * (stage + 2u) * sizeof(unsigned short) : 0;
        if (c.ncells <= 16384u) {
            if (mlds > (size_t)(48u << 10))   // (12-20 KiB of static LDS in front of it; the attribute is per device, so per launch)
                FTRY(hipFuncSetAttribute((const void *)knn_cells_match_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
            hipLaunchKernelGGL(knn_cells_match_kernel<16>, dim3(c.ncells / 64u), dim3(64 * 16), mlds, s, w.lo_tab, w.hi_tab, w.dup, m,
                               m_padded, g, c.ncells, c.cap, w.cell_lists, w.cell_counts, w.ctl_cur, stage);
        } else {
            if (mlds > (size_t)(48u << 10))
                FTRY(hipFuncSetAttribute((const void *)knn_cells_match_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
            hipLaunchKernelGGL(knn_cells_match_kernel<8>, dim3(c.ncells / 64u), dim3(64 * 8), mlds, s, w.lo_tab, w.hi_tab, w.dup, m,
                               m_padded, g, c.ncells, c.cap, w.cell_lists, w.cell_counts, w.ctl_cur, stage);
        }
    }
    FTRY(hipGetLastError());
    CellSelf self;
    memset(&self, 0, sizeof self);
    if (self_lists) {
        self.lo_t = w.lo_tab;
        self.hi = w.hi_tab;
        self.dup = w.dup;
        self.sa = c.sa;
        self.m_padded = m_padded;
    }
    if (centred) {
        self.dup = w.dup;
        self.frame = c.cell_frame;
    }
    const unsigned list_cap = self_lists ? CELL_SELF_CAP : c.cap;
    static const bool trace_cells = getenv("KNN_MI355X_TRACE_CELLS") != nullptr;   // (read once: a query may run beside a thread that changes the environment)
    if (trace_cells && !self_lists) {   // development aid: the lists of this batch and how evenly the scan's waves are loaded (synchronises)
        std::vector<unsigned> hc((size_t)c.ncells), ht((size_t)c.ncells + 1);
        FTRY(hipStreamSynchronize(s));
        FTRY(hipMemcpy(hc.data(), w.cell_counts, hc.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        FTRY(hipMemcpy(ht.data(), c.tile_start, ht.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        std::vector<unsigned> sorted(hc);
        std::sort(sorted.begin(), sorted.end());
        const unsigned nwaves = w.nlists;
        std::vector<double> wl((size_t)nwaves, 0.0);
        double total = 0, biggest = 0;
        for (unsigned cell = 0; cell < c.ncells; ++cell) {
            const double steps = (double)((std::min(hc[cell], c.cap + 1u) + 31u) / 32u) * (double)(ht[cell + 1] - ht[cell]);
            wl[cell % nwaves] += steps;   // (plain order: wave w takes cells w, w + nwaves, ...; the scan's scatter spreads them)
            total += steps;
            biggest = std::max(biggest, steps);
        }
        const double wmax = *std::max_element(wl.begin(), wl.end());
        fprintf(stderr, "[knn cells] m %d: %u cells; list length min %u  p10 %u  median %u  p90 %u  max %u; tile steps %.0f in all, %.0f in the largest "
                        "cell; %u waves: %.1f steps each on average, %.0f on the busiest\n", m, c.ncells, sorted.front(), sorted[sorted.size() / 10],
                sorted[sorted.size() / 2], sorted[sorted.size() * 9 / 10], sorted.back(), total, biggest, nwaves, total / nwaves, wmax);
    }
    if (timed && w.ev_begin)
        FTRY(hipEventRecord(w.ev_begin, s));
    const size_t lds = plan.lds_bytes;
    // Items handed out inside the block (LDS counter, see the kernel) when batches come one at a time: the launch is as long as
    // its busiest wave, and evening the waves out takes 6-8 % off it (alone, ms: 0.0433 -> 0.0408 at 2^21 rows, 0.060 -> 0.055 at
    // 2^22, 0.1176 -> 0.1094 at C3; one batch at a time 0.0855 -> 0.0828, 0.102 -> 0.097, 0.1677 -> 0.160).  With batches in
    // flight on several slots the next batch's kernels fill the gaps early finishers leave, and the fixed deal's cheaper prologue
    // and moving window win: the step is 1-4 % SHORTER with it (0.0415 / 0.0549 / 0.0794 / 0.1233 against 0.0425 / 0.0568 /
    // 0.0831 / 0.1249) — so that is what pipelined callers get.
    // (below two items per wave the counter has nothing to even out: 2^21 rows, one batch at a time, 0.0786 ms fixed / 0.0800 counter)
    // (round 4: the counter form lost its 8 bytes of scratch and requests its first descriptors in front of the LDS fill — 3.6 %
    // shorter alone at C3, 0.1127 against 0.1169 ms between events — and with that it also gives the shorter PIPELINED step on
    // the large shards that keep two blocks per CU, 0.1188-0.1202 against 0.1197-0.1214; a rank of eight stays with the fixed
    // deal, 0.0258-0.0265 against 0.0266-0.0293)
    const bool dyn = st.scan_deal == 2 ||
                     (st.scan_deal == 0 && c.nitems >= 2u * w.nlists && (!st.several_slots || c.ncells > 32768u));
    CellFinal fin;
    fin.gids = c.gids;
    fin.out_idx = out_idx;
    fin.defer = st.n_outliers != 0u ? 1 : 0;
    const long long npos = st.ntiles * 32;
#define KNN_SCAN_LAUNCH(DYNV, KV, SELFV, ...)                                                                              \
    hipLaunchKernelGGL((knn_cells_scan_kernel<DYNV, KV, SELFV, ##__VA_ARGS__>), dim3(gx), dim3(64 * plan.waves), lds, s, (const h8 *)st.ref_frags, \
                       st.ref_norms, c.items, c.nitems, (const h8 *)w.qry_frags, w.thr, m, m_padded, w.cell_counts,        \
                       w.cell_lists, list_cap, w.records, w.counts, w.ctl_cur, w.slice, w.ovf_base, w.ovf_cap, q, r, st.k, \
                       c.perm, npos, base, keys, fin, self)
#define KNN_SCAN_LAUNCH_K(DYNV, SELFV)                                                                                     \
    do {                                                                                                                   \
        if (st.k == 16)                                                                                                    \
            KNN_SCAN_LAUNCH(DYNV, 16, SELFV);                                                                              \
        else                                                                                                               \
            KNN_SCAN_LAUNCH(DYNV, 0, SELFV);                                                                               \
    } while (0)
#define KNN_SCAN_LAUNCH_K5(DYNV)   /* per-cell frames: 92 KiB of dynamic LDS for 1024 queries (their fp32 rows) */                  \
    do {                                                                                                                   \
        if (st.k == 16) {                                                                                                  \
            if (lds > (size_t)(64u << 10))                                                                                 \
                FTRY(hipFuncSetAttribute((const void *)knn_cells_scan_kernel<DYNV, 16, false, 1, true>,                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                          \
            KNN_SCAN_LAUNCH(DYNV, 16, false, 1, true);                                                                     \
        } else {                                                                                                           \
            if (lds > (size_t)(64u << 10))                                                                                 \
                FTRY(hipFuncSetAttribute((const void *)knn_cells_scan_kernel<DYNV, 0, false, 1, true>,                     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                          \
            KNN_SCAN_LAUNCH(DYNV, 0, false, 1, true);                                                                      \
        }                                                                                                                  \
    } while (0)
    if (centred) {   // per-cell frames: lists from the match launch
        if (dyn)
            KNN_SCAN_LAUNCH_K5(true);
        else
            KNN_SCAN_LAUNCH_K5(false);
    } else if (st.kt == 2 && st.k <= KNN_NIF_MAX_K) {   // 16 < k <= 30: run-time k, lists from the match launch, norms in the fragments
        if (lds > (size_t)(64u << 10)) {   // (more than the default limit of dynamic LDS: say so, per launch — the attribute is per device)
            if (dyn)
                FTRY(hipFuncSetAttribute((const void *)knn_cells_scan_kernel<true, 0, false, 2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            else
                FTRY(hipFuncSetAttribute((const void *)knn_cells_scan_kernel<false, 0, false, 2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (dyn)
            KNN_SCAN_LAUNCH(true, 0, false, 2, false, true);
        else
            KNN_SCAN_LAUNCH(false, 0, false, 2, false, true);
    } else if (st.kt == 2) {   // k 31, 32: no free K-slots — the norm window
        if (lds > (size_t)(64u << 10)) {   // (more than the default limit of dynamic LDS: say so, per launch — the attribute is per device)
            if (dyn)
                FTRY(hipFuncSetAttribute((const void *)knn_cells_scan_kernel<true, 0, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            else
                FTRY(hipFuncSetAttribute((const void *)knn_cells_scan_kernel<false, 0, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (dyn)
            KNN_SCAN_LAUNCH(true, 0, false, 2);
        else
            KNN_SCAN_LAUNCH(false, 0, false, 2);
    } else if (dyn) {
        if (self_lists)
            KNN_SCAN_LAUNCH_K(true, true);
        else
            KNN_SCAN_LAUNCH_K(true, false);
    } else {
        if (self_lists)
            KNN_SCAN_LAUNCH_K(false, true);
        else
            KNN_SCAN_LAUNCH_K(false, false);
    }
#undef KNN_SCAN_LAUNCH_K5
#undef KNN_SCAN_LAUNCH_K
#undef KNN_SCAN_LAUNCH
    FTRY(hipGetLastError());
    if (timed && w.ev_end)
        FTRY(hipEventRecord(w.ev_end, s));
    // rows outside the robust box never entered the layouts: exact scan of that (short) list
    FTRY(knn_exact_gather_launch(st.k, m, st.n_outliers, base, q, r, st.outliers, keys, num_cu, nullptr, s));
    // gated on the device: the whole shard exactly when the batch has a query nothing bounds (k = 16: inside the tail kernel)
    if (st.k != 16)
        FTRY(knn_exact_launch(st.k, m, st.n, base, q, r, keys, num_cu, w.ctl_cur + KNN_CTL_FALLBACK, s));
    {
        unsigned blocks = (unsigned)num_cu * 8u;
        if (blocks * KNN_WAVES > c.nitems)
            blocks = std::max(2u, (c.nitems + KNN_WAVES - 1u) / KNN_WAVES);
#define KNN_TAIL_LAUNCH(...)                                                                                               \
    hipLaunchKernelGGL((knn_cells_tail_kernel<__VA_ARGS__>), dim3(blocks), dim3(KNN_BLOCK), 0, s, q, r, st.k, m, st.n, npos, base, c.items,  \
                       c.nitems, w.cell_counts, w.cell_lists, list_cap, c.perm, w.records, w.ovf_base, w.ovf_cap, w.ctl_cur, \
                       keys, fin, w.counts, w.nlists, w.slice, self)
        if (st.kt == 2)
            KNN_TAIL_LAUNCH(0, 2);
        else
            switch (st.k) {
            case 16: KNN_TAIL_LAUNCH(16); break;
            case 8: KNN_TAIL_LAUNCH(8); break;
            default: KNN_TAIL_LAUNCH(0); break;
            }
#undef KNN_TAIL_LAUNCH
    }
    return hipGetLastError();
}
