// knn_filter.hip — low-precision MFMA filter with a rigorous error bound, in front of the exact
// re-rank (gfx950 / MI355X).
//
// Why: at m = 1024 the exact scan needs (3k+3)·m·n fp32 VALU lane-ops — ~80x more time than
// reading the references from HBM.  The distance matrix is a dense
// contraction, |q-r|^2 = |q|^2 + |r|^2 - 2 q·r, and k = 16 is exactly one K-step of
// v_mfma_f32_32x32x16_f16.  The filter evaluates, for every (query, reference) pair,
//      S = N_r + sum_d (-2 a_d) * b_d          (fp16 operands, fp32 accumulate, C operand = N_r)
// where a, b are the centred, power-of-two scaled coordinates rounded to fp16 and N_r = |b|^2,
// and keeps only pairs with S <= thr_q.  thr_q is derived so that the true nearest reference
// (under the reference's v0 fp32 arithmetic, core.cu:44-54) can never be discarded:
//
//   * an exact v0 distance E0 to some reference is known for the query (sample pre-pass with
//     the exact kernel), so the winner j* has E_j* <= E0 and real distance
//     D* <= sigma^2 (E0 (1+g2) + tau)                       [v0's own rounding: g2 = (k+3) 2^-24]
//   * fp16 rounding of centred coordinates moves each coordinate difference by at most
//     e = theta'(Amax + Bmax) + 2 nu0, so |D~ - D| <= 2 eta sqrt(D) + eta^2, eta = sqrt(k) e
//   * norm sums and the MFMA's internal fp32 accumulation add at most rho.  The accumulation term is the ONE
//     constant of this bound that is ASSUMED, not proven: omega = kt * 2^-18 relative to the sum of term
//     magnitudes (16x a single fp32 rounding per 16-wide K-step; the matrix core's internal summation order is
//     not documented).  It is measured on every GPU test run: test_mfma_accumulation_error_is_far_inside_the_
//     assumed_allowance rebuilds the fp16 operands on the host, evaluates N - 2 a.b in float64 and asserts the
//     device scores are within 2^-20 (a 4x margin to the allowance) for k = 16 ... 1024; test_filter_scores_stay_
//     inside_the_proven_error_bound checks the whole bound against float64 distances on 40 data sets (k = 3 ... 1100).
//   => S_j* <= Dup + 2 eta sqrt(Dup) + eta^2 + rho - M_q(1-g) =: thr_q.
//
// Survivors are written as records (query, reference tile, lane half) and re-evaluated with the
// exact arithmetic by knn_rerank_kernel, which folds them into the packed keys.  If anything
// rules the filter out (non-finite data, fp16 range, record overflow) a device-side flag makes
// the gated exact kernels scan everything instead: results are bit-exact either way.
//
// Scans by dimension (kt = 16-wide K-steps of the padded k, knn_kt_of): kt 1, 2 (and kt 4, 8 with few queries)
// knn_filter_kernel, fragments in registers; kt 4, 8 knn_filter_tiled_kernel, reference tiles shared by a block's waves
// through LDS, 4 query blocks per wave; kt 16, 32 (128 < k <= 512) the same kernel with 2 / 1 query blocks per wave;
// beyond (k <= 4096) knn_filter_chunked_kernel, K walked in chunks of 128 dimensions.  k <= 16 on large resident
// shards: the cell-pruned form, knn_cells.hip.
#include "knn_filter_dev.h"

#include <functional>
#include <thread>

// ------------------------------------------------------------------------------------------
// Per-dimension min / max of the reference coordinates (+ count of non-finite values).
//   stats[0..k) = ordered min, stats[k..2k) = ordered max, stats[2k] = #non-finite
// Each thread keeps one fixed dimension: the element stride is a multiple of k.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_ref_stats_kernel(const float *__restrict__ R, long long count,
                                                            int k, unsigned *__restrict__ stats)
{
    const long long threads = (long long)gridDim.x * blockDim.x;
    const long long stride = threads / k * k;
    const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int d = (int)(gtid % k);
    float lo = INFINITY, hi = -INFINITY;
    unsigned bad = 0;
    for (long long e = gtid < stride ? gtid : count; e < count; e += stride) {
        const float v = R[e];
        if (!(fabsf(v) < INFINITY))
            ++bad;   // counted, kept out of the range: such rows leave the filter for the exact list
        else {
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
    }
    // fold the block in LDS first: one guarded global atomic per dimension per block —
    // per-thread atomics on the same 2k words ran at the single-word rate (1.4 ms for a 20 MB shard)
    // (dynamic LDS, 2 k words: a static [KNN_FILTER_MAX_K] pair was 32 KiB per block at every k — ADVICE r03)
    extern __shared__ unsigned s_dyn_stats[];
    unsigned *s_lo = s_dyn_stats, *s_hi = s_dyn_stats + k;
    __shared__ unsigned s_bad;
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        s_lo[i] = 0xFFFFFFFFu;
        s_hi[i] = 0u;
    }
    if (threadIdx.x == 0)
        s_bad = 0u;
    __syncthreads();
    if (lo <= hi) {
        atomicMin(&s_lo[d], f2ord(lo));
        atomicMax(&s_hi[d], f2ord(hi));
    }
    if (bad)
        atomicAdd(&s_bad, bad);
    __syncthreads();
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        if (s_lo[i] != 0xFFFFFFFFu)
            guarded_atomic_min(&stats[i], s_lo[i]);
        if (s_hi[i] != 0u)
            guarded_atomic_max(&stats[k + i], s_hi[i]);
    }
    if (threadIdx.x == 0 && s_bad)
        atomicAdd(&stats[2 * k], s_bad);
}

// Same, 16 bytes per lane per step (k % 4 == 0, 16-byte aligned rows): every lane keeps four fixed
// dimensions, four independent loads in flight.
__global__ __launch_bounds__(256) void knn_ref_stats4_kernel(const f4v *__restrict__ R4, long long count4,
                                                             int k, unsigned *__restrict__ stats)
{
    const int k4 = k >> 2;
    const long long threads = (long long)gridDim.x * blockDim.x;
    const long long stride = threads / k4 * k4;
    const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int d0 = (int)(gtid % k4) * 4;
    f4v lo = {INFINITY, INFINITY, INFINITY, INFINITY}, hi = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    unsigned bad = 0;
#pragma unroll 4
    for (long long e = gtid < stride ? gtid : count4; e < count4; e += stride) {
        const f4v v = __builtin_nontemporal_load(&R4[e]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (!(fabsf(v[c]) < INFINITY))
                ++bad;   // counted, kept out of the range
            else {
                lo[c] = fminf(lo[c], v[c]);
                hi[c] = fmaxf(hi[c], v[c]);
            }
        }
    }
    // fold the block in LDS first: one guarded global atomic per dimension per block
    // instead of eight per thread on the same two cache lines
    extern __shared__ unsigned s_dyn_stats[];   // 2 k words (see knn_ref_stats_kernel)
    unsigned *s_lo = s_dyn_stats, *s_hi = s_dyn_stats + k;
    __shared__ unsigned s_bad;
    for (int d = threadIdx.x; d < k; d += blockDim.x) {
        s_lo[d] = 0xFFFFFFFFu;
        s_hi[d] = 0u;
    }
    if (threadIdx.x == 0)
        s_bad = 0u;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (lo[c] <= hi[c]) {
            atomicMin(&s_lo[d0 + c], f2ord(lo[c]));
            atomicMax(&s_hi[d0 + c], f2ord(hi[c]));
        }
    if (bad)
        atomicAdd(&s_bad, bad);
    __syncthreads();
    for (int d = threadIdx.x; d < k; d += blockDim.x) {
        if (s_lo[d] != 0xFFFFFFFFu)
            guarded_atomic_min(&stats[d], s_lo[d]);
        if (s_hi[d] != 0u)
            guarded_atomic_max(&stats[k + d], s_hi[d]);
    }
    if (threadIdx.x == 0 && s_bad)
        atomicAdd(&stats[2 * k], s_bad);
}

// Copies every row_stride-th row into a dense buffer (robust-box statistics are done on the host).
__global__ __launch_bounds__(256) void knn_sample_rows_kernel(const float *__restrict__ R, int k, long long row_stride,
                                                              long long samples, float *__restrict__ out)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < samples * k)
        out[e] = R[(size_t)(e / k) * row_stride * k + (size_t)(e % k)];
}

// ------------------------------------------------------------------------------------------
// fp32 AoS rows -> centred, scaled fp16 MFMA operand fragments + fp32 squared norms.
// Fragment order: frag[(tile*KT + kt)*64 + half*32 + r] holds coordinates 16kt + 8half .. +7 of
// row 32*tile + r — exactly what lane (half, r) of v_mfma_f32_32x32x16_f16 wants for A (rows =
// references) or B (columns = queries), so the hot loop's operand loads are linear 1 KiB bursts.
//   scale_out = 1 for references, -2 for queries (folds the -2 of -2 q·r into the operand).
//   out[0] = max |fp16 coordinate| (float bits), out[1] = max norm, out[2] = #non-finite fp16
//   (accumulated with atomics into pre-zeroed words, or per-block partials — see the end)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_frag_kernel(const float *__restrict__ X, long long rows,
                                                       long long rows_padded, int k, int kt,
                                                       const float *__restrict__ center, float sigma,
                                                       float scale_out, float pad_norm,
                                                       h8 *__restrict__ frag, float *__restrict__ norms,
                                                       unsigned *__restrict__ out, int out_is_partials,
                                                       unsigned *__restrict__ ctl, float *__restrict__ rowmax,
                                                       unsigned *__restrict__ olist, unsigned ocap,
                                                       unsigned row_base = 0u,
                                                       const unsigned *__restrict__ gather = nullptr)
{
    // (row_base: first row of this launch inside the shard when the layouts are built chunk by chunk;
    // X / frag / norms arrive already offset to that row, only the outlier list needs the number)
    // (gather: position i of the layout holds row gather[i] of X, ~0u = padding — the cell-sorted layout)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float vmax = 0.0f, nrm = 0.0f;
    unsigned bad = 0;
    if (i < rows_padded) {
        const long long tile = i >> 5;
        const int r = (int)(i & 31);
        bool real = i < rows;
        long long src = i;
        if (gather) {
            const unsigned gs = gather[i];
            real = gs != 0xFFFFFFFFu;
            src = (long long)gs;
        }
        const float *__restrict__ x = X + (size_t)(real ? src : 0) * k;
        if (olist && real) {
            // reference rows outside the robust box (|scaled coordinate| > 1) leave the filter: zero
            // fragment, +INF norm (never a survivor), listed for the exact gather scan
            bool outside = false;
            for (int d = 0; d < k; ++d) {
                const float back = (float)(_Float16)((x[d] - center[d]) * sigma);
                outside = outside || !(fabsf(back) <= 1.0f);   // NaN included
            }
            if (outside) {
                const unsigned pos = atomicAdd(&out[3], 1u);
                if (pos < ocap)
                    olist[pos] = gather ? (unsigned)src : row_base + (unsigned)i;
                real = false;
            }
        }
        for (int kk = 0; kk < kt; ++kk) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                h8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int d = kk * 16 + half * 8 + j;
                    float s = 0.0f;
                    if (real && d < k)
                        s = (x[d] - center[d]) * sigma;  // fp32 subtract, exact power-of-two scale
                    const _Float16 hval = (_Float16)s;   // round to nearest even
                    const float back = (float)hval;
                    if (!(fabsf(back) < INFINITY))
                        ++bad;
                    vmax = fmaxf(vmax, fabsf(back));
                    nrm = nrm + back * back;             // exact products, fp32 sum
                    v[j] = (_Float16)(back * scale_out);  // *1 or *-2: exact in fp16 below overflow
                    if (!(fabsf((float)v[j]) < INFINITY))
                        ++bad;
                }
                frag[((size_t)tile * kt + kk) * 64 + half * 32 + r] = v;
            }
        }
        norms[i] = real ? nrm : pad_norm;
        if (rowmax)
            rowmax[i] = vmax;  // this row's max |coordinate|: per-query error bound
        if (!real)
            nrm = 0.0f;
    }
    vmax = wave_max_f(vmax);
    nrm = wave_max_f(nrm);
    if (out_is_partials) {
        // query batches: no atomics and nothing to pre-zero — out[3*block .. +2] = this block's
        // {max |coord|, max norm, #bad}; the threshold kernel folds the blocks.  Block 0 also
        // resets the per-call control words the later kernels accumulate into.
        __shared__ float s_v[4], s_n[4];
        __shared__ unsigned s_b;
        if (threadIdx.x == 0)
            s_b = 0u;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            s_v[threadIdx.x >> 6] = vmax;
            s_n[threadIdx.x >> 6] = nrm;
        }
        if (bad)
            atomicAdd(&s_b, bad);
        __syncthreads();
        if (threadIdx.x == 0) {
            out[3 * blockIdx.x + 0] = __float_as_uint(fmaxf(fmaxf(s_v[0], s_v[1]), fmaxf(s_v[2], s_v[3])));
            out[3 * blockIdx.x + 1] = __float_as_uint(fmaxf(fmaxf(s_n[0], s_n[1]), fmaxf(s_n[2], s_n[3])));
            out[3 * blockIdx.x + 2] = s_b;
            if (blockIdx.x == 0) {
                ctl[KNN_CTL_FALLBACK] = 0u;
                ctl[KNN_CTL_RECORDS] = 0u;
                ctl[KNN_CTL_EXACT_CELLS] = 0u;
            }
        }
        return;
    }
    if ((threadIdx.x & 63) == 0) {
        guarded_atomic_max(&out[0], __float_as_uint(vmax));  // non-negative floats order like uints
        guarded_atomic_max(&out[1], __float_as_uint(nrm));
    }
    if (bad)
        atomicAdd(&out[2], bad);
}

// k = 16 references, 16-byte aligned: the block reads its 256 rows (16 KiB) as fully coalesced
// 16-byte chunks into LDS (XOR-swizzled so the row reads below are bank-conflict free), then
// every thread converts its own row.  Same outputs as knn_frag_kernel(scale_out = 1).
__global__ __launch_bounds__(256) void knn_frag16_kernel(const f4v *__restrict__ X4, long long rows,
                                                         long long rows_padded,
                                                         const float *__restrict__ center, float sigma,
                                                         h8 *__restrict__ frag, float *__restrict__ norms,
                                                         unsigned *__restrict__ out,
                                                         unsigned *__restrict__ olist, unsigned ocap,
                                                         unsigned row_base = 0u)
{
    __shared__ f4v s_x[256 * 4];
    const long long row0 = (long long)blockIdx.x * 256;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = threadIdx.x + j * 256;       // chunk index inside the block's 1024 chunks
        const int rr = c >> 2, cc = c & 3;
        f4v v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (row0 + rr < rows)
            v = __builtin_nontemporal_load(&X4[(size_t)row0 * 4 + c]);
        s_x[rr * 4 + (cc ^ ((rr >> 2) & 3))] = v;
    }
    __syncthreads();
    const long long i = row0 + threadIdx.x;
    float vmax = 0.0f, nrm = 0.0f;
    unsigned bad = 0;
    if (i < rows_padded) {
        const long long tile = i >> 5;
        const int r = (int)(i & 31);
        bool real = i < rows;
        const int sw = ((int)threadIdx.x >> 2) & 3;
        h8 v[2];
        float rowmax = 0.0f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int cc = half * 2 + g;
                const f4v x = s_x[threadIdx.x * 4 + (cc ^ sw)];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d = cc * 4 + j;
                    const float sc = real ? (x[j] - center[d]) * sigma : 0.0f;
                    const _Float16 hval = (_Float16)sc;
                    const float back = (float)hval;   // beyond fp16 range -> inf -> outlier below
                    rowmax = fmaxf(rowmax, fabsf(back));
                    nrm = nrm + back * back;
                    v[half][g * 4 + j] = hval;
                }
            }
        }
        if (real && (!(rowmax <= 1.0f) || !(nrm < INFINITY))) {
            // outside the robust box (or a NaN coordinate, which fmaxf above skips): out of the filter
            // (zero fragment, +INF norm), into the exact list
            const unsigned pos = atomicAdd(&out[3], 1u);
            if (pos < ocap)
                olist[pos] = row_base + (unsigned)i;
            real = false;
            v[0] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
            v[1] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
            rowmax = 0.0f;
        }
        vmax = rowmax;
        frag[(size_t)tile * 64 + r] = v[0];
        frag[(size_t)tile * 64 + 32 + r] = v[1];
        norms[i] = real ? nrm : INFINITY;
        if (!real)
            nrm = 0.0f;
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
    if (bad)
        atomicAdd(&out[2], bad);
}

// (BoundConsts, knn_bound_consts, knn_threshold: knn_filter_dev.h)

// umin: per-block minima of the sample pass, [nblocks][m_padded].  Block = 32 queries x 32
// parts: each part folds every 32nd sample block (coalesced over the 32 queries, all its loads
// in flight at once), LDS folds the parts.
#define THR_PARTS 32
__global__ __launch_bounds__(32 * THR_PARTS) void knn_thr_kernel(const float *__restrict__ umin, int nblocks,
                                                      const float *__restrict__ qnorm,
                                                      const float *__restrict__ qamax, int m,
                                                      int m_padded, int k, int kt, float sigma,
                                                      float bmax, float nmax, float amax_limit,
                                                      float *__restrict__ thr,
                                                      unsigned *__restrict__ ctl,
                                                      const unsigned *__restrict__ qpart, int qblocks,
                                                      unsigned *__restrict__ counts, unsigned nlists,
                                                      float *__restrict__ dup_out = nullptr,
                                                      // running thresholds of the deep-K scan (round 5; all three or none):
                                                      float *__restrict__ marg_out = nullptr, float *__restrict__ floor_out = nullptr,
                                                      unsigned *__restrict__ run_out = nullptr)
{
    __shared__ float s_part[THR_PARTS][32];
    // housekeeping folded in here to save launches: zero the record counters of the filter pass
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < nlists; i += gridDim.x * blockDim.x)
        counts[i] = 0u;
    const int ql = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + ql;  // m_padded is a multiple of 32
    float u = INFINITY;
#pragma unroll 16
    for (int b = part; b < nblocks; b += THR_PARTS)
        u = fminf(u, umin[(size_t)b * m_padded + i]);
    s_part[part][ql] = u;
    __syncthreads();
    if (part != 0)
        return;
#pragma unroll
    for (int p = 1; p < THR_PARTS; ++p)
        u = fminf(u, s_part[p][ql]);
    float amax = 0.0f;
    unsigned qbad = 0u;
    for (int b = 0; b < qblocks; ++b) {  // per-block partials of the query fragment kernel
        amax = fmaxf(amax, __uint_as_float(qpart[3 * b]));
        qbad |= qpart[3 * b + 2];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl[KNN_CTL_AMAX] = __float_as_uint(amax);
        ctl[KNN_CTL_QBAD] = qbad;
    }
    bool bad = qbad != 0u || !(amax <= amax_limit);
    float t = -INFINITY;  // padding queries never pass
    float dupf = -INFINITY;
    if (i < m) {
        if (!(u < INFINITY))
            bad = true;  // no finite sample score: cannot bound
        if (!bad) {
            // the error bound of THIS query: its own coordinate magnitude, not the batch maximum
            // (one far-away query must not loosen everybody's threshold)
            const BoundConsts c = knn_bound_consts(k, kt, sigma, qamax[i], bmax, nmax);
            double dup = 0.0;
            t = knn_threshold(c, u, qnorm[i], &dup);
            if (!(t < INFINITY))
                bad = true;
            else {
                dup *= 1.0 + 1e-6;
                dupf = (float)dup;
                if ((double)dupf < dup)
                    dupf = nextafterf(dupf, INFINITY);
            }
        }
    }
    if (dup_out)
        dup_out[i] = dupf;
    thr[i] = t;
    if (run_out) {
        // Running thresholds (round 5).  knn_threshold(u) is the threshold implied by the score u of ANY real row; the scan
        // meets lower scores than the sample pass's minimum u0 as it goes, and every one of them implies a lower threshold.
        // Recomputing knn_threshold in the scan (double arithmetic, two square roots) per improvement costs more than it
        // saves; instead: d thr / d u >= 1 wherever the clamp `dt < 0` is not active — thr = dup (1 + ...) + const with
        // d dup / d u = (1 + g2)^2 (1 + eta / sqrt(dt + 2 eta^2)) >= (1 + g2)^2, and (1 + g2)^2 - 1 >= 2 (k + 3) 2^-24 covers the
        // 1e-6 |thr| slack term's own slope for k >= 33 (the only callers) — so for a real row's score u' <= u0
        //     thr(u') <= max(thr(-inf), thr(u0) - (u0 - u')) = max(floor, u' + margin),     margin = thr(u0) - u0,
        // with thr(-inf) the value under the clamp (the lowest threshold any score implies).  margin is rounded UP with 2^-20
        // relative slack for the float additions on either side; a larger threshold only keeps more candidates.
        float mg = INFINITY, fl = -INFINITY;
        if (i < m && !bad) {
            const BoundConsts c = knn_bound_consts(k, kt, sigma, qamax[i], bmax, nmax);
            fl = knn_threshold(c, -INFINITY, qnorm[i]);
            const double md = (double)t - (double)u + 0x1p-20 * (fabs((double)t) + fabs((double)u)) + 1e-30;
            mg = (float)md;
            if ((double)mg < md)
                mg = nextafterf(mg, INFINITY);
        }
        marg_out[i] = mg;
        floor_out[i] = fl;
        run_out[i] = f2ord(t);
    }
    if (bad)
        ctl[KNN_CTL_FALLBACK] = 1u;  // benign race: every writer stores 1
}

// ------------------------------------------------------------------------------------------
// The filter.  One wave owns a contiguous range of 32-reference tiles; it keeps the B operands
// (queries) of QT query tiles and their thresholds in VGPRs, streams A operands (references) and
// the C tile (reference norms broadcast along the query axis) once from HBM with the next tile
// prefetched, and per (reference tile, query tile) issues KT MFMAs + an 8-op min3 tree + one
// compare.  Rows (registers) = references, columns (lanes) = queries: the reduction over
// references is in-lane, no cross-lane traffic, no LDS.
//   grid.x * 4 waves split the reference tiles; grid.y = groups of QT query tiles.
// ------------------------------------------------------------------------------------------
template <int KT>
__device__ __forceinline__ void load_ref_tile(const h8 *__restrict__ rf, const float *__restrict__ rn,
                                              long long tile, int lane, h8 (&a)[KT], f16v &c)
{
#pragma unroll
    for (int kk = 0; kk < KT; ++kk)
        a[kk] = rf[((size_t)tile * KT + kk) * 64 + lane];
    // C[row][col] = N_row: rows of register i are 8*(i>>2) + 4*half + (i&3)
    const f4v *__restrict__ rn4 = (const f4v *)(rn + (size_t)tile * 32 + 4 * (lane >> 5));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f4v v = rn4[2 * g];
        c[4 * g + 0] = v[0];
        c[4 * g + 1] = v[1];
        c[4 * g + 2] = v[2];
        c[4 * g + 3] = v[3];
    }
}


// One (reference tile, query tile) step, software-pipelined by one tile: the MFMA of the NEXT
// query tile is issued first (into the other accumulator buffer), then the min3 tree of the
// current one runs in its shadow.  Query tiles past the end of the batch hold a copy of the last
// real tile with threshold -INF: wasted MFMAs, never a record (the host picks QT to fit m).
template <int KT, int QT>
__device__ __forceinline__ void filter_ref_tile(const h8 (&a)[KT], const f16v &c, const h8 (&qf)[QT][KT],
                                                const float *__restrict__ s_thr, int lane, int qt0,
                                                long long tile, u64 *__restrict__ my_rec,
                                                unsigned &cnt, unsigned slice)
{
    f16v d[2];
    d[0] = c;
#pragma unroll
    for (int kk = 0; kk < KT; ++kk)
        d[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], qf[0][kk], d[0], 0, 0, 0);
    // The QT steps are ONE basic block: the hit masks are parked in SGPR pairs and looked at once
    // after the last step.  (With a branch per step the accumulator written by an MFMA was read
    // in a later block, and the compiler's cross-block MFMA->VALU wait-state count came out short
    // of the 12 the in-block rule gives: stale accumulator reads, i.e. missed survivors.)
    u64 masks[QT];
    u64 any = 0ull;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float th = s_thr[t * 32 + (lane & 31)];
        if (t + 1 < QT) {
            f16v &dn = d[(t + 1) & 1];
            dn = c;
#pragma unroll
            for (int kk = 0; kk < KT; ++kk)
                dn = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], qf[t + 1][kk], dn, 0, 0, 0);
        }
        const f16v &x = d[t & 1];
        const float m0 = min3f(x[0], x[1], x[2]);
        const float m1 = min3f(x[3], x[4], x[5]);
        const float m2 = min3f(x[6], x[7], x[8]);
        const float m3 = min3f(x[9], x[10], x[11]);
        const float m4 = min3f(x[12], x[13], x[14]);
        const float m5 = min3f(m0, m1, m2);
        const float m6 = min3f(m3, m4, x[15]);
        const float mn = min3f(m5, m6, th);
        masks[t] = __ballot(mn < th);  // rare: one of this lane's 16 rows may beat the bound
        any |= masks[t];
    }
    if (__builtin_expect(any != 0ull, 0)) {  // wave-uniform
        const u64 me = 1ull << lane;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const u64 mask = masks[t];
            if (mask != 0ull) {
                if (mask & me) {
                    const unsigned pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                         __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    if (pos < slice)
                        my_rec[pos] = ((u64)(unsigned)((qt0 + t) * 32 + (lane & 31)) << 32) | ((u64)tile << 1) |
                                      (u64)(lane >> 5);
                }
                cnt += (unsigned)__popcll(mask);
            }
        }
    }
}

template <int KT, int QT>
__global__ __launch_bounds__(FILTER_BLOCK, (KT == 1 && QT <= 16 ? 3 : 2)) void knn_filter_kernel(
    const h8 *__restrict__ rf, const float *__restrict__ rn, const h8 *__restrict__ qfg,
    const float *__restrict__ thrg, int qtiles, long long ntiles, u64 *__restrict__ rec,
    unsigned *__restrict__ counts, unsigned *__restrict__ ctl, unsigned slice)
{
    // One launch covers a PIECE of the batch: `qtiles` query tiles in groups of QT; qfg / thrg / rec /
    // counts arrive already offset to the piece and records carry piece-relative query numbers (the
    // re-rank adds the piece's first query: one more live value in this kernel costs it 6 %) (a batch is cut into pieces
    // with different QT so that a ragged tail does not pay for a full group, see plan_pieces()).
    // Records go to a slice of `rec` private to this wave (no shared counter: a single atomic
    // word serialises at ~88 returns/us); counts[wave] = records the wave wanted to write.
    __shared__ float s_thr[QT * 32];
    if (ctl[KNN_CTL_FALLBACK] != 0u)
        return;
    const int lane = threadIdx.x & 63;
    const int qt0 = blockIdx.y * QT;       // relative to the piece
    const int nq = min(QT, qtiles - qt0);  // wave-uniform
    for (int i = threadIdx.x; i < QT * 32; i += FILTER_BLOCK)
        s_thr[i] = i < nq * 32 ? thrg[(size_t)qt0 * 32 + i] : -INFINITY;
    __syncthreads();

    const long long wave = (long long)blockIdx.x * (FILTER_BLOCK / 64) +
                           __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long nwaves = (long long)gridDim.x * (FILTER_BLOCK / 64);
    const long long t0 = ntiles * wave / nwaves;
    const long long t1 = ntiles * (wave + 1) / nwaves;
    const size_t list = (size_t)blockIdx.y * (size_t)nwaves + (size_t)wave;
    u64 *__restrict__ my_rec = rec + list * slice;
    unsigned cnt = 0u;

    h8 qf[QT][KT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int tt = min(t, nq - 1);
#pragma unroll
        for (int kk = 0; kk < KT; ++kk)
            qf[t][kk] = qfg[((size_t)(qt0 + tt) * KT + kk) * 64 + lane];
    }

    if constexpr (KT == 1 && QT <= 2) {
        // Small batches are bound by memory instructions, not MFMAs: per 1 KiB fragment tile the
        // four broadcast norm loads cost four more TA passes.  Here the wave stages the norms of 8
        // tiles with ONE coalesced 1 KiB load into a private LDS window (double-buffered) and reads
        // its C tile back with four broadcast ds_read_b128; fragments are prefetched PF tiles ahead.
        __shared__ f4v s_nrm[FILTER_BLOCK / 64][2][64];
        const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        if (t0 < t1) {
            const long long nf4 = ntiles * 8;  // float4 chunks in the norm array
            auto stage = [&](long long chunk_tile, int buf) {
                const long long i4 = chunk_tile * 8 + lane;
                s_nrm[wib][buf][lane] = i4 < nf4 ? *(const f4v *)(rn + i4 * 4) : (f4v){0.f, 0.f, 0.f, 0.f};
            };
            const long long c0 = t0 & ~7ll;  // chunks are 8 tiles, aligned
            stage(c0, 0);
            constexpr int PF = 4;  // fragment tiles in flight per wave; slots are refilled in place
            h8 ar[PF];                 // (copying a just-loaded register would wait for the load)
#pragma unroll
            for (int p = 0; p < PF; ++p)
                ar[p] = rf[(size_t)min(t0 + p, t1 - 1) * 64 + lane];
            for (long long base_tile = t0; base_tile < t1; base_tile += PF) {
#pragma unroll
                for (int p = 0; p < PF; ++p) {
                    const long long tile = base_tile + p;
                    if (tile < t1) {  // wave-uniform
                        const int buf = (int)(((tile - c0) >> 3) & 1);
                        if (((tile - c0) & 7) == 0 || tile == t0) {
                            // entering a chunk: make its norms visible to the whole wave, prefetch the next
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            const long long next_chunk = ((tile - c0) & ~7ll) + c0 + 8;
                            if (next_chunk < t1)
                                stage(next_chunk, buf ^ 1);
                        }
                        h8 a[1] = {ar[p]};
                        f16v c;
                        const int ti = (int)((tile - c0) & 7);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f4v v = s_nrm[wib][buf][ti * 8 + 2 * g + (lane >> 5)];
                            c[4 * g + 0] = v[0];
                            c[4 * g + 1] = v[1];
                            c[4 * g + 2] = v[2];
                            c[4 * g + 3] = v[3];
                        }
                        filter_ref_tile<KT, QT>(a, c, qf, s_thr, lane, qt0, tile, my_rec, cnt, slice);
                        ar[p] = rf[(size_t)min(tile + PF, t1 - 1) * 64 + lane];  // refill this slot
                    }
                }
            }
        }
    } else if (t0 < t1) {
        // two operand sets used alternately in place (no register copies between tiles)
        h8 aA[KT], aB[KT];
        f16v cA, cB;
        load_ref_tile<KT>(rf, rn, t0, lane, aA, cA);
        for (long long tile = t0; tile < t1; tile += 2) {
            load_ref_tile<KT>(rf, rn, min(tile + 1, t1 - 1), lane, aB, cB);  // prefetch
            filter_ref_tile<KT, QT>(aA, cA, qf, s_thr, lane, qt0, tile, my_rec, cnt, slice);
            if (tile + 1 < t1) {  // wave-uniform
                load_ref_tile<KT>(rf, rn, min(tile + 2, t1 - 1), lane, aA, cA);
                filter_ref_tile<KT, QT>(aB, cB, qf, s_thr, lane, qt0, tile + 1, my_rec, cnt, slice);
            }
        }
    }
    if (lane == 0) {
        counts[list] = cnt;
        if (cnt > slice)
            ctl[KNN_CTL_FALLBACK] = 1u;  // survivors were dropped: the exact scan will take over
    }
}

// ------------------------------------------------------------------------------------------
// Deep-K variant (k > 32, e.g. config C5 k = 128): with KT K-steps per tile the A fragments of a
// reference tile are KT KiB, and the register-resident scheme above can only keep 2 query tiles per
// wave, so every KiB of A was used for 2 MFMAs and the kernel sat on L2 bandwidth (1.0 PF at k = 128).
// Here a block's 4 waves share each reference tile through LDS (double-buffered, one barrier per
// tile) and each wave owns QT = 4 DIFFERENT query tiles with one accumulator per query tile that
// runs down the K-steps: 4*QT*KT MFMAs per staged tile, 8x less A traffic.
//   SAMPLE = true : every stride-th tile, running minima -> umin[blockIdx.x][query]
//   SAMPLE = false: all tiles, threshold test -> per-wave record slices (as knn_filter_kernel)
//   grid.x blocks split the (sampled) reference tiles, grid.y = groups of 4*QT query tiles.
// ------------------------------------------------------------------------------------------
//   X16 (round 4): the same tiles out of v_mfma_f32_16x16x32_f16 — four 16 x 16 sub-tiles per (32 references x 32 queries),
//   two K-steps per instruction — instead of v_mfma_f32_32x32x16_f16.  Same flops per cycle, same layouts (the LDS and
//   register operands are fetched with per-lane addresses: any lane order is free), same records; what differs is the clock
//   the chip holds: register-only chains on random f16 data run 1.72-1.75 PFLOP/s at 1.70 GHz with the 32 x 32 shape and
//   1.98 at 1.96 GHz with the 16 x 16 one (tools/mfma_shape_probe.hip, profiles/r04_mfma_shape_probe.txt).
template <int KT, int QT, bool SAMPLE, int BLOCK = FILTER_BLOCK, int TPB = 1, bool X16 = false>
__global__ __launch_bounds__(BLOCK, 2) void knn_filter_tiled_kernel(
    const h8 *__restrict__ rf, const float *__restrict__ rn, const h8 *__restrict__ qfg,
    const float *__restrict__ thrg, int qtiles, long long ntiles, long long stride,
    float *__restrict__ umin, int m_padded, u64 *__restrict__ rec, unsigned *__restrict__ counts,
    unsigned *__restrict__ ctl, unsigned slice, unsigned short *__restrict__ rec_rows,
    // running thresholds (X16 scans only; null: the thresholds stay what knn_thr_kernel made them): per query the margin and
    // floor of knn_thr_kernel, and the lowest threshold any block of this launch has derived so far (ordered uints)
    const float *__restrict__ margg = nullptr, const float *__restrict__ floorg = nullptr, unsigned *__restrict__ rung = nullptr)
{
    constexpr int WAVES = BLOCK / 64;   // waves that share every staged reference tile
    constexpr bool RUN = X16 && !SAMPLE;   // thresholds tighten during the launch (k > 64: the 16 x 16 x 32 scans)
    constexpr int CHUNKS = KT * 64;                 // 16-byte chunks of A per tile
    constexpr int CPT = (CHUNKS + BLOCK - 1) / BLOCK;
    // TPB > 1 (round 3): TPB reference tiles per barrier, staged by LDS-DMA (global_load_lds, no staging registers) — the
    // per-tile barrier coupled the block's four waves, each of which shares its SIMD with a wave of another block, and
    // waves spent 38 % of their cycles parked at it (profiles/r02_c5_variants.txt); with TPB tiles between two barriers a
    // wave that is held up has TPB x 32 MFMAs of slack before the others wait for it
    // (TWO arrays per operand, not one [2][..]: the compiler orders every LDS read behind every LDS-DMA in flight that
    // may write what it reads, and with the buffer chosen by a run-time index that was every read — a vmcnt(0) right
    // behind the request for the next group, the copy never overlapped the scoring of the wave that asked for it.  Two
    // objects and a loop unrolled by two: reads of one array, requests into the other.)
    // (declared where they are used: s_a0 / s_a1 / s_n0 / s_n1 in the TPB > 1 branch, s_a[2] / s_n[2] in the other)
    if (!SAMPLE && ctl[KNN_CTL_FALLBACK] != 0u)
        return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qt0 = (blockIdx.y * WAVES + wib) * QT;   // this wave's first query tile
    const int nq = max(0, min(QT, qtiles - qt0));      // wave-uniform; 0 = padding wave

    const long long ns = (ntiles + stride - 1) / stride;
    const long long i0 = ns * blockIdx.x / gridDim.x;
    const long long i1 = ns * (blockIdx.x + 1) / gridDim.x;

    // X16: entry [t][2 j + ch] = B operand of K-step pair j for the 16 queries 16 ch .. of query tile t (lane l: query
    // 16 ch + (l & 15), dimensions 32 j + 8 (l >> 4) ..: the stored fragment of K-step 2 j + (l >> 5), half (l >> 4) & 1);
    // th / um [t][ch] in th[2 t + ch]
    static_assert(!X16 || KT % 2 == 0, "two K-steps per 16x16x32 instruction");
    h8 qf[QT][KT];
    float th[QT], um[X16 ? 2 * QT : QT];
    __shared__ float s_th16[X16 ? WAVES : 1][X16 ? QT * 32 : 1];
    __shared__ float s_mg16[RUN ? WAVES : 1][RUN ? QT * 32 : 1], s_fl16[RUN ? WAVES : 1][RUN ? QT * 32 : 1];
    const bool run = RUN && rung != nullptr;   // (kernel-uniform)
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int tt = min(qt0 + min(t, max(nq - 1, 0)), qtiles - 1);
        if constexpr (X16) {
#pragma unroll
            for (int j = 0; j < KT / 2; ++j)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch)
                    qf[t][2 * j + ch] = qfg[((size_t)tt * KT + 2 * j + (lane >> 5)) * 64 + ((lane >> 4) & 1) * 32 + 16 * ch + (lane & 15)];
            // (the 8 thresholds of a lane live in LDS, not in registers: with them the kernel needed 13 registers more than
            // the 256 two waves per SIMD leave each other, and the spill code sat inside the tile loop)
            if (lane < 32)
                s_th16[wib][t * 32 + lane] = (!SAMPLE && t < nq) ? thrg[(size_t)(qt0 + t) * 32 + lane] : -INFINITY;
            if constexpr (RUN)
                if (run && lane < 32) {
                    s_mg16[wib][t * 32 + lane] = t < nq ? margg[(size_t)(qt0 + t) * 32 + lane] : INFINITY;
                    s_fl16[wib][t * 32 + lane] = t < nq ? floorg[(size_t)(qt0 + t) * 32 + lane] : -INFINITY;
                }
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
                um[2 * t + ch] = INFINITY;
            th[t] = 0.0f;   // (unused in this form)
        } else {
#pragma unroll
            for (int kk = 0; kk < KT; ++kk)
                qf[t][kk] = qfg[((size_t)tt * KT + kk) * 64 + lane];
            th[t] = (!SAMPLE && t < nq) ? thrg[(size_t)(qt0 + t) * 32 + (lane & 31)] : -INFINITY;
            um[t] = INFINITY;
        }
    }
    const size_t list = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * WAVES + wib;
    u64 *__restrict__ my_rec = SAMPLE ? nullptr : rec + list * slice;
    // which of the lane's 16 rows are under the threshold: the re-rank reads only those (at k = 128 a
    // record's 16 rows are 8 KiB; without the mask the re-rank of C5's 640k records cost as much as the scan)
    unsigned short *__restrict__ my_rows = SAMPLE ? nullptr : rec_rows + list * slice;
    unsigned cnt = 0u;

    // one reference tile (fragments at a_lds, norm tile at n_lds) against this wave's QT query tiles
    auto score_tile = [&](const h8 *a_lds, const f4v *n_lds, long long i) __attribute__((always_inline)) {
        if constexpr (X16) {
            // lane l of a 16 x 16 sub-tile holds rows 4 (l >> 4) .. + 3 of its 16 references for query l & 15: the norms of
            // rows 16 rh + 4 (l >> 4) .. start the accumulators
            const int g = lane >> 4;
            f4v cn[2];
            cn[0] = n_lds[g];
            cn[1] = n_lds[4 + g];
            f4v d[QT][2][2];   // [t][ch][rh]
#pragma unroll
            for (int j = 0; j < KT / 2; ++j)
#pragma unroll
                for (int rh = 0; rh < 2; ++rh) {
                    // A operand: reference 16 rh + (l & 15), dimensions 32 j + 8 (l >> 4) ..
                    const h8 a = a_lds[(2 * j + (lane >> 5)) * 64 + ((lane >> 4) & 1) * 32 + 16 * rh + (lane & 15)];
#pragma unroll
                    for (int t = 0; t < QT; ++t)
#pragma unroll
                        for (int ch = 0; ch < 2; ++ch)
                            d[t][ch][rh] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qf[t][2 * j + ch], j == 0 ? cn[rh] : d[t][ch][rh], 0, 0, 0);
                }
            // (no MFMA may be scheduled down among the trees: with a tree's branch between an MFMA and the next block's reader
            // of its result the compiler left 6 wait states where the 4-pass shape needs 8 — tools/mfma_hazard_audit.py, the
            // cross-block miscount of DESIGN 4.2 again)
            __builtin_amdgcn_sched_barrier(0);
            // the lane's 8 thresholds, requested together now that the operand registers are free (one LDS round trip for
            // the eight trees, not one each)
            float thv[2 * QT];
            if (!SAMPLE) {
#pragma unroll
                for (int e = 0; e < 2 * QT; ++e)
                    thv[e] = s_th16[wib][(e >> 1) * 32 + 16 * (e & 1) + (lane & 15)];
#pragma unroll
                for (int e = 0; e < 2 * QT; ++e)
                    asm volatile("" : "+v"(thv[e]));   // (keeps the reads here: left alone they sank in front of each tree)
            }
            // (round 4, measured and not kept: ONE branch per tile over the eight trees' results, the records worked out again
            // in a slow path — C5 0.745 -> 0.925 ms.  Hits are not rare at C5: a dozen candidates per query leave a hit in
            // two of three (tile, 128 queries) pairs, and the slow path then paid for all eight trees' masks.)
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const f4v &x0 = d[t][ch][0], &x1 = d[t][ch][1];
                    const float m0 = min3f(x0[0], x0[1], x0[2]);
                    const float m1 = min3f(x0[3], x1[0], x1[1]);
                    const float m2 = min3f(x1[2], x1[3], m0);
                    if (SAMPLE) {
                        um[2 * t + ch] = min3f(m1, m2, um[2 * t + ch]);
                    } else {
                        const float thq = thv[2 * t + ch];
                        const float mn = RUN ? fminf(m1, m2) : min3f(m1, m2, thq);   // (RUN: the lowest score itself, a real row's)
                        const bool hit = mn < thq;
                        if (__builtin_expect(__ballot(hit) != 0ull, 0)) {
                            if constexpr (RUN) {
                                // a score below the threshold is also a new bound for its query (knn_thr_kernel): the four
                                // lanes that share the query may each write one — any of them is a valid threshold
                                if (run && hit) {
                                    // (indices worked out HERE, behind an asm barrier: hoisted out of the tile loop the eight
                                    // LDS offsets and eight 64-bit addresses were 28 bytes of scratch at 256 registers)
                                    unsigned ql = (unsigned)(t * 32 + 16 * ch) + ((unsigned)lane & 15u);
                                    asm volatile("" : "+v"(ql));
                                    const float tn = fmaxf(mn + s_mg16[wib][ql], s_fl16[wib][ql]);
                                    if (tn < thq) {
                                        s_th16[wib][ql] = tn;
                                        atomicMin(&rung[(size_t)qt0 * 32 + ql], f2ord(tn));
                                    }
                                }
                            }
                            // the record format of the 32 x 32 scan: (query, tile, half) + a mask over rows
                            // 8 (reg >> 2) + 4 half + (reg & 3).  This lane's rows 16 rh + 4 g + r are half = g & 1,
                            // reg = 8 rh + 4 (g >> 1) + r; lanes l and l ^ 32 (groups g and g ^ 2) hold the other eight rows of
                            // the SAME (query, half): their masks are merged and the lower lane writes ONE record — as many
                            // records as the 32 x 32 form leaves (two per hit filled the slices twice as fast: a far-away
                            // query, whose threshold lets most rows through, tipped a batch into the exact scan)
                            unsigned rm = 0u;
                            if (hit) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    rm |= x0[r] < thq ? (1u << (4 * (g >> 1) + r)) : 0u;
                                    rm |= x1[r] < thq ? (1u << (8 + 4 * (g >> 1) + r)) : 0u;
                                }
                            }
                            rm |= (unsigned)__shfl_xor((int)rm, 32, KNN_WAVE);
                            const bool emit = lane < 32 && rm != 0u;
                            const u64 mask = __ballot(emit);
                            if (emit) {
                                const unsigned pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                                     __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                                if (pos < slice) {
                                    unsigned qv = (unsigned)((qt0 + t) * 32 + 16 * ch) + ((unsigned)lane & 15u);
                                    asm volatile("" : "+v"(qv));   // (worked out HERE, in the rare branch: hoisted out of the
                                                                   // tile loop the eight of them were spilled to scratch)
                                    my_rec[pos] = ((u64)qv << 32) | ((u64)(i * stride) << 1) | (u64)(g & 1);
                                    my_rows[pos] = (unsigned short)rm;
                                }
                            }
                            cnt += (unsigned)__popcll(mask);
                        }
                    }
                }
            return;
        }
        f16v c;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f4v v = n_lds[2 * g + (lane >> 5)];
            c[4 * g + 0] = v[0];
            c[4 * g + 1] = v[1];
            c[4 * g + 2] = v[2];
            c[4 * g + 3] = v[3];
        }
        f16v d[QT];
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            const h8 a = a_lds[kk * 64 + lane];
#pragma unroll
            for (int t = 0; t < QT; ++t)
                d[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[t][kk], kk == 0 ? c : d[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const f16v &x = d[t];
            const float m0 = min3f(x[0], x[1], x[2]);
            const float m1 = min3f(x[3], x[4], x[5]);
            const float m2 = min3f(x[6], x[7], x[8]);
            const float m3 = min3f(x[9], x[10], x[11]);
            const float m4 = min3f(x[12], x[13], x[14]);
            const float m5 = min3f(m0, m1, m2);
            const float m6 = min3f(m3, m4, x[15]);
            if (SAMPLE) {
                um[t] = min3f(m5, m6, um[t]);
            } else {
                const float mn = min3f(m5, m6, th[t]);
                const bool hit = mn < th[t];
                const u64 mask = __ballot(hit);
                if (__builtin_expect(mask != 0ull, 0)) {
                    if (hit) {
                        const unsigned pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                             __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                        if (pos < slice) {
                            my_rec[pos] = ((u64)(unsigned)((qt0 + t) * 32 + (lane & 31)) << 32) |
                                          ((u64)(i * stride) << 1) | (u64)(lane >> 5);
                            unsigned rm = 0u;
#pragma unroll
                            for (int r16 = 0; r16 < 16; ++r16)
                                rm |= x[r16] < th[t] ? (1u << r16) : 0u;
                            my_rows[pos] = (unsigned short)rm;
                        }
                    }
                    cnt += (unsigned)__popcll(mask);
                }
            }
        }
    };
    if constexpr (TPB > 1) {
        __shared__ h8 s_a0[TPB * CHUNKS], s_a1[TPB * CHUNKS];
        __shared__ f4v s_n0[TPB * 8], s_n1[TPB * 8];
        static_assert(CHUNKS % 64 == 0 && (CHUNKS / 64) % WAVES == 0, "a tile's 1 KiB pieces are dealt out evenly to the waves");
        constexpr int PPW = CHUNKS / 64 / WAVES;   // 1 KiB pieces of a tile each wave requests
        // LDS-DMA: lane l of the wave copies 16 bytes from ITS global address to (wave-uniform LDS base) + 16 l
        auto issue = [&](h8 *sa, f4v *sn, long long j0) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < TPB; ++u) {
                const long long i = j0 + u;
                if (i < i1) {   // block-uniform
                    const long long tile = i * stride;
#pragma unroll
                    for (int pp = 0; pp < PPW; ++pp) {
                        const int piece = wib * PPW + pp;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rf + (size_t)tile * CHUNKS + piece * 64 + lane),
                                                         (__attribute__((address_space(3))) void *)&sa[u * CHUNKS + piece * 64], 16, 0, 0);
                    }
                    if (wib == 0 && lane < 8)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rn + (size_t)tile * 32 + 4 * lane),
                                                         (__attribute__((address_space(3))) void *)&sn[u * 8], 16, 0, 0);
                }
            }
        };
        // one group of TPB tiles: scored out of (ra, rn_) while the next group is on its way into (wa, wn)
        auto group = [&](const h8 *ra, const f4v *rn_, h8 *wa, f4v *wn, long long j0) __attribute__((always_inline)) {
            if (j0 + TPB < i1)
                issue(wa, wn, j0 + TPB);
#pragma unroll
            for (int u = 0; u < TPB; ++u)
                if (j0 + u < i1)   // block-uniform
                    score_tile(&ra[u * CHUNKS], &rn_[u * 8], j0 + u);
            if constexpr (RUN) {
                // what the other blocks that scan these queries (other tile ranges) have found meanwhile: requested in front of
                // the barrier — it waits for the next tiles anyway, and the scoring registers are dead here — and folded into this
                // wave's thresholds behind it.  Agent scope: the words are changed by other XCDs' atomics.
                static_assert(!RUN || QT * 32 <= 128, "at most two words per lane");
                unsigned o0 = 0xFFFFFFFFu, o1 = 0xFFFFFFFFu;
                if (run && nq > 0) {
                    const size_t qb = (size_t)qt0 * 32;
                    const size_t last = (size_t)(qt0 + nq) * 32 - 1;
                    o0 = __hip_atomic_load(&rung[min(qb + (size_t)lane, last)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    o1 = __hip_atomic_load(&rung[min(qb + 64 + (size_t)lane, last)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                if (run && nq > 0) {
                    if (lane < nq * 32)
                        s_th16[wib][lane] = fminf(s_th16[wib][lane], ord2f(o0));
                    if (64 + lane < nq * 32)
                        s_th16[wib][64 + lane] = fminf(s_th16[wib][64 + lane], ord2f(o1));
                }
            } else
            __syncthreads();   // the next group has landed (hipcc drains vmcnt in front of the barrier); nobody reads this one any more
        };
        if (i0 < i1) {
            issue(s_a0, s_n0, i0);
            __syncthreads();
            for (long long j0 = i0; j0 < i1; j0 += 2 * TPB) {
                group(s_a0, s_n0, s_a1, s_n1, j0);
                if (j0 + TPB < i1)   // block-uniform
                    group(s_a1, s_n1, s_a0, s_n0, j0 + TPB);
            }
        }
    } else if (i0 < i1) {
        __shared__ h8 s_a[2][CHUNKS];
        __shared__ f4v s_n[2][8];
        h8 stage_a[CPT];
        f4v stage_n = {0.f, 0.f, 0.f, 0.f};
        auto fetch = [&](long long i) {
            const long long tile = i * stride;
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int c = tid + j * BLOCK;
                if (c < CHUNKS)
                    stage_a[j] = rf[(size_t)tile * CHUNKS + c];
            }
            if (tid < 8)
                stage_n = *(const f4v *)(rn + (size_t)tile * 32 + 4 * tid);
        };
        auto park = [&](int buf) {
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int c = tid + j * BLOCK;
                if (c < CHUNKS)
                    s_a[buf][c] = stage_a[j];
            }
            if (tid < 8)
                s_n[buf][tid] = stage_n;
        };
        fetch(i0);
        park(0);
        __syncthreads();
        for (long long i = i0; i < i1; ++i) {
            const int buf = (int)((i - i0) & 1);
            if (i + 1 < i1)
                fetch(i + 1);  // global loads in flight while this tile is consumed
            score_tile(&s_a[buf][0], &s_n[buf][0], i);
            if (i + 1 < i1)
                park(buf ^ 1);  // the other buffer was last read one iteration ago
            if constexpr (RUN) {
                // (running thresholds, as in the four-tiles-per-barrier form above: what the other blocks have found, every
                // fourth tile, requested in front of the barrier and folded in behind it; QT x 32 <= 64 queries: one word per lane)
                static_assert(!RUN || TPB > 1 || QT * 32 <= 64, "one word per lane");
                const bool reload = run && nq > 0 && ((i - i0) & 3) == 3;   // (block-uniform but for nq: no barrier inside)
                unsigned o0 = 0xFFFFFFFFu;
                if (reload)
                    o0 = __hip_atomic_load(&rung[min((size_t)qt0 * 32 + (size_t)lane, (size_t)(qt0 + nq) * 32 - 1)], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                if (reload && lane < nq * 32)
                    s_th16[wib][lane] = fminf(s_th16[wib][lane], ord2f(o0));
            } else
            __syncthreads();
        }
    }
    if (SAMPLE) {
        if constexpr (X16) {
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {   // a query's column sits on the four lanes l & 15, one per group of rows
                    float v = um[2 * t + ch];
                    v = fminf(v, __shfl_xor(v, 16, KNN_WAVE));
                    v = fminf(v, __shfl_xor(v, 32, KNN_WAVE));
                    if (lane < 16 && t < nq)
                        umin[(size_t)blockIdx.x * m_padded + (size_t)(qt0 + t) * 32 + 16 * ch + lane] = v;
                }
        } else {
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const float v = fminf(um[t], __shfl_xor(um[t], 32, KNN_WAVE));
                if (lane < 32 && t < nq)
                    umin[(size_t)blockIdx.x * m_padded + (size_t)(qt0 + t) * 32 + lane] = v;
            }
        }
    } else if (lane == 0) {
        counts[list] = cnt;
        if (cnt > slice)
            ctl[KNN_CTL_FALLBACK] = 1u;
    }
}

// ------------------------------------------------------------------------------------------
// Any k beyond 512 (round 3).  The B operands of even one block of 32 queries no longer fit a wave's registers, so K is
// walked in CHUNKS of KC = 8 K-steps (128 dimensions; the layouts pad k to a multiple of 128): a wave owns QT = 2 blocks of
// queries and keeps the accumulators of T = 4 reference tiles x QT alive (128 registers) while the chunks go by; per
// chunk it fetches its B fragments (64 registers, from L2: 16 KiB per 64 MFMAs) and the block's four waves share the
// T tiles' A fragments of that chunk through LDS (32 KiB, LDS-DMA, double-buffered: one barrier per chunk, every
// ds_read_b128 feeds two MFMAs).  After the last chunk the T x QT accumulators go through the same min3 tree /
// threshold / record code as knn_filter_tiled_kernel (records carry row masks).  SAMPLE as there.
//   grid.x blocks split the groups of T (sampled) reference tiles, grid.y = groups of 4 * QT query tiles.
// ------------------------------------------------------------------------------------------
#define CHK_KC 8
#define CHK_T 4
#define CHK_QT 2
template <bool SAMPLE>
__global__ __launch_bounds__(FILTER_BLOCK, 2) void knn_filter_chunked_kernel(
    const h8 *__restrict__ rf, const float *__restrict__ rn, const h8 *__restrict__ qfg,
    const float *__restrict__ thrg, int kt, int qtiles, long long ntiles, long long stride,
    float *__restrict__ umin, int m_padded, u64 *__restrict__ rec, unsigned *__restrict__ counts,
    unsigned *__restrict__ ctl, unsigned slice, unsigned short *__restrict__ rec_rows,
    // != 0: a ONE-dimensional grid of ranges x (query groups rounded up to eight) blocks, see below
    unsigned ranges, unsigned qgroups)
{
    constexpr int WAVES = FILTER_BLOCK / 64;
    constexpr int PIECES = CHK_T * CHK_KC;          // 1 KiB pieces of A per stage
    static_assert(PIECES % WAVES == 0, "a stage's pieces are dealt out evenly to the waves");
    static_assert(CHK_KC % 2 == 0, "two K-steps per 16x16x32 instruction");
    // (two arrays per operand and the stage loop unrolled by two, as in knn_filter_tiled_kernel: a read of the array a
    // request in flight may write waits for that request)
    __shared__ h8 s_a0[PIECES * 64], s_a1[PIECES * 64];
    static_assert(CHK_T == WAVES, "wave w requests the norms of the stage's tile w");
    __shared__ f4v s_n0[CHK_T * 64], s_n1[CHK_T * 64];   // [tile][64]: a whole wave's LDS-DMA (entries 8.. of a tile are copies)
    if (!SAMPLE && ctl[KNN_CTL_FALLBACK] != 0u)
        return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Which (range of reference tiles, group of 256 queries) a block scores.  Round 4: workgroups go to the XCDs round
    // robin by their linear number, and with the plain (x = range, y = query group) grid of eight ranges an XCD's 64
    // resident blocks were 64 DIFFERENT query groups on one range — 32 MB of B fragments cycling through a 4 MB L2, every
    // block fetching its 512 KiB of B again for each group of tiles: 68 GB per launch at (1024, 65536, 65536), which
    // ran at 0.31 of the MFMA peak where (1024, 16384, 65536) ran at 0.49.  So the launch is one-dimensional and block L
    // is the (L / 8)-th block of XCD L % 8, which walks ITS query groups (g ≡ XCD mod 8) one after the other, all `ranges`
    // ranges of a query group side by side: 64 / ranges query groups' B (2 MB at 16 ranges) stay in the L2, each range's A
    // is shared by those few blocks, and what comes from beyond the L2 is every operand about once per XCD pass.
    unsigned bx = blockIdx.x, by = blockIdx.y, nbx = gridDim.x;
    if (!SAMPLE && ranges != 0u) {   // (the sample pass keeps the plain grid: it is launched that way, and its registers are all taken)
        const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
        bx = j % ranges;
        by = (j / ranges) * 8u + xcd;
        nbx = ranges;
        if (by >= qgroups)
            return;   // (the grid is rounded up to eight query groups; block-uniform, before any barrier)
    }
    const int qt0 = (int)(by * WAVES + wib) * CHK_QT;   // this wave's first query tile
    const int nq = max(0, min(CHK_QT, qtiles - qt0));      // wave-uniform; 0 = padding wave
    const int nchunks = kt / CHK_KC;

    const long long ns = (ntiles + stride - 1) / stride;            // (sampled) tiles
    const long long ng = (ns + CHK_T - 1) / CHK_T;                  // groups of T
    const long long g0 = ng * bx / nbx, g1 = ng * (bx + 1) / nbx;

    // The 16 x 16 x 32 MFMA shape on the layouts of the 32 x 32 x 16 one (round 4, as knn_filter_tiled_kernel's X16: the chip
    // holds 1.96 GHz on the 4-pass shape and 1.70 on the 8-pass one, profiles/r04_mfma_shape_probe.txt).  Lane l of a
    // 16 x 16 sub-tile: query 16 ch + (l & 15) of query tile t, rows 4 (l >> 4) .. + 3 of the 16 references 16 rh ..;
    // um [t][ch] in um[2 t + ch]; the thresholds are fetched in the epilogue, once per K sweep (four registers the
    // chunk loop does not have: with them the scan kernel spilled 28 bytes).
    float um[2 * CHK_QT];
    int qtile[CHK_QT];
#pragma unroll
    for (int t = 0; t < CHK_QT; ++t) {
        qtile[t] = min(qt0 + min(t, max(nq - 1, 0)), qtiles - 1);
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
            um[2 * t + ch] = INFINITY;
    }
    const size_t list = ((size_t)by * nbx + bx) * WAVES + wib;
    u64 *__restrict__ my_rec = SAMPLE ? nullptr : rec + list * slice;
    unsigned short *__restrict__ my_rows = SAMPLE ? nullptr : rec_rows + list * slice;
    unsigned cnt = 0u;
    const int g4 = lane >> 4;   // which four rows of a 16 x 16 sub-tile this lane holds
    // B operand of K-step pair j of a chunk for the 16 queries 16 ch .. of a query tile: lane l takes dimensions
    // 32 j + 8 (l >> 4) .., i.e. the stored fragment of K-step 2 j + (l >> 5), half (l >> 4) & 1
    const int b_lane = ((lane >> 4) & 1) * 32 + (lane & 15);
    auto b_at = [&](int t, int chunk, int j, int ch) __attribute__((always_inline)) {
        return qfg[((size_t)qtile[t] * kt + (size_t)chunk * CHK_KC + 2 * j + (lane >> 5)) * 64 + b_lane + 16 * ch];
    };

    // stage = (group, chunk): the A fragments of the group's T tiles for K-steps [8 chunk, 8 chunk + 8), and the tiles'
    // norms.  Tiles past the end are clamped to the last one (their scores are dropped in the epilogue).
    auto issue = [&](h8 *wa, f4v *wn, long long g, int chunk) __attribute__((always_inline)) {
#pragma unroll
        for (int pp = 0; pp < PIECES / WAVES; ++pp) {
            const int piece = wib * (PIECES / WAVES) + pp;      // = tt * KC + kk
            const int tt = piece / CHK_KC, kk = piece % CHK_KC;
            const long long tile = min((g * CHK_T + tt) * stride, ntiles - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rf + ((size_t)tile * kt + (size_t)chunk * CHK_KC + kk) * 64 + lane),
                                             (__attribute__((address_space(3))) void *)&wa[piece * 64], 16, 0, 0);
        }
        // the tiles' norms with every stage, by every wave, all lanes (no branch around a request: the compiler counts what
        // is in flight, and a request that may or may not have been made turns every later wait into a wait for everything)
        {
            const long long tile = min((g * CHK_T + wib) * stride, ntiles - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rn + (size_t)tile * 32 + 4 * (lane & 7)),
                                             (__attribute__((address_space(3))) void *)&wn[wib * 64], 16, 0, 0);
        }
    };
    if (g0 < g1) {
        f4v acc[CHK_T][CHK_QT][2][2];          // [tile][query tile][ch][rh]
        h8 qf[CHK_QT][CHK_KC];                 // [t][2 j + ch]: the B fragments of the stage being scored (loaded one stage ahead)
        // one stage: scored out of (ra, rn_) while the next one is on its way into (wa, wn)
        auto stage = [&](const h8 *ra, const f4v *rn_, h8 *wa, f4v *wn, long long g, int chunk) __attribute__((always_inline)) {
            // the next stage (behind the last one: that one again — requests nobody reads, but no branch)
            const bool wrap = chunk + 1 == nchunks;
            const long long gn = wrap ? min(g + 1, g1 - 1) : g;
            const int cn = wrap ? (g + 1 < g1 ? 0 : chunk) : chunk + 1;
            issue(wa, wn, gn, cn);
            if (chunk == 0) {   // the norms of the lane's rows start the accumulators
#pragma unroll
                for (int tt = 0; tt < CHK_T; ++tt)
#pragma unroll
                    for (int rh = 0; rh < 2; ++rh) {
                        const f4v c = rn_[tt * 64 + 4 * rh + g4];
#pragma unroll
                        for (int t = 0; t < CHK_QT; ++t)
#pragma unroll
                            for (int ch = 0; ch < 2; ++ch)
                                acc[tt][t][ch][rh] = c;
                    }
            }
            // K-step pair by K-step pair over the T tiles: the B fragments of a pair are dead once its MFMAs are issued, and
            // the NEXT stage's fragments of that pair are requested into the same registers right there (round 4; round 3
            // requested a stage's 16 KiB of B at its start and waited for them in front of its first MFMA).  The barrier at
            // the end of the stage drains them with the A requests; inside a stage no MFMA waits for memory.
#pragma unroll
            for (int j = 0; j < CHK_KC / 2; ++j) {
#pragma unroll
                for (int tt = 0; tt < CHK_T; ++tt)
#pragma unroll
                    for (int rh = 0; rh < 2; ++rh) {
                        // A operand: reference 16 rh + (l & 15) of tile tt, dimensions 32 j + 8 (l >> 4) ..
                        const h8 a = ra[(tt * CHK_KC + 2 * j + (lane >> 5)) * 64 + ((lane >> 4) & 1) * 32 + 16 * rh + (lane & 15)];
#pragma unroll
                        for (int t = 0; t < CHK_QT; ++t)
#pragma unroll
                            for (int ch = 0; ch < 2; ++ch)
                                acc[tt][t][ch][rh] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qf[t][2 * j + ch], acc[tt][t][ch][rh], 0, 0, 0);
                    }
#pragma unroll
                for (int t = 0; t < CHK_QT; ++t)
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch)
                        qf[t][2 * j + ch] = b_at(t, cn, j, ch);
            }
            if (chunk + 1 == nchunks) {
                // (no MFMA may be scheduled down among the trees: see knn_filter_tiled_kernel)
                __builtin_amdgcn_sched_barrier(0);
                // everything the epilogue derives from the lane's number is worked out HERE, from a value the compiler cannot
                // see through: as loop invariants the thresholds' address, the row-mask shifts and the shuffle's lane went to
                // scratch across the stage loop (44 bytes — and a kernel that touches scratch at all pays for it on every launch)
                unsigned le = threadIdx.x;
                asm volatile("" : "+v"(le));
                const unsigned l15 = le & 15u, ge = (le >> 4) & 3u;
                float th[2 * CHK_QT];
#pragma unroll
                for (int t = 0; t < CHK_QT; ++t)
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch)
                        th[2 * t + ch] = (!SAMPLE && t < nq) ? thrg[(size_t)(qt0 + t) * 32 + 16 * ch + l15] : -INFINITY;
#pragma unroll
                for (int tt = 0; tt < CHK_T; ++tt) {
                    const long long i = g * CHK_T + tt;   // (sampled) tile number
                    if (i >= ns)
                        continue;   // block-uniform: a clamped copy of the last tile
#pragma unroll
                    for (int t = 0; t < CHK_QT; ++t)
#pragma unroll
                        for (int ch = 0; ch < 2; ++ch) {
                            const f4v &x0 = acc[tt][t][ch][0], &x1 = acc[tt][t][ch][1];
                            const float m0 = min3f(x0[0], x0[1], x0[2]);
                            const float m1 = min3f(x0[3], x1[0], x1[1]);
                            const float m2 = min3f(x1[2], x1[3], m0);
                            if (SAMPLE) {
                                um[2 * t + ch] = min3f(m1, m2, um[2 * t + ch]);
                            } else {
                                const float thq = th[2 * t + ch];
                                const float mn = min3f(m1, m2, thq);
                                const bool hit = mn < thq;
                                if (__builtin_expect(__ballot(hit) != 0ull, 0)) {
                                    // the record format of the 32 x 32 scan, one record per (query, half): see the X16
                                    // epilogue of knn_filter_tiled_kernel
                                    unsigned rm = 0u;
                                    if (hit) {
                                        const unsigned sh = 4u * (ge >> 1);
#pragma unroll
                                        for (int r = 0; r < 4; ++r) {
                                            rm |= x0[r] < thq ? (1u << (sh + r)) : 0u;
                                            rm |= x1[r] < thq ? (256u << (sh + r)) : 0u;
                                        }
                                    }
                                    rm |= (unsigned)__builtin_amdgcn_ds_bpermute((int)(((le ^ 32u) & 63u) << 2), (int)rm);   // lane l ^ 32's rows
                                    const bool emit = (le & 32u) == 0u && rm != 0u;
                                    const u64 mask = __ballot(emit);
                                    if (emit) {
                                        const unsigned pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                                             __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                                        if (pos < slice) {
                                            const unsigned qv = (unsigned)((qt0 + t) * 32 + 16 * ch) + l15;
                                            my_rec[pos] = ((u64)qv << 32) | ((u64)(i * stride) << 1) | (u64)(ge & 1u);
                                            my_rows[pos] = (unsigned short)rm;
                                        }
                                    }
                                    cnt += (unsigned)__popcll(mask);
                                }
                            }
                        }
                }
            }
            __syncthreads();   // the next stage has landed; nobody reads this one any more
        };
        issue(s_a0, s_n0, g0, 0);
#pragma unroll
        for (int t = 0; t < CHK_QT; ++t)
#pragma unroll
            for (int j = 0; j < CHK_KC / 2; ++j)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch)
                    qf[t][2 * j + ch] = b_at(t, 0, j, ch);
        __syncthreads();   // (waits for the DMA: hipcc drains vmcnt in front of the barrier)
        long long g = g0;
        int chunk = 0;
        for (;;) {   // stages (g, chunk) in order, alternating arrays
            stage(s_a0, s_n0, s_a1, s_n1, g, chunk);
            if (++chunk == nchunks) {
                chunk = 0;
                if (++g >= g1)
                    break;
            }
            stage(s_a1, s_n1, s_a0, s_n0, g, chunk);
            if (++chunk == nchunks) {
                chunk = 0;
                if (++g >= g1)
                    break;
            }
        }
    }
    if (SAMPLE) {
#pragma unroll
        for (int t = 0; t < CHK_QT; ++t)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {   // a query's column sits on the four lanes l & 15, one per group of rows
                float v = um[2 * t + ch];
                v = fminf(v, __shfl_xor(v, 16, KNN_WAVE));
                v = fminf(v, __shfl_xor(v, 32, KNN_WAVE));
                if (lane < 16 && t < nq)
                    umin[(size_t)bx * m_padded + (size_t)(qt0 + t) * 32 + 16 * ch + lane] = v;
            }
    } else if (lane == 0) {
        counts[list] = cnt;
        if (cnt > slice)
            ctl[KNN_CTL_FALLBACK] = 1u;
    }
}

// Sample pass: the same MFMA stream over every `stride`-th reference tile, keeping only the
// running minimum score per query (no thresholds, no branches).  It replaces an exact pre-pass:
// the minimum is a score of a real reference, which is all knn_threshold needs.  Per-block
// minima go to umin[blockIdx.x][query]; the threshold kernel folds the blocks.
template <int KT, int QT>
__global__ __launch_bounds__(FILTER_BLOCK, 2) void knn_filter_sample_kernel(
    const h8 *__restrict__ rf, const float *__restrict__ rn, const h8 *__restrict__ qfg, int qtiles,
    long long ntiles, long long stride, float *__restrict__ umin, int m_padded,
    const unsigned *__restrict__ ctl, int qt_base)
{
    __shared__ float s_min[FILTER_BLOCK / 64][QT * 32];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int qt0 = qt_base + blockIdx.y * QT;
    const int nq = min(QT, qtiles - qt0);
    const long long wave = (long long)blockIdx.x * (FILTER_BLOCK / 64) + wib;
    const long long nwaves = (long long)gridDim.x * (FILTER_BLOCK / 64);
    const long long ns = (ntiles + stride - 1) / stride;  // sampled tiles: 0, stride, 2 stride, ...
    const long long i0 = ns * wave / nwaves;
    const long long i1 = ns * (wave + 1) / nwaves;

    h8 qf[QT][KT];
    float um[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int tt = min(t, nq - 1);
#pragma unroll
        for (int kk = 0; kk < KT; ++kk)
            qf[t][kk] = qfg[((size_t)(qt0 + tt) * KT + kk) * 64 + lane];
        um[t] = INFINITY;
    }
    if (i0 < i1) {
        h8 a[KT];
        f16v c;
        load_ref_tile<KT>(rf, rn, i0 * stride, lane, a, c);
        for (long long i = i0; i < i1; ++i) {
            h8 an[KT];
            f16v cn;
            load_ref_tile<KT>(rf, rn, min(i + 1, i1 - 1) * stride, lane, an, cn);
            f16v d[2];
            d[0] = c;
#pragma unroll
            for (int kk = 0; kk < KT; ++kk)
                d[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], qf[0][kk], d[0], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                if (t + 1 < QT) {
                    f16v &dn = d[(t + 1) & 1];
                    dn = c;
#pragma unroll
                    for (int kk = 0; kk < KT; ++kk)
                        dn = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], qf[t + 1][kk], dn, 0, 0, 0);
                }
                const f16v &x = d[t & 1];
                const float m0 = min3f(x[0], x[1], x[2]);
                const float m1 = min3f(x[3], x[4], x[5]);
                const float m2 = min3f(x[6], x[7], x[8]);
                const float m3 = min3f(x[9], x[10], x[11]);
                const float m4 = min3f(x[12], x[13], x[14]);
                const float m5 = min3f(m0, m1, m2);
                const float m6 = min3f(m3, m4, x[15]);
                um[t] = min3f(m5, m6, um[t]);
            }
#pragma unroll
            for (int kk = 0; kk < KT; ++kk)
                a[kk] = an[kk];
            c = cn;
        }
    }
    // lanes l and l+32 hold the same query column: fold, then fold the block's waves through LDS
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float v = fminf(um[t], __shfl_xor(um[t], 32, KNN_WAVE));
        if (lane < 32)
            s_min[wib][t * 32 + lane] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nq * 32; i += FILTER_BLOCK) {
        float v = s_min[0][i];
#pragma unroll
        for (int w = 1; w < FILTER_BLOCK / 64; ++w)
            v = fminf(v, s_min[w][i]);
        umin[(size_t)blockIdx.x * m_padded + (size_t)qt0 * 32 + i] = v;
    }
}

// Test hook: all scores of one (reference tile, query tile) pair per wave.
template <int KT>
__global__ __launch_bounds__(64) void knn_filter_scores_kernel(const h8 *__restrict__ rf,
                                                               const float *__restrict__ rn,
                                                               const h8 *__restrict__ qfg, int m,
                                                               long long n, float *__restrict__ scores)
{
    const int lane = threadIdx.x & 63;
    h8 a[KT];
    f16v c;
    load_ref_tile<KT>(rf, rn, blockIdx.x, lane, a, c);
    f16v d = c;
#pragma unroll
    for (int kk = 0; kk < KT; ++kk)
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], qfg[((size_t)blockIdx.y * KT + kk) * 64 + lane], d,
                                                   0, 0, 0);
    const long long q = (long long)blockIdx.y * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const long long r = (long long)blockIdx.x * 32 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
        if (q < m && r < n)
            scores[(size_t)q * n + r] = d[i];
    }
}

// The same for any kt (k > 512): K-steps in a loop, fragments straight from memory.
__global__ __launch_bounds__(64) void knn_filter_scores_rt_kernel(const h8 *__restrict__ rf, const float *__restrict__ rn,
                                                                  const h8 *__restrict__ qfg, int kt, int m, long long n,
                                                                  float *__restrict__ scores)
{
    const int lane = threadIdx.x & 63;
    f16v d;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const f4v v = *(const f4v *)(rn + (size_t)blockIdx.x * 32 + 8 * gq + 4 * (lane >> 5));
        d[4 * gq + 0] = v[0];
        d[4 * gq + 1] = v[1];
        d[4 * gq + 2] = v[2];
        d[4 * gq + 3] = v[3];
    }
    for (int kk = 0; kk < kt; ++kk)
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(rf[((size_t)blockIdx.x * kt + kk) * 64 + lane],
                                                   qfg[((size_t)blockIdx.y * kt + kk) * 64 + lane], d, 0, 0, 0);
    const long long q = (long long)blockIdx.y * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const long long r = (long long)blockIdx.x * 32 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
        if (q < m && r < n)
            scores[(size_t)q * n + r] = d[i];
    }
}

// ------------------------------------------------------------------------------------------
// Host side.
// ------------------------------------------------------------------------------------------

static const unsigned kRecordCapacity = KNN_RECORD_CAPACITY;  // 4M records = 32 MiB, split evenly over the waves
static const unsigned kMaxLists = KNN_MAX_LISTS;
static const unsigned kSampleBlocks = 512;           // most blocks the sample pass uses (x 4 waves)

void knn_filter_free(FilterState &st)
{
    knn_cells_free(st.cells);
    (void)KNN_DEV_FREE(st.center);
    (void)KNN_DEV_FREE(st.ref_frags);
    (void)KNN_DEV_FREE(st.ref_norms);
    (void)KNN_DEV_FREE(st.ref_norms2);
    (void)KNN_DEV_FREE(st.outliers);
    if (st.scan_done)
        (void)hipEventDestroy(st.scan_done);
    for (FilterWorkspace &w : st.ws) {
        (void)KNN_DEV_FREE(w.qry_frags);
        (void)KNN_DEV_FREE(w.qry_norms);
        (void)KNN_DEV_FREE(w.qry_amax);
        (void)KNN_DEV_FREE(w.thr);
        (void)KNN_DEV_FREE(w.ctl);
        (void)KNN_DEV_FREE(w.records);
        (void)KNN_DEV_FREE(w.counts);
        (void)KNN_DEV_FREE(w.umin);
        (void)KNN_DEV_FREE(w.qpart);
        knn_cells_workspace_free(w);
    }
    st = FilterState();
}

// Robust box of a sample (samples x k, row-major): per dimension [median - w s, median + w s] clipped to
// [lo_d, hi_d], s = 1.4826 * MAD (median/MAD do not move when a few rows sit 300 sigma out; mean/std do).
// A few far-out rows would otherwise stretch the box, and with it the fp16 step, for everybody.  ANY
// box is correct: rows outside it leave the filter and are scanned exactly on every query, so the box
// is only worth it if it leaves out a handful of rows — w doubles from 12 until at most 1 % of the
// sample falls outside (heavy tails), and a second mode further out than 96 s (more than 1 % of the
// rows) keeps the plain [lo, hi].  Returns the half-width of the widest dimension; center[16 kt].
static double robust_box(const std::vector<float> &samp, long long samples, int k, int kp, const std::vector<float> &lo_d,
                         const std::vector<float> &hi_d, std::vector<float> &center)
{
    center.assign((size_t)kp, 0.0f);
    std::vector<float> col((size_t)samples), blo((size_t)k), bhi((size_t)k);
    std::vector<double> med((size_t)k), mad((size_t)k);
    for (int d = 0; d < k; ++d) {
        for (long long i = 0; i < samples; ++i)
            col[(size_t)i] = samp[(size_t)i * k + d];
        std::nth_element(col.begin(), col.begin() + samples / 2, col.end());
        med[(size_t)d] = col[(size_t)(samples / 2)];
        for (long long i = 0; i < samples; ++i)
            col[(size_t)i] = fabsf(col[(size_t)i] - (float)med[(size_t)d]);
        std::nth_element(col.begin(), col.begin() + samples / 2, col.end());
        mad[(size_t)d] = 1.4826 * (double)col[(size_t)(samples / 2)];
    }
    bool clipped = false;
    for (double w = 12.0; w <= 96.0 && !clipped; w *= 2.0) {
        for (int d = 0; d < k; ++d) {
            float lo = lo_d[(size_t)d], hi = hi_d[(size_t)d];
            const double sr = mad[(size_t)d];
            if (sr > 0.0 && sr < 1e30) {
                const double rlo = med[(size_t)d] - w * sr, rhi = med[(size_t)d] + w * sr;
                if (rlo > (double)lo && rlo < (double)hi)
                    lo = (float)rlo;
                if (rhi < (double)hi && rhi > (double)lo)
                    hi = (float)rhi;
            }
            blo[(size_t)d] = lo;
            bhi[(size_t)d] = hi;
        }
        long long outside = 0;
        for (long long i = 0; i < samples; ++i) {
            bool out = false;
            for (int d = 0; d < k && !out; ++d) {
                const float x = samp[(size_t)i * k + d];
                out = x < blo[(size_t)d] || x > bhi[(size_t)d];
            }
            outside += out;
        }
        clipped = outside * 100 <= samples;
    }
    if (!clipped)
        for (int d = 0; d < k; ++d) {
            blo[(size_t)d] = lo_d[(size_t)d];
            bhi[(size_t)d] = hi_d[(size_t)d];
        }
    double h = 0.0;
    for (int d = 0; d < k; ++d) {
        const float lo = blo[(size_t)d], hi = bhi[(size_t)d];
        const float c = 0.5f * lo + 0.5f * hi;
        center[(size_t)d] = c;
        h = fmax(h, fmax((double)hi - (double)c, (double)c - (double)lo));
    }
    return h;
}

// Centre and power-of-two scale of a sample's robust box (what knn_filter_build_from_host derives from its host sample):
// false when the sample holds non-finite values or its box is degenerate.
static bool box_from_sample(const float *sample, long long samples, int k, int kp, std::vector<float> &center, float *sigma_out)
{
    const long long sub = samples >= 4096 ? 4 : 1, nsub = samples / sub;
    std::vector<float> samp((size_t)nsub * k), dlo((size_t)k, INFINITY), dhi((size_t)k, -INFINITY);
    for (long long i = 0; i < samples; ++i) {
        const float *x = sample + (size_t)i * k;
        for (int d = 0; d < k; ++d) {
            const float v = x[d];
            if (!(fabsf(v) < INFINITY))
                return false;
            if (i % sub == 0 && i / sub < nsub)
                samp[(size_t)(i / sub) * k + d] = v;
            dlo[(size_t)d] = fminf(dlo[(size_t)d], v);
            dhi[(size_t)d] = fmaxf(dhi[(size_t)d], v);
        }
    }
    for (int d = 0; d < k; ++d) {   // the sample's range, widened by 1 / 32 of its width
        const float pad = (dhi[(size_t)d] - dlo[(size_t)d]) * (1.0f / 32.0f);
        dlo[(size_t)d] -= pad;
        dhi[(size_t)d] += pad;
    }
    const double h = robust_box(samp, nsub, k, kp, dlo, dhi, center);
    if (!(h <= 1e15) || (h != 0.0 && h < 1e-15))
        return false;
    float sigma = 1.0f;
    if (h > 0.0) {
        int ex;
        (void)frexp(h, &ex);
        sigma = (float)ldexp(1.0, -ex);
    }
    *sigma_out = sigma;
    return true;
}

bool knn_geom_from_sample(ShardGeom &g, int k, long long n_global, int nranks, const float *sample, long long samples,
                          int seed_tiles)
{
    if (!sample || k < 1 || k > 16)
        return false;
    std::vector<float> center;
    float sigma = 1.0f;
    if (!box_from_sample(sample, samples, k, 16, center, &sigma))
        return false;
    if (!knn_geom_cells(g, k, n_global, nranks, sample, samples, seed_tiles))
        return false;
    for (int d = 0; d < 16; ++d)
        g.center[d] = d < k ? center[(size_t)d] : 0.0f;
    g.sigma = sigma;
    return true;
}

hipError_t knn_filter_build(FilterState &st, int k, long long n, const float *r, hipStream_t s, int want_cells,
                            const ShardGeom *geom, int rank, unsigned *bad_rows_out)
{
    st = FilterState();
    const int kt = knn_kt_of(k);
    if (n <= 0 || kt == 0)
        return hipSuccess;
    const bool trace = getenv("KNN_MI355X_TRACE_BUILD") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace)
            return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[knn build] %-24s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    // want_cells: 1 = the fast two-pass build, counted build if a bucket overflows; 2 = one-pass placement; 3 = the counted two-pass build
    bool fast_build = want_cells == 1, fast_overflow = false;
    // The cell-sorted layout from a GIVEN frame (centre, scale) and either a shard geometry's cuts or cuts taken from `samp`.
    // Returns with st.usable set when the layout stands; st.cells null (and st reset) when the shard does not suit the cells.
    // late_frame (nullable): called once the cells are counted; fills centre and scale then (false: no frame — give up)
    auto sorted_layout = [&](float *center16, float sigma, const std::vector<float> &samp, long long samples,
                             const std::function<bool(float *, float *)> &late_frame) -> hipError_t {
        long long ntiles = 0;
        unsigned *cell_code = nullptr, *cell_fill = nullptr;
        FTRY(knn_cells_build(&st.cells, k, n, r, samp, samples, s, &ntiles, &cell_code, &cell_fill, want_cells == 2 || kt == 2, geom, rank,
                             bad_rows_out, fast_build && kt == 1));
        lap(st.cells ? (st.cells->build_res ? "buckets + cell prefix (enqueued)" : "cell codes + counts") : "cell codes (not kept)");
        if (!st.cells)
            return hipSuccess;
        if (late_frame && !late_frame(center16, &sigma)) {
            (void)KNN_DEV_FREE(cell_code);
            (void)KNN_DEV_FREE(cell_fill);
            knn_filter_free(st);
            return hipSuccess;
        }
        st.k = k;
        st.kt = kt;   // (1, or 2 for 16 < k <= 32)
        st.n = n;
        st.ntiles = ntiles;
        st.sigma = sigma;
        unsigned *dout = nullptr;
        hipError_t e = KNN_DEV_ALLOC((void **)&st.center, (size_t)16 * kt * sizeof(float));
        if (e == hipSuccess)
            e = KNN_DEV_ALLOC(&st.ref_frags, (size_t)ntiles * kt * 64 * 16);
        if (e == hipSuccess)
            e = KNN_DEV_ALLOC((void **)&st.ref_norms, (size_t)ntiles * 32 * sizeof(float));
        if (e == hipSuccess)
            e = KNN_DEV_ALLOC((void **)&st.ref_norms2, (size_t)ntiles * 32 * sizeof(unsigned));
        const unsigned ocap = (unsigned)(n / 32 > 4096 ? n / 32 : 4096);
        if (e == hipSuccess)
            e = KNN_DEV_ALLOC((void **)&st.outliers, (size_t)ocap * sizeof(unsigned));
        if (e == hipSuccess)
            e = KNN_DEV_ALLOC((void **)&dout, 4 * sizeof(unsigned));
        if (e == hipSuccess)
            e = hipMemsetAsync(dout, 0, 4 * sizeof(unsigned), s);
        if (e == hipSuccess)
            e = hipMemcpyAsync(st.center, center16, (size_t)16 * kt * sizeof(float), hipMemcpyHostToDevice, s);
        lap("allocations");
        unsigned hout[4] = {0, 0, 0, 0}, hres[4] = {0, 0, 0, 0};
        const bool fast_built = st.cells->build_res != nullptr;
        if (e == hipSuccess)
            e = knn_cells_place_rows(st, r, cell_code, cell_fill, dout, ocap, s);
        if (e == hipSuccess)
            e = hipMemcpyAsync(hout, dout, sizeof hout, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && fast_built)
            e = hipMemcpyAsync(hres, st.cells->build_res, sizeof hres, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess)
            e = hipStreamSynchronize(s);   // (also keeps `center16` alive until the copy is done)
        lap(fast_built ? "build kernels + placement + sync" : "placement + sync");
        if (e == hipSuccess && fast_built) {
            // the fast build's verdict: tiles, items, the largest cell — or a bucket that outgrew its fixed room
            st.ntiles = hres[0];
            st.cells->nitems = hres[1];
            st.cells->max_cell_rows = hres[2];
            if (hres[3] != 0u || hres[1] == 0u)
                fast_overflow = true;
            if (trace)
                fprintf(stderr, "[knn build] fast build: %u tiles (room for %lld), %u items, largest cell %u rows, overflow %u\n", hres[0], ntiles,
                        hres[1], hres[2], hres[3]);
        }
        (void)KNN_DEV_FREE(cell_code);
        (void)KNN_DEV_FREE(cell_fill);
        (void)KNN_DEV_FREE(st.cells->tmp_rows);
        (void)KNN_DEV_FREE(st.cells->tmp_meta);
        (void)KNN_DEV_FREE(st.cells->bucket_start);
        st.cells->tmp_rows = nullptr;
        st.cells->tmp_meta = nullptr;
        st.cells->bucket_start = nullptr;
        (void)KNN_DEV_FREE(dout);
        st.cells->bucket_fill = nullptr;
        st.cells->build_res = nullptr;
        if (e != hipSuccess || fast_overflow || hout[2] != 0u || hout[3] > ocap) {   // (too many rows outside the box: no layouts from this frame)
            knn_filter_free(st);
            return e;
        }
        st.n_outliers = hout[3];
        memcpy(&st.bmax, &hout[0], 4);
        memcpy(&st.nmax, &hout[1], 4);
        st.usable = true;
        if (!geom) {   // clustered data (or on request): the fragments again, each cell in its own frame
            e = knn_cells_maybe_recentre(st, r, samples > 0 ? samp.data() : nullptr, samples, s);
            lap("per-cell frames (if any)");
            if (e != hipSuccess)
                knn_filter_free(st);
        }
        return e;
    };
    if (geom) {
        // Cell-range shard of a global grid: centre, scale and cuts are the grid's (identical on every rank: the ranks' fp16
        // fragments — the seed layer — must live in one frame), the layout is this rank's cells of it.
        if (kt != 1 || geom->k != k)
            return hipSuccess;
        std::vector<float> none;
        float center16[16];
        memcpy(center16, geom->center, sizeof center16);
        return sorted_layout(center16, geom->sigma, none, 0, nullptr);
    }
    if (want_cells && kt <= 2 && n >= (1ll << 17)) {
        // Round 4: the frame of a cell-sorted layout from a strided SAMPLE of the rows (4096 of them: range widened by 1 / 32,
        // median / MAD box inside it — what knn_filter_build_from_host does for host rows), not from a pass over the whole
        // shard: the full-range statistics kernel read 1 GiB for a box that any representative sample gives, and its round
        // trip to the host stood in front of everything else (0.25 of the build's 3.4 ms at C3).  ANY box is correct: rows
        // outside it go to the exact list; if the sample was not representative (more than n / 32 rows outside, or values that
        // are not finite in the sample) the classic build below starts over with full-range statistics.
        const long long samples = 4096, row_stride = n / samples;
        std::vector<float> samp((size_t)samples * k);
        float *dsamp = nullptr;
        FTRY(KNN_DEV_ALLOC((void **)&dsamp, samp.size() * sizeof(float)));
        hipLaunchKernelGGL(knn_sample_rows_kernel, dim3((unsigned)((samples * k + 255) / 256)), dim3(256), 0, s, r, k, row_stride,
                           samples, dsamp);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess)
            e = hipMemcpyAsync(samp.data(), dsamp, samp.size() * sizeof(float), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess)
            e = hipStreamSynchronize(s);
        (void)KNN_DEV_FREE(dsamp);
        FTRY(e);
        lap("sample rows + copy");
        // (the frame is host arithmetic on the sample — 0.25 ms — and only the placement needs it: it is worked out on a thread
        // of its own while the cell codes and buckets are made on the GPU)
        std::vector<float> center;
        float sigma = 1.0f;
        bool box_ok = false;
        std::thread boxer([&] { box_ok = box_from_sample(samp.data(), samples, k, 16 * kt, center, &sigma); });
        std::vector<float> cut_samp((size_t)(samples / 4) * k);   // (the cuts from every fourth sample row, 1024 as before)
        for (long long i = 0; i < samples / 4; ++i)
            memcpy(&cut_samp[(size_t)i * k], &samp[(size_t)(4 * i) * k], (size_t)k * sizeof(float));
        float center16[32] = {0};   // (16 kt entries used)
        struct Joiner {   // (every way out of this block waits for the thread)
            std::thread &t;
            ~Joiner()
            {
                if (t.joinable())
                    t.join();
            }
        } joiner{boxer};
        auto frame = [&](float *c16, float *sg) -> bool {
            if (boxer.joinable())
                boxer.join();
            if (!box_ok)
                return false;
            for (int d = 0; d < 16 * kt; ++d)
                c16[d] = d < k ? center[(size_t)d] : 0.0f;
            *sg = sigma;
            return true;
        };
        {
            FTRY(sorted_layout(center16, sigma, cut_samp, samples / 4, frame));
            if (st.usable)
                return hipSuccess;
            if (fast_overflow) {   // a bucket outgrew its fixed room (rows the cuts do not spread evenly): the counted build
                lap("fast build: bucket overflow");
                fast_overflow = false;
                fast_build = false;
                st = FilterState();
                FTRY(sorted_layout(center16, sigma, cut_samp, samples / 4, frame));
                if (st.usable)
                    return hipSuccess;
            }
            st = FilterState();   // (declined, or the sampled frame left too many rows out: the classic build decides)
        }
    }
    const int kp = 16 * kt;
    long long ntiles = (n + 31) / 32;

    // 1. per-dimension range
    std::vector<unsigned> hstats((size_t)2 * k + 1);
    for (int d = 0; d < k; ++d) {
        hstats[(size_t)d] = 0xFFFFFFFFu;
        hstats[(size_t)k + d] = 0u;
    }
    hstats[(size_t)2 * k] = 0u;
    unsigned *dstats = nullptr;
    FTRY(KNN_DEV_ALLOC((void **)&dstats, hstats.size() * sizeof(unsigned)));
    hipError_t e = hipMemcpyAsync(dstats, hstats.data(), hstats.size() * sizeof(unsigned),
                                  hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        if (k % 4 == 0 && ((uintptr_t)r & 15u) == 0)
            hipLaunchKernelGGL(knn_ref_stats4_kernel, dim3(2048), dim3(256), 2 * (size_t)k * sizeof(unsigned), s, (const f4v *)r,
                               n * (long long)k / 4, k, dstats);
        else
            hipLaunchKernelGGL(knn_ref_stats_kernel, dim3(2048), dim3(256), 2 * (size_t)k * sizeof(unsigned), s, r, n * (long long)k, k, dstats);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(hstats.data(), dstats, hstats.size() * sizeof(unsigned), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);
    (void)KNN_DEV_FREE(dstats);
    FTRY(e);
    lap("range kernel + sync");
    // NaN / Inf among the references: those rows are outside any box and go to the exact list like every other
    // outlier (the fragment kernels test for them); only a shard that is mostly such rows gets no layouts
    const bool has_nonfinite = hstats[(size_t)2 * k] != 0u;
    if ((long long)hstats[(size_t)2 * k] > n / 32)
        return hipSuccess;
    for (int d = 0; d < k; ++d)
        if (hstats[(size_t)d] == 0xFFFFFFFFu || hstats[(size_t)k + d] == 0u)
            return hipSuccess;  // a dimension without a single finite value

    // 1b. robust box: per dimension [median - w s, median + w s] clipped to [min, max], with
    // s = 1.4826 * MAD from a strided sample of up to 1024 rows (median/MAD do not move when a few rows sit
    // 300 sigma out; mean/std do).  A few far-out rows would otherwise stretch the box, and with it
    // the fp16 step, for everybody.  ANY box is correct: rows outside it leave the filter and are
    // scanned exactly on every query, so the box is only worth it if it leaves out a handful of
    // rows — w doubles from 12 until at most 1 % of the sample falls outside (heavy tails), and a
    // second mode further out than 96 s (more than 1 % of the rows) keeps the plain [min, max].
    const long long want = k <= 16 ? 1024 : (16384 / k > 256 ? 16384 / k : 256);  // host work ~0.2 ms at any k
    const long long samples = n < want ? n : want;
    const long long row_stride = n / samples;
    std::vector<float> samp((size_t)samples * k);
    {
        float *dsamp = nullptr;
        FTRY(KNN_DEV_ALLOC((void **)&dsamp, samp.size() * sizeof(float)));
        hipLaunchKernelGGL(knn_sample_rows_kernel, dim3((unsigned)((samples * k + 255) / 256)), dim3(256), 0, s, r, k,
                           row_stride, samples, dsamp);
        e = hipGetLastError();
        if (e == hipSuccess)
            e = hipMemcpyAsync(samp.data(), dsamp, samp.size() * sizeof(float), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess)
            e = hipStreamSynchronize(s);
        (void)KNN_DEV_FREE(dsamp);
        FTRY(e);
        lap("sample rows + copy");
    }
    long long samples_used = samples;
    if (has_nonfinite) {   // the statistics below want finite rows only
        samples_used = 0;
        for (long long i = 0; i < samples; ++i) {
            bool fin = true;
            for (int d = 0; d < k; ++d)
                fin = fin && fabsf(samp[(size_t)i * k + d]) < INFINITY;
            if (fin) {
                if (samples_used != i)
                    memcpy(&samp[(size_t)samples_used * k], &samp[(size_t)i * k], (size_t)k * sizeof(float));
                ++samples_used;
            }
        }
        if (samples_used < 1)
            return hipSuccess;
    }
    std::vector<float> center, dlo((size_t)k), dhi((size_t)k);
    for (int d = 0; d < k; ++d) {
        dlo[(size_t)d] = ord2f_host(hstats[(size_t)d]);
        dhi[(size_t)d] = ord2f_host(hstats[(size_t)k + d]);
    }
    const double h = robust_box(samp, samples_used, k, kp, dlo, dhi, center);
    lap("median / MAD box (host)");
    if (!(h <= 1e15) || (h != 0.0 && h < 1e-15))
        return hipSuccess;
    float sigma = 1.0f;
    if (h > 0.0) {
        int ex;
        (void)frexp(h, &ex);  // h = f * 2^ex, f in [0.5, 1)  ->  h * 2^-ex < 1
        sigma = (float)ldexp(1.0, -ex);
    }

    // 1c. cell-sorted layout (k <= 16, resident indexes): ntiles becomes the padded tile count
    unsigned *cell_code = nullptr, *cell_fill = nullptr;
    if (want_cells && kt == 1) {
        FTRY(knn_cells_build(&st.cells, k, n, r, samp, samples_used, s, &ntiles, &cell_code, &cell_fill, want_cells == 2));
        lap(st.cells ? "cell codes + counts" : "cell codes (not kept)");
    }

    // 2. fragments + norms
    st.k = k;
    st.kt = kt;
    st.n = n;
    st.ntiles = ntiles;
    st.sigma = sigma;
    unsigned *dout = nullptr;
    e = KNN_DEV_ALLOC((void **)&st.center, (size_t)kp * sizeof(float));
    if (e == hipSuccess)
        e = KNN_DEV_ALLOC(&st.ref_frags, (size_t)ntiles * kt * 64 * 16);
    if (e == hipSuccess)
        e = KNN_DEV_ALLOC((void **)&st.ref_norms, (size_t)ntiles * 32 * sizeof(float));
    if (e == hipSuccess && st.cells)
        e = KNN_DEV_ALLOC((void **)&st.ref_norms2, (size_t)ntiles * 32 * sizeof(unsigned));
    const unsigned ocap = (unsigned)(n / 32 > 4096 ? n / 32 : 4096);  // more outliers than this: no filter
    if (e == hipSuccess)
        e = KNN_DEV_ALLOC((void **)&st.outliers, (size_t)ocap * sizeof(unsigned));
    if (e == hipSuccess)
        e = KNN_DEV_ALLOC((void **)&dout, 4 * sizeof(unsigned));
    if (e == hipSuccess)
        e = hipMemsetAsync(dout, 0, 4 * sizeof(unsigned), s);
    if (e == hipSuccess)
        e = hipMemcpyAsync(st.center, center.data(), (size_t)kp * sizeof(float), hipMemcpyHostToDevice, s);
    unsigned hout[4] = {0, 0, 0, 0};
    lap("allocations");
    if (e == hipSuccess) {
        const long long rows_padded = ntiles * 32;
        if (st.cells) {
            e = knn_cells_place_rows(st, r, cell_code, cell_fill, dout, ocap, s);
        } else if (k == 16 && ((uintptr_t)r & 15u) == 0)
            hipLaunchKernelGGL(knn_frag16_kernel, dim3((unsigned)((rows_padded + 255) / 256)), dim3(256), 0, s,
                               (const f4v *)r, n, rows_padded, st.center, sigma, (h8 *)st.ref_frags,
                               st.ref_norms, dout, st.outliers, ocap);
        else
            hipLaunchKernelGGL(knn_frag_kernel, dim3((unsigned)((rows_padded + 255) / 256)), dim3(256), 0, s, r,
                               n, rows_padded, k, kt, st.center, sigma, 1.0f, INFINITY, (h8 *)st.ref_frags,
                               st.ref_norms, dout, 0, nullptr, nullptr, st.outliers, ocap);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(hout, dout, sizeof hout, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);  // also keeps `center` alive until the copy is done
    lap("fragment kernel + sync");
    (void)KNN_DEV_FREE(cell_code);
    (void)KNN_DEV_FREE(cell_fill);
    if (st.cells) {   // the two-pass build's scratch (the stream has been synchronised: the placement kernel is done)
        (void)KNN_DEV_FREE(st.cells->tmp_rows);
        (void)KNN_DEV_FREE(st.cells->tmp_meta);
        (void)KNN_DEV_FREE(st.cells->bucket_start);
        st.cells->tmp_rows = nullptr;
        st.cells->tmp_meta = nullptr;
        st.cells->bucket_start = nullptr;
    }
    (void)KNN_DEV_FREE(dout);
    if (e != hipSuccess) {
        knn_filter_free(st);
        return e;
    }
    if (hout[2] != 0u || hout[3] > ocap) {  // fp16 range trouble, or too many rows outside the box
        knn_filter_free(st);
        return hipSuccess;
    }
    st.n_outliers = hout[3];
    memcpy(&st.bmax, &hout[0], 4);
    memcpy(&st.nmax, &hout[1], 4);
    st.usable = true;
    if (st.cells) {
        e = knn_cells_maybe_recentre(st, r, samp.data(), samples_used, s);
        if (e != hipSuccess)
            knn_filter_free(st);
    }
    return e;
}

// ------------------------------------------------------------------------------------------
// Ingest: host rows -> device rows + filter layouts in ONE pass over PCIe.
// The reference copies the whole shard with a pageable cudaMemcpy and only then starts working on it
// (core.cu:885-891); its own whole-callback timings are that copy (README.md:291-292).  Here the
// rows go over in chunks on `copy` and every chunk is turned into fragments + norms on `compute` as
// soon as it has landed, so the index is resident — rows AND MFMA layouts — one fragment kernel
// (~20 us per 64 MiB chunk) after the last byte arrives.
// What made the layouts need the whole shard was the robust box (range + median/MAD).  With the rows
// still on the host the box comes from a strided HOST sample of up to 4096 rows taken before the
// first byte moves (~0.5 ms): median/MAD as before, clipped to the sample's min/max widened by 1/32
// of its width.  Any box is correct — rows outside it (a far-out row in a
// late chunk, a NaN, an Inf) go to the exact list as always; only if more than n/32 rows end up
// there (the sample was not representative) are the layouts rebuilt from the resident rows with the
// full-range statistics of knn_filter_build.
// r_dev: destination, n x k floats on the current device.  Synchronous.
// ------------------------------------------------------------------------------------------
hipError_t knn_filter_build_from_host(FilterState &st, int k, long long n, float *r_dev, const float *r_host,
                                      hipStream_t copy, hipStream_t compute)
{
    st = FilterState();
    const bool trace = getenv("KNN_MI355X_TRACE_BUILD") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace)
            return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[knn ingest] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const size_t row_bytes = (size_t)k * sizeof(float);
    const bool layouts = n > 0 && knn_kt_of(k) != 0;
    const int kt = layouts ? knn_kt_of(k) : 1;
    const int kp = 16 * kt;
    const long long ntiles = (n + 31) / 32;

    // 1. box from a host sample
    bool usable = layouts;
    std::vector<float> center;
    float sigma = 1.0f;
    if (usable) {
        // 4096 strided rows give the range (every read is a cache + TLB miss: 16384 rows cost 3.6 ms, more than
        // the layouts), every 4th of them — 1024 rows, as in knn_filter_build — the median / MAD
        const long long samples = n < 4096 ? n : 4096;
        const long long row_stride = n / samples;
        const long long sub = samples >= 4096 ? 4 : 1, nsub = samples / sub;
        std::vector<float> samp((size_t)nsub * k), dlo((size_t)k, INFINITY), dhi((size_t)k, -INFINITY);
        for (long long i = 0; i < samples && usable; ++i) {
            const float *x = r_host + (size_t)(i * row_stride) * k;
            for (int d = 0; d < k; ++d) {
                const float v = x[d];
                if (!(fabsf(v) < INFINITY))
                    usable = false;   // non-finite rows in the sample: leave it to the classic build
                if (i % sub == 0 && i / sub < nsub)
                    samp[(size_t)(i / sub) * k + d] = v;
                dlo[(size_t)d] = fminf(dlo[(size_t)d], v);
                dhi[(size_t)d] = fmaxf(dhi[(size_t)d], v);
            }
        }
        if (usable) {
            for (int d = 0; d < k; ++d) {
                const float pad = (dhi[(size_t)d] - dlo[(size_t)d]) * (1.0f / 32.0f);
                dlo[(size_t)d] -= pad;
                dhi[(size_t)d] += pad;
            }
            const double h = robust_box(samp, nsub, k, kp, dlo, dhi, center);
            if (!(h <= 1e15) || (h != 0.0 && h < 1e-15))
                usable = false;
            else if (h > 0.0) {
                int ex;
                (void)frexp(h, &ex);
                sigma = (float)ldexp(1.0, -ex);
            }
        }
    }

    lap("host sample + box");
    // 2. buffers
    unsigned *dout = nullptr;
    const unsigned ocap = (unsigned)(n / 32 > 4096 ? n / 32 : 4096);
    hipError_t e = hipSuccess;
    if (usable) {
        st.k = k;
        st.kt = kt;
        st.n = n;
        st.ntiles = ntiles;
        st.sigma = sigma;
        e = KNN_DEV_ALLOC((void **)&st.center, (size_t)kp * sizeof(float));
        if (e == hipSuccess)
            e = KNN_DEV_ALLOC(&st.ref_frags, (size_t)ntiles * kt * 64 * 16);
        if (e == hipSuccess)
            e = KNN_DEV_ALLOC((void **)&st.ref_norms, (size_t)ntiles * 32 * sizeof(float));
        if (e == hipSuccess)
            e = KNN_DEV_ALLOC((void **)&st.outliers, (size_t)ocap * sizeof(unsigned));
        if (e == hipSuccess)
            e = KNN_DEV_ALLOC((void **)&dout, 4 * sizeof(unsigned));
        if (e == hipSuccess)
            e = hipMemsetAsync(dout, 0, 4 * sizeof(unsigned), compute);
        if (e == hipSuccess)
            e = hipMemcpyAsync(st.center, center.data(), (size_t)kp * sizeof(float), hipMemcpyHostToDevice, compute);
        if (e == hipErrorOutOfMemory) {   // no room for the layouts beside the rows: rows only
            (void)hipGetLastError();
            (void)KNN_DEV_FREE(dout);
            dout = nullptr;
            knn_filter_free(st);
            usable = false;
            e = hipSuccess;
        }
        if (e != hipSuccess) {
            (void)KNN_DEV_FREE(dout);
            knn_filter_free(st);
            return e;
        }
    }

    lap("allocations");
    // 3. Two chunks: everything but the last 64 MiB in ONE pageable copy, then the tail.  A pageable
    // hipMemcpy runs at the link rate (55 GB/s: the runtime pins the caller's pages as it goes) but every
    // call costs ~0.25 ms of pipeline fill, so sixteen 64 MiB chunks lost 4 ms against one 1 GiB copy
    // (profiles/r02_ingest_timing.txt) where the whole layout build is 0.8 ms.  The big chunk's fragment
    // kernels (0.8 ms for 960 MiB) run under the tail's copy (1.2 ms); what is left after the last byte is
    // the tail's own fragment kernel (~0.05 ms).  Shards up to 128 MiB go over in one piece.
    const long long tail_rows = ((long long)((64u << 20) / row_bytes) + 1023) / 1024 * 1024;
    const long long head_rows = (size_t)n * row_bytes > ((size_t)128u << 20) ? (n - tail_rows) / 1024 * 1024 : n;
    std::vector<hipEvent_t> events;
    for (long long r0 = 0; r0 < n && e == hipSuccess;) {
        const long long chunk_rows = r0 == 0 ? head_rows : n - r0;
        const long long r1 = std::min(n, r0 + chunk_rows);
        e = hipMemcpyAsync(r_dev + (size_t)r0 * k, r_host + (size_t)r0 * k, (size_t)(r1 - r0) * row_bytes,
                           hipMemcpyHostToDevice, copy);
        const long long r0_this = r0;
        r0 = r1;
        lap("copy call returned");
        if (!usable || e != hipSuccess)
            continue;
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess)
            break;
        events.push_back(ev);
        e = hipEventRecord(ev, copy);
        if (e == hipSuccess)
            e = hipStreamWaitEvent(compute, ev, 0);
        if (e != hipSuccess)
            break;
        // rows c0 .. r1 (c0 is a multiple of 1024): tiles c0/32 .., the last chunk pads its last tile
        const long long c0 = r0_this;
        const long long rows = r1 - c0;
        const long long rows_padded = r1 == n ? ntiles * 32 - c0 : rows;
        const float *x = r_dev + (size_t)c0 * k;
        h8 *frag = (h8 *)st.ref_frags + (size_t)(c0 / 32) * kt * 64;
        float *norms = st.ref_norms + c0;
        const unsigned blocks = (unsigned)((rows_padded + 255) / 256);
        if (k == 16 && ((uintptr_t)x & 15u) == 0)
            hipLaunchKernelGGL(knn_frag16_kernel, dim3(blocks), dim3(256), 0, compute, (const f4v *)x, rows, rows_padded,
                               st.center, sigma, frag, norms, dout, st.outliers, ocap, (unsigned)c0);
        else
            hipLaunchKernelGGL(knn_frag_kernel, dim3(blocks), dim3(256), 0, compute, x, rows, rows_padded, k, kt, st.center,
                               sigma, 1.0f, INFINITY, frag, norms, dout, 0, nullptr, nullptr, st.outliers, ocap, (unsigned)c0);
        e = hipGetLastError();
    }
    unsigned hout[4] = {0, 0, 0, 0};
    if (usable && e == hipSuccess)
        e = hipMemcpyAsync(hout, dout, sizeof hout, hipMemcpyDeviceToHost, compute);
    const hipError_t e1 = hipStreamSynchronize(copy), e2 = hipStreamSynchronize(compute);
    lap("streams drained");
    if (e == hipSuccess)
        e = e1 != hipSuccess ? e1 : e2;
    for (hipEvent_t ev : events)
        (void)hipEventDestroy(ev);
    (void)KNN_DEV_FREE(dout);
    lap("events + scratch released");
    if (e != hipSuccess) {
        knn_filter_free(st);
        return e;
    }
    if (!layouts)
        return hipSuccess;
    if (usable && hout[2] == 0u && hout[3] <= ocap) {
        st.n_outliers = hout[3];
        memcpy(&st.bmax, &hout[0], 4);
        memcpy(&st.nmax, &hout[1], 4);
        st.usable = true;
        return hipSuccess;
    }
    // the sampled box did not fit the data (or the sample held non-finite values): classic build from
    // the rows that are now resident
    knn_filter_free(st);
    return knn_filter_build(st, k, n, r_dev, compute);
}

// ------------------------------------------------------------------------------------------
// Ingest into a CELL-SORTED index (round 5; SURVEY §8 f1, VERDICT r04 missing 5): host rows -> device rows + the pruned scan's
// layouts, the bucket pass of the fast build running chunk by chunk UNDER the copy.  Rounds 2-4 copied first and sorted
// afterwards (the counted build needs every row before it can place one: +2.5 ms behind the last byte at C3, +13 % over the
// bare copy).  The fast build's buckets have fixed room, so a chunk can be scattered the moment it has landed; what is left
// behind the last byte is the tail chunk's scatter, the cell prefix and the placement (~0.8 ms).  Box and cuts come from a
// strided HOST sample of 4096 rows taken before the first byte moves (as knn_filter_build_from_host).
// Always leaves the rows on the device (r_dev) when it returns hipSuccess; st.usable says whether the layouts stand — if not
// (a bucket overflowed, the sample was not finite or not representative, no room for the scratch) the caller builds from the
// resident rows.  Synchronous.
// ------------------------------------------------------------------------------------------
hipError_t knn_filter_build_cells_from_host(FilterState &st, int k, long long n, float *r_dev, const float *r_host,
                                            hipStream_t copy, hipStream_t compute)
{
    st = FilterState();
    const bool trace = getenv("KNN_MI355X_TRACE_BUILD") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace)
            return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[knn ingest] %-36s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const size_t row_bytes = (size_t)k * sizeof(float);
    auto plain_copy = [&]() -> hipError_t {
        FTRY(hipMemcpyAsync(r_dev, r_host, (size_t)n * row_bytes, hipMemcpyHostToDevice, copy));
        return hipStreamSynchronize(copy);
    };
    if (k > 16 || n < (1ll << 17))
        return plain_copy();
    // 1. frame and cuts from a host sample
    const long long samples = 4096, row_stride = n / samples;
    std::vector<float> samp((size_t)samples * k);
    for (long long i = 0; i < samples; ++i)
        memcpy(&samp[(size_t)i * k], r_host + (size_t)(i * row_stride) * k, row_bytes);
    std::vector<float> center;
    float sigma = 1.0f;
    if (!box_from_sample(samp.data(), samples, k, 16, center, &sigma))
        return plain_copy();
    std::vector<float> cut_samp((size_t)(samples / 4) * k);
    for (long long i = 0; i < samples / 4; ++i)
        memcpy(&cut_samp[(size_t)i * k], &samp[(size_t)(4 * i) * k], row_bytes);
    lap("host sample, box, cuts");
    // 2. the fast build, everything but its scatter
    long long ntiles = 0;
    unsigned *cell_code = nullptr, *cell_fill = nullptr;
    FTRY(knn_cells_build(&st.cells, k, n, r_dev, cut_samp, samples / 4, compute, &ntiles, &cell_code, &cell_fill, false, nullptr, 0, nullptr,
                         true, true));
    if (!st.cells)
        return plain_copy();
    st.k = k;
    st.kt = 1;
    st.n = n;
    st.ntiles = ntiles;   // (room; the build's own count replaces it below)
    st.sigma = sigma;
    float center16[16];
    for (int d = 0; d < 16; ++d)
        center16[d] = d < k ? center[(size_t)d] : 0.0f;
    unsigned *dout = nullptr;
    const unsigned ocap = (unsigned)(n / 32 > 4096 ? n / 32 : 4096);
    hipError_t e = KNN_DEV_ALLOC((void **)&st.center, 16 * sizeof(float));
    if (e == hipSuccess)
        e = KNN_DEV_ALLOC(&st.ref_frags, (size_t)ntiles * 64 * 16);
    if (e == hipSuccess)
        e = KNN_DEV_ALLOC((void **)&st.ref_norms, (size_t)ntiles * 32 * sizeof(float));
    if (e == hipSuccess)
        e = KNN_DEV_ALLOC((void **)&st.ref_norms2, (size_t)ntiles * 32 * sizeof(unsigned));
    if (e == hipSuccess)
        e = KNN_DEV_ALLOC((void **)&st.outliers, (size_t)ocap * sizeof(unsigned));
    if (e == hipSuccess)
        e = KNN_DEV_ALLOC((void **)&dout, 4 * sizeof(unsigned));
    if (e == hipSuccess)
        e = hipMemsetAsync(dout, 0, 4 * sizeof(unsigned), compute);
    if (e == hipSuccess)
        e = hipMemcpyAsync(st.center, center16, sizeof center16, hipMemcpyHostToDevice, compute);
    if (e != hipSuccess) {   // no room for the layouts: rows only (the caller's build will find the same and say so)
        (void)hipGetLastError();
        (void)hipStreamSynchronize(compute);
        (void)KNN_DEV_FREE(dout);
        (void)KNN_DEV_FREE(cell_fill);
        knn_filter_free(st);
        return plain_copy();
    }
    lap("allocations");
    // 3. two chunks (see knn_filter_build_from_host: a pageable copy runs at the link rate but every call costs ~0.25 ms of
    // pipeline fill): everything but the last 64 MiB, then the tail; each is scattered into the buckets as soon as it is there
    const long long tail_rows = ((long long)((64u << 20) / row_bytes) + 4095) / 4096 * 4096;
    const long long head_rows = (size_t)n * row_bytes > ((size_t)128u << 20) ? (n - tail_rows) / 4096 * 4096 : n;
    std::vector<hipEvent_t> events;
    for (long long r0 = 0; r0 < n && e == hipSuccess;) {
        const long long r1 = r0 == 0 ? head_rows : n;
        e = hipMemcpyAsync(r_dev + (size_t)r0 * k, r_host + (size_t)r0 * k, (size_t)(r1 - r0) * row_bytes, hipMemcpyHostToDevice, copy);
        lap("copy call returned");
        hipEvent_t ev = nullptr;
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) {
            events.push_back(ev);
            e = hipEventRecord(ev, copy);
        }
        if (e == hipSuccess)
            e = hipStreamWaitEvent(compute, ev, 0);
        if (e == hipSuccess)
            e = knn_cells_fast_scatter(*st.cells, k, r_dev, r0, r1, compute);
        r0 = r1;
    }
    // 4. cell prefix, placement, padding — and ONE synchronisation
    unsigned hout[4] = {0, 0, 0, 0}, hres[4] = {0, 0, 0, 0};
    if (e == hipSuccess)
        e = knn_cells_fast_finish(*st.cells, cell_fill, compute);
    if (e == hipSuccess)
        e = knn_cells_place_rows(st, r_dev, nullptr, cell_fill, dout, ocap, compute);
    if (e == hipSuccess)
        e = hipMemcpyAsync(hout, dout, sizeof hout, hipMemcpyDeviceToHost, compute);
    if (e == hipSuccess)
        e = hipMemcpyAsync(hres, st.cells->build_res, sizeof hres, hipMemcpyDeviceToHost, compute);
    const hipError_t e1 = hipStreamSynchronize(copy), e2 = hipStreamSynchronize(compute);
    lap("streams drained");
    if (e == hipSuccess)
        e = e1 != hipSuccess ? e1 : e2;
    for (hipEvent_t ev : events)
        (void)hipEventDestroy(ev);
    (void)KNN_DEV_FREE(cell_fill);
    (void)KNN_DEV_FREE(st.cells->tmp_rows);
    (void)KNN_DEV_FREE(st.cells->tmp_meta);
    (void)KNN_DEV_FREE(st.cells->bucket_start);
    st.cells->tmp_rows = nullptr;
    st.cells->tmp_meta = nullptr;
    st.cells->bucket_start = nullptr;
    st.cells->bucket_fill = nullptr;
    st.cells->build_res = nullptr;
    (void)KNN_DEV_FREE(dout);
    if (trace)
        fprintf(stderr, "[knn ingest] fast build under the copy: %u tiles (room for %lld), %u items, largest cell %u rows, overflow %u, outside the box %u\n",
                hres[0], ntiles, hres[1], hres[2], hres[3], hout[3]);
    if (e != hipSuccess) {
        knn_filter_free(st);
        return e;
    }
    if (hres[3] != 0u || hres[1] == 0u || hout[2] != 0u || hout[3] > ocap) {   // the caller builds from the resident rows
        knn_filter_free(st);
        return hipSuccess;
    }
    st.ntiles = hres[0];
    st.cells->nitems = hres[1];
    st.cells->max_cell_rows = hres[2];
    st.n_outliers = hout[3];
    memcpy(&st.bmax, &hout[0], 4);
    memcpy(&st.nmax, &hout[1], 4);
    st.usable = true;
    e = knn_cells_maybe_recentre(st, r_dev, cut_samp.data(), samples / 4, compute);
    if (e != hipSuccess)
        knn_filter_free(st);
    return e;
}

static hipError_t ensure_workspace(FilterState &st, FilterWorkspace &w, int m)
{
    if (!w.ctl) {
        // three blocks of control words (FilterWorkspace::ctl); the cell-pruned path wants its two cleared once
        FTRY(KNN_DEV_ALLOC((void **)&w.ctl, 3 * KNN_CTL_WORDS * sizeof(unsigned)));
        FTRY(hipMemset(w.ctl, 0, 3 * KNN_CTL_WORDS * sizeof(unsigned)));
        w.ctl_cur = w.ctl;
    }
    if (!w.records) {
        // 8-byte records followed by their 2-byte row masks (written by the deep-K kernel only)
        FTRY(KNN_DEV_ALLOC((void **)&w.records, (size_t)kRecordCapacity * (sizeof(u64) + sizeof(unsigned short))));
        w.rec_cap = kRecordCapacity;
    }
    if (!w.counts)
        FTRY(KNN_DEV_ALLOC((void **)&w.counts, (size_t)kMaxLists * sizeof(unsigned)));
    if (m > w.m_cap) {
        (void)KNN_DEV_FREE(w.qry_frags);
        (void)KNN_DEV_FREE(w.qry_norms);
        (void)KNN_DEV_FREE(w.thr);
        w.qry_frags = nullptr;
        w.qry_norms = nullptr;
        w.thr = nullptr;
        w.m_cap = 0;
        const size_t qtiles = (size_t)(m + 31) / 32;
        FTRY(KNN_DEV_ALLOC(&w.qry_frags, qtiles * st.kt * 64 * 16));
        FTRY(KNN_DEV_ALLOC((void **)&w.qry_norms, qtiles * 32 * sizeof(float)));
        (void)KNN_DEV_FREE(w.qry_amax);
        w.qry_amax = nullptr;
        FTRY(KNN_DEV_ALLOC((void **)&w.qry_amax, qtiles * 32 * sizeof(float)));
        FTRY(KNN_DEV_ALLOC((void **)&w.thr, 4 * qtiles * 32 * sizeof(float)));   // thresholds | margins | floors | running (see knn_thr_kernel)
        (void)KNN_DEV_FREE(w.qpart);
        w.qpart = nullptr;
        FTRY(KNN_DEV_ALLOC((void **)&w.qpart, 3 * ((qtiles * 32 + 255) / 256) * sizeof(unsigned)));
        w.m_cap = (int)(qtiles * 32);
    }
    return hipSuccess;
}

static hipError_t prep_queries(FilterState &st, FilterWorkspace &w, int m, const float *q, hipStream_t s)
{
    // no memsets: the fragment kernel writes per-block partials and resets FALLBACK / RECORDS
    const long long rows_padded = ((long long)m + 31) / 32 * 32;
    const unsigned blocks = (unsigned)((rows_padded + 255) / 256);
    hipLaunchKernelGGL(knn_frag_kernel, dim3(blocks), dim3(256), 0, s, q, (long long)m, rows_padded, st.k, st.kt,
                       st.center, st.sigma, -2.0f, 0.0f, (h8 *)w.qry_frags, w.qry_norms, w.qpart, 1, w.ctl,
                       w.qry_amax, nullptr, 0u);
    return hipGetLastError();
}

// A batch's query tiles are cut into pieces, each scanned by a launch whose waves keep QT tiles in
// registers.  A ragged tail no longer pays for a full group: m = 1100 (35 tiles) used to run two
// groups of 32 (1.09 ms at C3's n), now 32 + a piece of 8 (0.76 ms).  Costs per piece measured at
// n = 2^24: QT 32: 0.54 ms, 16: 0.31, 8: 0.22, 2: 0.11.
struct FilterPiece {
    int qt;       // query tiles per wave
    int begin;    // first query tile
    int count;    // query tiles in the piece
    unsigned gx, gy, list_base;
};

static int plan_pieces(int kt, int qtiles, int force_qt, FilterPiece out[4])
{
    int np = 0, pos = 0, rem = qtiles;
    auto push = [&](int qt, int cnt) {
        out[np].qt = qt;
        out[np].begin = pos;
        out[np].count = cnt;
        ++np;
        pos += cnt;
        rem -= cnt;
    };
    if (force_qt > 0 && kt == 1) {
        push(force_qt, rem);
        return np;
    }
    if (kt == 1) {
        if (rem >= 32)
            push(32, rem / 32 * 32);
        if (rem > 18)
            push(32, rem);
        else if (rem > 16) {
            push(16, 16);
            push(2, rem);
        } else if (rem > 8)
            push(16, rem);
        else if (rem > 2)
            push(8, rem);
        else if (rem > 0)
            push(2, rem);
    } else if (kt == 2) {
        if (rem >= 16)
            push(16, rem / 16 * 16);
        if (rem > 8)
            push(16, rem);
        else if (rem > 0)
            push(8, rem);
    } else if (kt == 4) {
        push(4, rem);
    } else {
        push(2, rem);
    }
    return np;
}

template <int KT, int QT>
static void launch_sample_piece(const FilterState &st, const FilterWorkspace &w, const FilterPiece &p, unsigned sb,
                                long long stride, int m_padded, hipStream_t s)
{
    hipLaunchKernelGGL((knn_filter_sample_kernel<KT, QT>), dim3(sb, p.gy), dim3(FILTER_BLOCK), 0, s,
                       (const h8 *)st.ref_frags, st.ref_norms, (const h8 *)w.qry_frags, p.begin + p.count, st.ntiles,
                       stride, w.umin, m_padded, w.ctl, p.begin);
}

template <int KT, int QT>
static void launch_scan_piece(const FilterState &st, const FilterWorkspace &w, const FilterPiece &p, hipStream_t s)
{
    hipLaunchKernelGGL((knn_filter_kernel<KT, QT>), dim3(p.gx, p.gy), dim3(FILTER_BLOCK), 0, s,
                       (const h8 *)st.ref_frags, st.ref_norms, (const h8 *)w.qry_frags + (size_t)p.begin * KT * 64,
                       w.thr + (size_t)p.begin * 32, p.count, st.ntiles, w.records + (size_t)p.list_base * w.slice,
                       w.counts + p.list_base, w.ctl, w.slice);
}

template <int KT>
static hipError_t launch_filter(FilterState &st, FilterWorkspace &w, int m, int num_cu, hipStream_t s)
{
    const int qtiles = (m + 31) / 32;
    const int m_padded = qtiles * 32;
    FilterPiece pc[4];
    const int np = plan_pieces(KT, qtiles, st.force_qt, pc);
    const unsigned target_blocks = (unsigned)num_cu * 8;

    // grids: 2 waves per SIMD when the wave's registers are full of query fragments, more when they
    // are not (small batches are HBM-latency-bound; 5 waves per SIMD measured best at m = 8..64)
    unsigned gy_sum = 0, nlists = 0;
    for (int i = 0; i < np; ++i) {
        FilterPiece &p = pc[i];
        const int qk = p.qt * KT;
        p.gy = (unsigned)((p.count + p.qt - 1) / p.qt);
        long long waves = (long long)num_cu * (qk > 16 ? 8 : qk == 16 ? 12 : qk <= 2 ? 20 : 16);
        waves *= st.force_rounds > 0 ? st.force_rounds : 1;
        if (waves > st.ntiles)
            waves = st.ntiles;
        p.gx = (unsigned)((waves + 3) / 4);
        // many query groups (large m): split the references over fewer waves so every wave still
        // streams a long run of tiles per load of its query fragments
        if (p.gy > 1 && (size_t)p.gx * p.gy > target_blocks)
            p.gx = (target_blocks + p.gy - 1) / p.gy;
        if (p.gx < 1)
            p.gx = 1;
        while ((size_t)p.gx * 4 * p.gy > kMaxLists / 4 && p.gx > 1)
            p.gx = (p.gx + 1) / 2;
        p.list_base = nlists;
        nlists += p.gx * 4 * p.gy;
        gy_sum += p.gy;
    }
    if (nlists == 0 || nlists > kMaxLists)
        return hipErrorInvalidValue;
    w.nlists = nlists;
    w.slice = w.rec_cap / w.nlists;
    w.has_rows = false;
    w.pieces.n = np;
    for (int i = 0; i < 4; ++i) {
        w.pieces.list_base[i] = i < np ? pc[i].list_base : 0xFFFFFFFFu;
        w.pieces.qrow_base[i] = i < np ? (unsigned)pc[i].begin * 32u : 0u;
    }

    // 1. sample pass over every stride-th tile (about 1/16 of the shard) -> per-query minima
    long long stride = st.ntiles / 256;
    if (stride < 1)
        stride = 1;
    if (stride > 16)   // (32 / 64 / 8 A/B'd at C3 in round 2: 0.547 / 0.555 / 0.585 ms per step against 0.5505: flat)
        stride = 16;
    const long long ns = (st.ntiles + stride - 1) / stride;
    unsigned sb = (unsigned)num_cu * 2;  // 2 waves per SIMD, like the main pass
    if (sb > kSampleBlocks)
        sb = kSampleBlocks;
    if (gy_sum > 1 && (size_t)sb * gy_sum > target_blocks)
        sb = (target_blocks + gy_sum - 1) / gy_sum;
    // at least 8 sampled tiles per wave: a wave's prologue (its query fragments, QT KiB) is not
    // worth fewer, and the threshold kernel folds one partial row per block
    if ((long long)sb * 32 > ns)
        sb = (unsigned)((ns + 31) / 32);
    if (sb < 1)
        sb = 1;
    {   // per-block minima buffer, grown on demand
        const size_t need = (size_t)sb * (size_t)m_padded;
        if (need > w.umin_cap) {
            (void)KNN_DEV_FREE(w.umin);
            w.umin = nullptr;
            w.umin_cap = 0;
            FTRY(KNN_DEV_ALLOC((void **)&w.umin, need * sizeof(float)));
            w.umin_cap = need;
        }
    }
    for (int i = 0; i < np; ++i) {
        const FilterPiece &p = pc[i];
        if constexpr (KT == 1) {
            switch (p.qt) {
            case 2: launch_sample_piece<1, 2>(st, w, p, sb, stride, m_padded, s); break;
            case 8: launch_sample_piece<1, 8>(st, w, p, sb, stride, m_padded, s); break;
            case 16: launch_sample_piece<1, 16>(st, w, p, sb, stride, m_padded, s); break;
            default: launch_sample_piece<1, 32>(st, w, p, sb, stride, m_padded, s); break;
            }
        } else if constexpr (KT == 2) {
            if (p.qt == 8)
                launch_sample_piece<2, 8>(st, w, p, sb, stride, m_padded, s);
            else
                launch_sample_piece<2, 16>(st, w, p, sb, stride, m_padded, s);
        } else if constexpr (KT == 4) {
            launch_sample_piece<4, 4>(st, w, p, sb, stride, m_padded, s);
        } else {
            launch_sample_piece<8, 2>(st, w, p, sb, stride, m_padded, s);
        }
        FTRY(hipGetLastError());
    }

    // 2. thresholds
    hipLaunchKernelGGL(knn_thr_kernel, dim3((unsigned)(m_padded / 32)), dim3(32 * THR_PARTS), 0, s, w.umin,
                       (int)sb, w.qry_norms, w.qry_amax, m, m_padded, st.k, st.kt, st.sigma, st.bmax, st.nmax, kAmaxLimit,
                       w.thr, w.ctl, w.qpart, (m_padded + 255) / 256, w.counts, w.nlists);
    FTRY(hipGetLastError());

    // 3. the filter proper (timed: the dominant kernel).  Ordered after the other slot's scan.
    if (!st.scan_done)
        FTRY(hipEventCreateWithFlags(&st.scan_done, hipEventDisableTiming));
    // Scans of different slots: free to overlap (heads fill the other's tail, no event round trip:
    // -17 % per step at n = 2M, -5 % at 8M, -2.5 % at 16M with three batches in flight) unless the
    // shard is >= 16M rows, where they are chained so that a launch's duration stays that of the kernel
    // itself (the roofline is quoted from it; two 256-VGPR scans sharing the SIMDs take ~1.7x as long each).
    const bool no_chain = st.chain_policy == 2 || (st.chain_policy == 0 && st.ntiles < (1ll << 19));
    if (st.scan_recorded && !no_chain)
        FTRY(hipStreamWaitEvent(s, st.scan_done, 0));
    if (w.ev_begin)
        FTRY(hipEventRecord(w.ev_begin, s));
    for (int i = 0; i < np; ++i) {
        const FilterPiece &p = pc[i];
        if constexpr (KT == 1) {
            switch (p.qt) {
            case 2: launch_scan_piece<1, 2>(st, w, p, s); break;
            case 8: launch_scan_piece<1, 8>(st, w, p, s); break;
            case 16: launch_scan_piece<1, 16>(st, w, p, s); break;
            default: launch_scan_piece<1, 32>(st, w, p, s); break;
            }
        } else if constexpr (KT == 2) {
            if (p.qt == 8)
                launch_scan_piece<2, 8>(st, w, p, s);
            else
                launch_scan_piece<2, 16>(st, w, p, s);
        } else if constexpr (KT == 4) {
            launch_scan_piece<4, 4>(st, w, p, s);
        } else {
            launch_scan_piece<8, 2>(st, w, p, s);
        }
        FTRY(hipGetLastError());
    }
    if (w.ev_end)
        FTRY(hipEventRecord(w.ev_end, s));
    if (!no_chain)
        FTRY(hipEventRecord(st.scan_done, s));
    st.scan_recorded = true;
    return hipSuccess;
}

template <int KT, int QT>
static hipError_t launch_filter_tiled(FilterState &st, FilterWorkspace &w, int m, int num_cu, hipStream_t s)
{
    const int qtiles = (m + 31) / 32;
    const int m_padded = qtiles * 32;
    const unsigned gy = (unsigned)((qtiles + 4 * QT - 1) / (4 * QT));
    const unsigned target_blocks = (unsigned)num_cu * 8;
    unsigned gx = (target_blocks + gy - 1) / gy;
    if ((long long)gx > st.ntiles)
        gx = (unsigned)st.ntiles;
    if (gx < 1)
        gx = 1;
    while ((size_t)gx * 4 * gy > kMaxLists && gx > 1)
        gx = (gx + 1) / 2;
    if ((size_t)gx * 4 * gy > kMaxLists)
        return hipErrorInvalidValue;
    w.nlists = gx * 4 * gy;
    w.slice = w.rec_cap / w.nlists;

    long long stride = st.ntiles / 256;
    if (stride < 1)
        stride = 1;
    if (stride > 16)
        stride = 16;
    // Running thresholds (KT = 8, round 5): the scan tightens every query's threshold as it goes, so the sample pass only has
    // to give it a start — every 32nd tile instead of every 8th at C5 (2048 tiles): the pass shrinks 4x, the candidates grow
    // from 210k to 355k of the 642k a fixed threshold left, ms per step 0.9155 (stride 8) / 0.8909 (16) / 0.8829 (32) / 0.9091
    // (64) / 0.9378 (128) on one box (profiles/r05_c5_running_thresholds.txt).  KNN_MI355X_SAMPLE_STRIDE: the sweep's knob.
    if (KT >= 8 && st.run_thresholds != 2)   // (KT = 16, 32: k 129 .. 512, the same scheme since the end of round 5)
        stride = std::min<long long>(32, std::max<long long>(1, st.ntiles / 64));
    if (st.sample_stride > 0)   // option `sample_stride` (the sweep's knob, and the tests' like-for-like comparison)
        stride = std::min<long long>(st.sample_stride, std::max<long long>(1, st.ntiles / 16));
    const long long ns = (st.ntiles + stride - 1) / stride;
    unsigned sb = gx;
    if ((long long)sb > ns)
        sb = (unsigned)ns;
    {
        const size_t need = (size_t)sb * (size_t)m_padded;
        if (need > w.umin_cap) {
            (void)KNN_DEV_FREE(w.umin);
            w.umin = nullptr;
            w.umin_cap = 0;
            FTRY(KNN_DEV_ALLOC((void **)&w.umin, need * sizeof(float)));
            w.umin_cap = need;
        }
    }
    // (the sample pass — 1 / 16 of the tiles, running minima only — keeps the 32 x 32 shape: with eight running minima the
    // 16 x 16 form of it spilled)
    hipLaunchKernelGGL((knn_filter_tiled_kernel<KT, QT, true>), dim3(sb, gy), dim3(FILTER_BLOCK), 0, s,
                       (const h8 *)st.ref_frags, st.ref_norms, (const h8 *)w.qry_frags, w.thr, qtiles, st.ntiles,
                       stride, w.umin, m_padded, w.records, w.counts, w.ctl, w.slice,
                       (unsigned short *)(w.records + w.rec_cap));
    FTRY(hipGetLastError());
    hipLaunchKernelGGL(knn_thr_kernel, dim3((unsigned)(m_padded / 32)), dim3(32 * THR_PARTS), 0, s, w.umin,
                       (int)sb, w.qry_norms, w.qry_amax, m, m_padded, st.k, st.kt, st.sigma, st.bmax, st.nmax, kAmaxLimit,
                       w.thr, w.ctl, w.qpart, (m_padded + 255) / 256, w.counts, w.nlists, nullptr,
                       w.thr + (size_t)w.m_cap, w.thr + 2 * (size_t)w.m_cap, (unsigned *)(w.thr + 3 * (size_t)w.m_cap));
    FTRY(hipGetLastError());
    if (!st.scan_done)
        FTRY(hipEventCreateWithFlags(&st.scan_done, hipEventDisableTiming));
    // Scans of different slots: free to overlap unless the shard is >= 16M rows (as in launch_filter).
    // Chaining the deep-K scans was measured at C5 (k 128, m = n = 65536; profiles/r02_c5_chain_ab.txt): a
    // launch's duration drops from 1.64 to 1.00 ms (0.84 with the GPU to itself) but the step goes UP, 1.043 ->
    // 1.081 ms: the other batch's sample pass, re-rank (640k records) and fragment kernels are ~0.25 ms of real
    // work that overlapping scans were absorbing in each other's tails.
    const bool no_chain = st.chain_policy == 2 || (st.chain_policy == 0 && st.ntiles < (1ll << 19));
    if (st.scan_recorded && !no_chain)
        FTRY(hipStreamWaitEvent(s, st.scan_done, 0));
    if (w.ev_begin)
        FTRY(hipEventRecord(w.ev_begin, s));
    // k > 64 (KT = 8): four reference tiles per barrier staged by LDS-DMA (-2.4 % at C5 against one tile per barrier through
    // registers); k > 128: one tile per barrier — a tile is 16 or 32 KiB there, four of them twice over do not fit the LDS.
    // (8 waves per block and two tiles per barrier were measured too: profiles/r02_c5_variants.txt, r03_deepk.txt.)
    if constexpr (KT == 8)
        hipLaunchKernelGGL((knn_filter_tiled_kernel<KT, QT, false, FILTER_BLOCK, 4, true>), dim3(gx, gy), dim3(FILTER_BLOCK), 0, s,
                           (const h8 *)st.ref_frags, st.ref_norms, (const h8 *)w.qry_frags, w.thr, qtiles, st.ntiles,
                           1ll, w.umin, m_padded, w.records, w.counts, w.ctl, w.slice,
                           (unsigned short *)(w.records + w.rec_cap),
                           st.run_thresholds != 2 ? w.thr + (size_t)w.m_cap : nullptr, w.thr + 2 * (size_t)w.m_cap,
                           st.run_thresholds != 2 ? (unsigned *)(w.thr + 3 * (size_t)w.m_cap) : nullptr);
    else   // (16 x 16 shape from k = 65 up: measured -8 % at k 128, -6 % at k 256 and 512, +7 % at k 64 where a tile is only 16 MFMAs)
        hipLaunchKernelGGL((knn_filter_tiled_kernel<KT, QT, false, FILTER_BLOCK, 1, (KT > 4)>), dim3(gx, gy), dim3(FILTER_BLOCK), 0, s,
                           (const h8 *)st.ref_frags, st.ref_norms, (const h8 *)w.qry_frags, w.thr, qtiles, st.ntiles,
                           1ll, w.umin, m_padded, w.records, w.counts, w.ctl, w.slice,
                           (unsigned short *)(w.records + w.rec_cap),
                           st.run_thresholds != 2 ? w.thr + (size_t)w.m_cap : nullptr, w.thr + 2 * (size_t)w.m_cap,
                           st.run_thresholds != 2 ? (unsigned *)(w.thr + 3 * (size_t)w.m_cap) : nullptr);
    w.has_rows = true;
    w.pieces = RerankPieces();
    FTRY(hipGetLastError());
    if (w.ev_end)
        FTRY(hipEventRecord(w.ev_end, s));
    if (!no_chain)
        FTRY(hipEventRecord(st.scan_done, s));
    st.scan_recorded = true;
    return hipSuccess;
}

// k > 512: the chunked-K scan (knn_filter_chunked_kernel), sample pass -> thresholds -> scan, as launch_filter_tiled.
static hipError_t launch_filter_chunked(FilterState &st, FilterWorkspace &w, int m, int num_cu, hipStream_t s)
{
    const int qtiles = (m + 31) / 32;
    const int m_padded = qtiles * 32;
    const unsigned gy = (unsigned)((qtiles + 4 * CHK_QT - 1) / (4 * CHK_QT));
    const long long groups = (st.ntiles + CHK_T - 1) / CHK_T;
    // ranges of reference tiles: enough blocks for the chip, and at least 16 so that an XCD's 64 resident blocks are FEW
    // query groups (see the kernel: their B fragments have to fit its L2)
    // (measured: 8 / 16 / 32 / 64 ranges within 2 % of each other at (1024, 65536, 65536); four blocks per CU in all instead of
    // eight: the mid-size shapes' step 2-9 % shorter — profiles/r04_deepk.txt)
    unsigned gx = std::max(16u, ((unsigned)num_cu * 4u + gy - 1) / gy);
    if ((long long)gx > groups)
        gx = (unsigned)groups;
    if (gx < 1)
        gx = 1;
    while ((size_t)gx * 4 * gy > kMaxLists && gx > 1)
        gx = (gx + 1) / 2;
    if ((size_t)gx * 4 * gy > kMaxLists)
        return hipErrorInvalidValue;
    w.nlists = gx * 4 * gy;
    w.slice = w.rec_cap / w.nlists;
    long long stride = st.ntiles / 256;
    if (stride < 1)
        stride = 1;
    if (stride > 16)
        stride = 16;
    const long long sgroups = ((st.ntiles + stride - 1) / stride + CHK_T - 1) / CHK_T;
    unsigned sb = gx;
    if ((long long)sb > sgroups)
        sb = (unsigned)sgroups;
    {
        const size_t need = (size_t)sb * (size_t)m_padded;
        if (need > w.umin_cap) {
            (void)KNN_DEV_FREE(w.umin);
            w.umin = nullptr;
            w.umin_cap = 0;
            FTRY(KNN_DEV_ALLOC((void **)&w.umin, need * sizeof(float)));
            w.umin_cap = need;
        }
    }
    hipLaunchKernelGGL(knn_filter_chunked_kernel<true>, dim3(sb, gy), dim3(FILTER_BLOCK), 0, s, (const h8 *)st.ref_frags,
                       st.ref_norms, (const h8 *)w.qry_frags, w.thr, st.kt, qtiles, st.ntiles, stride, w.umin, m_padded, w.records,
                       w.counts, w.ctl, w.slice, (unsigned short *)(w.records + w.rec_cap), 0u, gy);
    FTRY(hipGetLastError());
    hipLaunchKernelGGL(knn_thr_kernel, dim3((unsigned)(m_padded / 32)), dim3(32 * THR_PARTS), 0, s, w.umin,
                       (int)sb, w.qry_norms, w.qry_amax, m, m_padded, st.k, st.kt, st.sigma, st.bmax, st.nmax, kAmaxLimit,
                       w.thr, w.ctl, w.qpart, (m_padded + 255) / 256, w.counts, w.nlists);
    FTRY(hipGetLastError());
    if (w.ev_begin)
        FTRY(hipEventRecord(w.ev_begin, s));
    hipLaunchKernelGGL(knn_filter_chunked_kernel<false>, dim3(gx * ((gy + 7u) / 8u) * 8u), dim3(FILTER_BLOCK), 0, s,
                       (const h8 *)st.ref_frags, st.ref_norms, (const h8 *)w.qry_frags, w.thr, st.kt, qtiles, st.ntiles, 1ll, w.umin,
                       m_padded, w.records, w.counts, w.ctl, w.slice, (unsigned short *)(w.records + w.rec_cap), gx, gy);
    w.has_rows = true;
    w.pieces = RerankPieces();
    FTRY(hipGetLastError());
    if (w.ev_end)
        FTRY(hipEventRecord(w.ev_end, s));
    return hipSuccess;
}

hipError_t knn_filter_query(FilterState &st, int slot, int m, const float *q, const float *r, long long base,
                            u64 *keys, int num_cu, hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end, bool init_keys,
                            int *out_idx)
{
    FilterWorkspace &w = st.ws[slot];
    const unsigned *perm = st.cells ? st.cells->perm : nullptr;
    const long long positions = st.cells ? st.ntiles * 32 : st.n;
    // Every batch decides for itself: one the cells cannot serve (a query nothing bounds, a record slice overflowing)
    // raises its own FALLBACK flag on the device and is answered by the gated exact scan; the next batch is back on
    // the pruned path.  (Round 2 sent the whole index to full scans for 256 calls after such a batch, on a pinned
    // host word read here whenever the host happened to get to it.)
    const bool cells = st.cells && (st.cells_policy != 2 || st.cells->centred) && st.kt <= 2;   // (per-cell frames: the full scan cannot read them)
    w.last_used_cells = cells;
    w.ev_begin = ev_begin;
    w.ev_end = ev_end;
    if (cells) {
        const int cell_batch = KNN_CELL_BATCH;   // (the scan's LDS holds a pass's B operands: 36 KiB, 68 KiB for 16 < k <= 32)
        FTRY(ensure_workspace(st, w, std::min(m, cell_batch)));
        for (int q0 = 0; q0 < m; q0 += cell_batch) {
            const int mb = std::min(cell_batch, m - q0);
            const float *qb = q + (size_t)q0 * st.k;
            u64 *kb = keys + q0;
            FTRY(knn_cells_query(st, w, mb, qb, r, base, kb, num_cu, q0 == 0, s, init_keys, out_idx ? out_idx + q0 : nullptr));
        }
        return hipSuccess;
    }
    FTRY(ensure_workspace(st, w, m));
    w.ctl_cur = w.ctl;
    w.ovf_base = w.ovf_cap = 0u;
    if (init_keys)
        FTRY(knn_keys_fill_launch(keys, m, s));
    FTRY(prep_queries(st, w, m, q, s));
    const int qtiles = (m + 31) / 32;
    switch (st.kt) {
    case 1: FTRY(launch_filter<1>(st, w, m, num_cu, s)); break;
    case 2: FTRY(launch_filter<2>(st, w, m, num_cu, s)); break;
    case 4:
        if (qtiles >= 16)
            FTRY((launch_filter_tiled<4, 4>(st, w, m, num_cu, s)));
        else
            FTRY(launch_filter<4>(st, w, m, num_cu, s));
        break;
    case 8:
        if (qtiles >= 16)
            FTRY((launch_filter_tiled<8, 4>(st, w, m, num_cu, s)));
        else
            FTRY(launch_filter<8>(st, w, m, num_cu, s));
        break;
    // 128 < k <= 512: always the LDS-tiled scan (a reference tile is 16 / 32 KiB: no wave can hold one in registers beside its
    // queries); the B operands of 2 / 1 blocks of 32 queries are the wave's 128 operand registers
    case 16: FTRY((launch_filter_tiled<16, 2>(st, w, m, num_cu, s))); break;
    case 32: FTRY((launch_filter_tiled<32, 1>(st, w, m, num_cu, s))); break;
    default:   // k > 512: K in chunks of 128 dimensions
        if (st.kt % CHK_KC != 0)
            return hipErrorInvalidValue;
        FTRY(launch_filter_chunked(st, w, m, num_cu, s));
        break;
    }
    // exact re-rank of the survivors; a list that overflowed its slice raises the fallback flag
    FTRY(knn_rerank_launch(st.k, positions, q, r, base, w.records,
                           w.has_rows ? (const unsigned short *)(w.records + w.rec_cap) : nullptr, w.counts, w.nlists,
                           w.slice, w.ctl, keys, w.pieces, s, perm));
    // rows outside the robust box never entered the filter: exact scan of that (short) list
    FTRY(knn_exact_gather_launch(st.k, m, st.n_outliers, base, q, r, st.outliers, keys, num_cu, nullptr, s));
    // gated: runs only if the filter was ruled out on the device (bad queries, overflow)
    FTRY(knn_exact_launch(st.k, m, st.n, base, q, r, keys, num_cu, w.ctl + KNN_CTL_FALLBACK, s));
    if (out_idx)
        FTRY(knn_keys_unpack_launch(keys, m, out_idx, s));
    return hipSuccess;
}

hipError_t knn_filter_debug(FilterState &st, int m, const float *q, const float *r, float *scores,
                            float *thr_out, float *qnorm_out, double consts[8], hipStream_t s)
{
    FilterWorkspace &w = st.ws[0];
    if (st.cells)
        return hipErrorInvalidValue;  // scores[q][row] assumes the layout is in row order
    FTRY(ensure_workspace(st, w, m));
    FTRY(prep_queries(st, w, m, q, s));
    const int qtiles = (m + 31) / 32;
    const dim3 grid((unsigned)st.ntiles, (unsigned)qtiles);
    switch (st.kt) {
    case 1: hipLaunchKernelGGL(knn_filter_scores_kernel<1>, grid, dim3(64), 0, s, (const h8 *)st.ref_frags, st.ref_norms, (const h8 *)w.qry_frags, m, st.n, scores); break;
    case 2: hipLaunchKernelGGL(knn_filter_scores_kernel<2>, grid, dim3(64), 0, s, (const h8 *)st.ref_frags, st.ref_norms, (const h8 *)w.qry_frags, m, st.n, scores); break;
    case 4: hipLaunchKernelGGL(knn_filter_scores_kernel<4>, grid, dim3(64), 0, s, (const h8 *)st.ref_frags, st.ref_norms, (const h8 *)w.qry_frags, m, st.n, scores); break;
    case 8: hipLaunchKernelGGL(knn_filter_scores_kernel<8>, grid, dim3(64), 0, s, (const h8 *)st.ref_frags, st.ref_norms, (const h8 *)w.qry_frags, m, st.n, scores); break;
    case 16: hipLaunchKernelGGL(knn_filter_scores_kernel<16>, grid, dim3(64), 0, s, (const h8 *)st.ref_frags, st.ref_norms, (const h8 *)w.qry_frags, m, st.n, scores); break;
    case 32: hipLaunchKernelGGL(knn_filter_scores_kernel<32>, grid, dim3(64), 0, s, (const h8 *)st.ref_frags, st.ref_norms, (const h8 *)w.qry_frags, m, st.n, scores); break;
    default: hipLaunchKernelGGL(knn_filter_scores_rt_kernel, grid, dim3(64), 0, s, (const h8 *)st.ref_frags, st.ref_norms, (const h8 *)w.qry_frags, st.kt, m, st.n, scores); break;
    }
    FTRY(hipGetLastError());
    FTRY(hipMemcpyAsync(qnorm_out, w.qry_norms, (size_t)m * sizeof(float), hipMemcpyDeviceToDevice, s));
    const int qblocks = ((m + 31) / 32 * 32 + 255) / 256;
    std::vector<unsigned> part((size_t)3 * qblocks);
    FTRY(hipMemcpyAsync(part.data(), w.qpart, part.size() * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    FTRY(hipStreamSynchronize(s));
    float amax = 0.0f;
    unsigned qbad = 0u;
    for (int b = 0; b < qblocks; ++b) {
        float v;
        memcpy(&v, &part[(size_t)3 * b], 4);
        amax = fmaxf(amax, v);
        qbad += part[(size_t)3 * b + 2];
    }
    const BoundConsts c = knn_bound_consts(st.k, st.kt, st.sigma, amax, st.bmax, st.nmax);
    consts[0] = st.sigma;
    consts[1] = c.eta;
    consts[2] = c.rho;
    consts[3] = amax;
    consts[4] = st.bmax;
    consts[5] = c.g2;
    consts[6] = c.gam;
    consts[7] = (double)qbad;  // #non-finite fp16 query coordinates
    (void)thr_out;
    (void)r;
    return hipSuccess;
}
