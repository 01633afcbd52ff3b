// knn_exact.hip — exact fused distance + argmin kernels for gfx950 (MI355X).
//
// Every kernel here evaluates the squared L2 distance with the arithmetic of the reference's
// serial path v0 (sources/src/core.cu:44-49): d = q - r; p = d * d; acc = acc + p for
// dimension 0..k-1, each operation rounded once to binary32 (the file is compiled with
// -ffp-contract=off and tests/test_build.py checks the code object holds no fused
// multiply-add), and folds (distance, global index) into a packed 64-bit key with an unsigned
// min, which reproduces v0's "first strict minimum" (core.cu:50-54) for any visiting order.
//
// Two shapes of kernel, chosen by m (host side: knn_exact_launch):
//   * qreg  — many queries: each lane keeps 2*QP queries' coordinates in VGPRs (packed in
//             float2 so sub/mul/add issue as v_pk_*_f32 on two queries at once) and every
//             wave streams its block's slice of references through SGPRs (wave-uniform
//             s_load of one AoS row = one k-vector).  The (min, argmin) pair lives in the
//             lane, so no cross-lane reduction; one 64-bit atomic min per query per block.
//             VALU-bound: 3k+3 lane-ops per (query, reference) pair.
//   * rlane — few queries: each lane owns one reference row per step (read once from HBM),
//             query coordinates are wave-uniform; wave64 shuffle reduction of the packed
//             key at the end.  HBM-bound for m <~ 10.
//
// This is not a translation of the reference's one-block-per-query kernels (core.cu:808-855):
// no SoA transpose pass, no per-block result array, no host second-level reduce.
#include "knn_exact_dev.h"

#include <math.h>
#include <stdlib.h>



// ------------------------------------------------------------------------------------------
// qreg: queries in VGPRs (pairs packed as float2), references streamed as wave-uniform rows.
//   grid.x = reference slices, grid.y = groups of KNN_WAVES * 128 * QP queries, block = 256.
// ------------------------------------------------------------------------------------------
template <int K, int QP>
__global__ __launch_bounds__(KNN_BLOCK) void knn_exact_qreg(const float *__restrict__ Q,
                                                            const float *__restrict__ R, int m,
                                                            long long n, long long base,
                                                            u64 *__restrict__ keys,
                                                            long long refs_per_block,
                                                            const unsigned *__restrict__ gate)
{
    if (gate && *gate == 0u)
        return;
    exact_qreg_body<K, QP>(Q, R, m, n, base, keys, refs_per_block, blockIdx.x, gridDim.x, blockIdx.y);
}

// Any k up to 16*KC without a compile-time K: the same scheme (queries in VGPRs, rows through
// wave-uniform scalar loads), dimensions in chunks of 16 with the last chunk guarded by uniform
// branches.  PACK = two queries per lane as a float2 (KC <= 4), else one (KC <= 8, k <= 128).  The
// generic row-per-lane kernel runs ~10x below this for m >= 48.
template <bool PACK> struct QregLane;
template <> struct QregLane<true> {
    typedef f2 T;
    static __device__ __forceinline__ f2 make(float a, float b) { return (f2){a, b}; }
    static __device__ __forceinline__ float second(const f2 &v) { return v.y; }
    static __device__ __forceinline__ float first(const f2 &v) { return v.x; }
};
template <> struct QregLane<false> {
    typedef float T;
    static __device__ __forceinline__ float make(float a, float) { return a; }
    static __device__ __forceinline__ float second(const float &) { return INFINITY; }
    static __device__ __forceinline__ float first(const float &v) { return v; }
};

template <int KC, bool PACK>
__global__ __launch_bounds__(KNN_BLOCK) void knn_exact_qregn(const float *__restrict__ Q,
                                                             const float *__restrict__ R, int k, int m,
                                                             long long n, long long base,
                                                             u64 *__restrict__ keys,
                                                             long long refs_per_block,
                                                             const unsigned *__restrict__ gate)
{
#pragma clang fp contract(off)
    if (gate && *gate == 0u)
        return;
    typedef QregLane<PACK> L;
    typedef typename L::T T;
    constexpr int KM = 16 * KC;
    constexpr int QL = PACK ? 2 : 1;  // queries per lane
    const int lane = threadIdx.x & (KNN_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int q0 = (blockIdx.y * KNN_WAVES + wave) * (QL * KNN_WAVE);
    if (q0 >= m)
        return;
    const int qa = min(q0 + lane, m - 1);
    const int qb = min(q0 + KNN_WAVE + lane, m - 1);
    T q[KM];
#pragma unroll
    for (int d = 0; d < KM; ++d)
        q[d] = L::make(d < k ? Q[(size_t)qa * k + d] : 0.0f, PACK && d < k ? Q[(size_t)qb * k + d] : 0.0f);
    float best0 = INFINITY, best1 = INFINITY;
    unsigned bidx0 = 0u, bidx1 = 0u;

    // Slices are strided over: an ordinary launch has one block per slice; the GATED launch (the filter's
    // device-side fallback, almost always a no-op) has at most ~5 resident blocks per CU, so looking at
    // the flag costs a few hundred blocks instead of a full grid queued on every query.
    for (long long i0 = (long long)blockIdx.x * refs_per_block; i0 < n; i0 += (long long)gridDim.x * refs_per_block) {
    const long long i1 = min(n, i0 + refs_per_block);
    unsigned gidx = (unsigned)(base + i0);  // global index of the current reference (low 32 bits)
    const float *__restrict__ r = R + (size_t)i0 * k;
    for (long long i = i0; i < i1; ++i, ++gidx, r += k) {
        T acc = L::make(0.0f, 0.0f);
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int rem = k - 16 * c;  // wave-uniform
            if (rem >= 16) {
                float rv[16];
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    rv[j] = r[16 * c + j];  // wave-uniform address -> scalar loads
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const T diff = q[16 * c + j] - L::make(rv[j], rv[j]);
                    const T sq = diff * diff;
                    acc = acc + sq;
                }
            } else if (rem > 0) {
#pragma unroll
                for (int j = 0; j < 15; ++j)
                    if (j < rem) {
                        const float rj = r[16 * c + j];
                        const T diff = q[16 * c + j] - L::make(rj, rj);
                        const T sq = diff * diff;
                        acc = acc + sq;
                    }
            }
        }
        if (best0 > L::first(acc)) {
            best0 = L::first(acc);
            bidx0 = gidx;
        }
        if (PACK && best1 > L::second(acc)) {
            best1 = L::second(acc);
            bidx1 = gidx;
        }
    }
    }  // slices
    if (q0 + lane < m && best0 < INFINITY) {
        const u64 key = pack_key(best0, bidx0);
        if (key < keys[q0 + lane])
            key_atomic_min(&keys[q0 + lane], key);
    }
    if (PACK && q0 + KNN_WAVE + lane < m && best1 < INFINITY) {
        const u64 key = pack_key(best1, bidx1);
        if (key < keys[q0 + KNN_WAVE + lane])
            key_atomic_min(&keys[q0 + KNN_WAVE + lane], key);
    }
}

// One query per lane, no packing: m in [~48, 128*KNN_WAVES) where pairs would idle lanes.
template <int K>
__global__ __launch_bounds__(KNN_BLOCK) void knn_exact_qreg1(const float *__restrict__ Q,
                                                             const float *__restrict__ R, int m,
                                                             long long n, long long base,
                                                             u64 *__restrict__ keys,
                                                             long long refs_per_block,
                                                             const unsigned *__restrict__ gate)
{
#pragma clang fp contract(off)
    if (gate && *gate == 0u)
        return;
    const int lane = threadIdx.x & (KNN_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int q0 = (blockIdx.y * KNN_WAVES + wave) * KNN_WAVE;
    if (q0 >= m)
        return;
    const int qi = q0 + lane;
    const int qc = min(qi, m - 1);
    float q[K];
#pragma unroll
    for (int d = 0; d < K; ++d)
        q[d] = Q[(size_t)qc * K + d];
    float best = INFINITY;
    unsigned bidx = 0u;

    // Slices are strided over: an ordinary launch has one block per slice; the GATED launch (the filter's
    // device-side fallback, almost always a no-op) has at most ~5 resident blocks per CU, so looking at
    // the flag costs a few hundred blocks instead of a full grid queued on every query.
    for (long long i0 = (long long)blockIdx.x * refs_per_block; i0 < n; i0 += (long long)gridDim.x * refs_per_block) {
    const long long i1 = min(n, i0 + refs_per_block);
    unsigned gidx = (unsigned)(base + i0);  // global index of the current reference (low 32 bits)
    const float *__restrict__ r = R + (size_t)i0 * K;
#pragma unroll 4
    for (long long i = i0; i < i1; ++i, ++gidx, r += K) {
        float acc = 0.0f;
#pragma unroll
        for (int d = 0; d < K; ++d) {
            const float diff = q[d] - r[d];
            const float sq = diff * diff;
            acc = acc + sq;
        }
        if (best > acc) {
            best = acc;
            bidx = gidx;
        }
    }
    }  // slices
    if (qi < m && best < INFINITY) {
        const u64 key = pack_key(best, bidx);
        if (key < keys[qi])
            key_atomic_min(&keys[qi], key);
    }
}

// ------------------------------------------------------------------------------------------
// rlane: one reference row per lane per step, QT wave-uniform queries per pass.
//   grid.x = blocks striding over the references, grid.y = ceil(m / QT), block = 256.
//   K > 0: compile-time dimension (rows read with the widest aligned loads the compiler
//   finds); K == 0: run-time k.
// ------------------------------------------------------------------------------------------

template <int K, int QT>
__global__ __launch_bounds__(KNN_BLOCK) void knn_exact_rlane(const float *__restrict__ Q,
                                                             const float *__restrict__ R, int krt,
                                                             int m, long long n, long long base,
                                                             u64 *__restrict__ keys,
                                                             const unsigned *__restrict__ gate,
                                                             const unsigned *__restrict__ gather)
{
#pragma clang fp contract(off)
    // gather != null: scan only the listed rows (the filter's outlier references), n = list length
    if (gate && *gate == 0u)
        return;
    const int k = K > 0 ? K : krt;
    const int nqt = (m + QT - 1) / QT;
    // grid.y may be smaller than the number of query tiles (the launch caps the grid so that a
    // gated no-op launch stays cheap): stride over them
    for (int qt = blockIdx.y; qt < nqt; qt += gridDim.y) {
    const int q0 = qt * QT;
    float best[QT];
    unsigned bidx[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        best[t] = INFINITY;
        bidx[t] = 0u;
    }
    const float *__restrict__ qrow[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t)
        qrow[t] = Q + (size_t)min(q0 + t, m - 1) * k;  // wave-uniform

    const long long stride = (long long)gridDim.x * KNN_BLOCK;
    for (long long i = (long long)blockIdx.x * KNN_BLOCK + threadIdx.x; i < n; i += stride) {
        const long long row = gather ? (long long)gather[i] : i;
        const float *__restrict__ r = R + (size_t)row * k;
        float acc[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t)
            acc[t] = 0.0f;
        if (K > 0) {
            float rv[K > 0 ? K : 1];
#pragma unroll
            for (int d = 0; d < K; ++d)
                rv[d] = r[d];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
#pragma unroll
                for (int d = 0; d < K; ++d) {
                    const float diff = qrow[t][d] - rv[d];
                    const float sq = diff * diff;
                    acc[t] = acc[t] + sq;
                }
            }
        } else {
            for (int d = 0; d < k; ++d) {
                const float rv = r[d];
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    const float diff = qrow[t][d] - rv;
                    const float sq = diff * diff;
                    acc[t] = acc[t] + sq;
                }
            }
        }
        const unsigned gidx = (unsigned)(base + row);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            if (best[t] > acc[t]) {
                best[t] = acc[t];
                bidx[t] = gidx;
            }
        }
    }

    const int lane = threadIdx.x & (KNN_WAVE - 1);
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        // A lane that never updated carries (+INF, 0) == kKeyInit, neutral under min.
        const u64 key = wave_min_u64(pack_key(best[t], bidx[t]));
        if (lane == 0 && q0 + t < m && key < kKeyInit && key < keys[q0 + t])
            key_atomic_min(&keys[q0 + t], key);
    }
    }  // query tiles
}

// ------------------------------------------------------------------------------------------
// rlane16: the k = 16, few-queries (HBM-bound) case with LDS-staged reference tiles.  A block
// reads 256 rows = 16 KiB as fully coalesced 16-byte chunks (lane l of a wave takes chunk l: one
// 1 KiB burst per wave-instruction instead of 64 partial lines), parks them in LDS with an XOR
// swizzle of the chunk index, and every lane reads back its own row with four conflict-free
// ds_read_b128.  The next tile's global loads are issued before the current tile is consumed.
// Arithmetic, tie-break and reduction are those of knn_exact_rlane.
// ------------------------------------------------------------------------------------------
typedef float f4x __attribute__((ext_vector_type(4)));

template <int QT>
__global__ __launch_bounds__(KNN_BLOCK) void knn_exact_rlane16(const float *__restrict__ Q,
                                                               const f4x *__restrict__ R4, int m,
                                                               long long n, long long base,
                                                               u64 *__restrict__ keys,
                                                               const unsigned *__restrict__ gate)
{
#pragma clang fp contract(off)
    constexpr int K = 16;
    __shared__ f4x s_x[KNN_BLOCK * 4];
    if (gate && *gate == 0u)
        return;
    const int tid = threadIdx.x;
    const int q0 = blockIdx.y * QT;
    float qv[QT][K];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int d = 0; d < K; ++d)
            qv[t][d] = Q[(size_t)min(q0 + t, m - 1) * K + d];  // wave-uniform -> SGPRs
    float best[QT];
    unsigned bidx[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        best[t] = INFINITY;
        bidx[t] = 0u;
    }
    const long long ntiles = (n + KNN_BLOCK - 1) / KNN_BLOCK;
    const long long nchunks = n * 4;  // 16-byte chunks in the shard
    const int sw = (tid >> 2) & 3;

    f4x pre[4];
    long long tile = blockIdx.x;
    if (tile < ntiles) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long c = tile * (KNN_BLOCK * 4) + tid + j * KNN_BLOCK;
            pre[j] = c < nchunks ? __builtin_nontemporal_load(&R4[c]) : (f4x){0.f, 0.f, 0.f, 0.f};
        }
    }
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();  // the previous tile's rows have been consumed
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = tid + j * KNN_BLOCK;
            const int rr = c >> 2, cc = c & 3;
            s_x[rr * 4 + (cc ^ ((rr >> 2) & 3))] = pre[j];
        }
        __syncthreads();
        const long long nxt = tile + gridDim.x;
        if (nxt < ntiles) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long c = nxt * (KNN_BLOCK * 4) + tid + j * KNN_BLOCK;
                pre[j] = c < nchunks ? __builtin_nontemporal_load(&R4[c]) : (f4x){0.f, 0.f, 0.f, 0.f};
            }
        }
        const long long i = tile * KNN_BLOCK + tid;
        float rv[K];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const f4x x = s_x[tid * 4 + (cc ^ sw)];
            rv[4 * cc + 0] = x[0];
            rv[4 * cc + 1] = x[1];
            rv[4 * cc + 2] = x[2];
            rv[4 * cc + 3] = x[3];
        }
        if (i < n) {
            const unsigned gidx = (unsigned)(base + i);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                float acc = 0.0f;
#pragma unroll
                for (int d = 0; d < K; ++d) {
                    const float diff = qv[t][d] - rv[d];
                    const float sq = diff * diff;
                    acc = acc + sq;
                }
                if (best[t] > acc) {
                    best[t] = acc;
                    bidx[t] = gidx;
                }
            }
        }
    }
    const int lane = tid & (KNN_WAVE - 1);
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const u64 key = wave_min_u64(pack_key(best[t], bidx[t]));
        if (lane == 0 && q0 + t < m && key < kKeyInit && key < keys[q0 + t])
            key_atomic_min(&keys[q0 + t], key);
    }
}

// ------------------------------------------------------------------------------------------
// Exact re-rank of the MFMA filter's candidate records.  A record names one query and the 16
// references one lane of a 32x32 MFMA tile covers: (query << 32) | (ref_tile << 1) | half, rows
// 8g + 4*half + i (g, i in 0..3) of that tile.  One (record, row) pair per thread, v0 arithmetic.
// ------------------------------------------------------------------------------------------
template <int K, bool LPW = false>  // K > 0: compile-time dimension (all row loads issue before the first use)
__global__ __launch_bounds__(KNN_BLOCK) void knn_rerank_kernel(const float *__restrict__ Q,
                                                               const float *__restrict__ R, int krt,
                                                               long long n, long long base,
                                                               const u64 *__restrict__ rec,
                                                               const unsigned short *__restrict__ rec_rows,
                                                               const unsigned *__restrict__ counts, unsigned nlists,
                                                               unsigned slice, unsigned *__restrict__ ctl,
                                                               u64 *__restrict__ keys, RerankPieces pieces,
                                                               const unsigned *__restrict__ perm,
                                                               unsigned ovf_base, unsigned ovf_cap)
{
    // LPW = true (the cell-pruned path: ~3000 records over 6144 lists, most of them empty): one WAVE per record list, four
    // lists per block — a quarter of the blocks and of the registers the launch holds while the next batch's kernels
    // wait for room; else one block per list (= per filter wave).  The list's piece gives its first query.
    const unsigned list_id = LPW ? blockIdx.x * (KNN_BLOCK / KNN_WAVE) + (threadIdx.x >> 6) : blockIdx.x;
    const unsigned tid_l = LPW ? (threadIdx.x & 63u) : threadIdx.x;       // thread inside the group that owns the list
    constexpr unsigned GROUP = LPW ? KNN_WAVE : KNN_BLOCK;
    unsigned qrow_base = pieces.qrow_base[0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (list_id >= pieces.list_base[i])
            qrow_base = pieces.qrow_base[i];
    if (ctl[KNN_CTL_FALLBACK] != 0u)
        return;  // the exact scan is going to run anyway: do not re-rank a truncated candidate set
    if (ovf_cap != 0u && ctl[KNN_CTL_RECORDS] > ovf_cap) {
        // (cell-pruned path) more candidates than the buffers hold: the fp16 scores do not separate this batch's rows (a
        // cluster tighter than the fp16 step).  Its listed (cell, query) pairs are evaluated exactly instead
        // (knn_cells_exact_kernel) — not the whole shard, as round 2 did — and the records are left alone
        if (blockIdx.x == 0 && threadIdx.x == 0)
            ctl[KNN_CTL_EXACT_CELLS] = 1u;
        return;
    }
    const int k = K > 0 ? K : krt;
    if (list_id < nlists) {
        const unsigned want = counts[list_id];
        const unsigned nrec = min(want, slice);
        // (no shared record counter here: 2048 blocks adding to one word cost ~23 us; the host sums
        // counts[] when statistics are asked for)
        if (tid_l == 0 && want > slice)
            ctl[KNN_CTL_FALLBACK] = 1u;  // candidates were dropped: the gated exact scan takes over
        const u64 *__restrict__ list = rec + (size_t)list_id * slice;
        // 16 consecutive lanes share one record (one query): their keys are min-folded with shuffles
        // and ONE guarded atomic is issued per record (the keys sit on a handful of cache lines; the
        // unguarded 16-per-record form spent ~1 ms in atomic contention at 270k records).
        const unsigned total = (nrec * 16u + GROUP - 1) / GROUP * GROUP;
        // Four chunks of pairs per trip: a pair is a chain of dependent loads (record -> perm -> rows), and a wave that owns a
        // long list (skewed data: 136k records over 6144 lists, the longest in the thousands) walked it one round trip at a
        // time — 0.084 ms for that batch's re-rank, 0.072 with four in flight (a block of four waves per list: 0.062 — it is
        // the few very long lists, not the width).  Short lists (the usual case) make one trip either way.
        constexpr unsigned U = LPW ? 4u : 1u;
        for (unsigned c0 = tid_l; c0 < total; c0 += GROUP * U) {
            u64 key[U];
            unsigned qi[U];
#pragma unroll
            for (unsigned u = 0; u < U; ++u) {
                const unsigned c = c0 + u * GROUP;
                key[u] = ~0ull;
                qi[u] = 0u;
                if (c < nrec * 16u) {
                    const unsigned rmask = rec_rows ? rec_rows[(size_t)list_id * slice + (c >> 4)] : 0xFFFFu;
                    key[u] = rerank_pair<K>(Q, R, k, n, base, list[c >> 4], c & 15u, rmask, qrow_base, perm, qi[u]);
                }
            }
#pragma unroll
            for (unsigned u = 0; u < U; ++u) {
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) {
                    const u64 o = __shfl_xor(key[u], off, KNN_WAVE);
                    key[u] = o < key[u] ? o : key[u];
                }
                // keys[] only ever decreases, so a stale (larger) read can only cause a spare atomic
                if ((threadIdx.x & 15u) == 0u && key[u] != ~0ull && key[u] < keys[qi[u]])
                    key_atomic_min(&keys[qi[u]], key[u]);
            }
        }
    }
    // the shared overflow area (cell-pruned path: what did not fit a wave's slice), all blocks striding over it
    if (ovf_cap != 0u) {
        const unsigned have = ctl[KNN_CTL_RECORDS];
        if (have != 0u) {   // block-uniform
            const u64 *__restrict__ ovf = rec + ovf_base;
            const unsigned pairs = have * 16u;
            const unsigned padded = (pairs + KNN_BLOCK - 1) / KNN_BLOCK * KNN_BLOCK;
            for (unsigned c = blockIdx.x * KNN_BLOCK + threadIdx.x; c < padded; c += gridDim.x * KNN_BLOCK) {
                u64 key = ~0ull;
                unsigned qi = 0u;
                if (c < pairs)
                    key = rerank_pair<K>(Q, R, k, n, base, ovf[c >> 4], c & 15u, 0xFFFFu, 0u, perm, qi);
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

// Re-rank for records that carry a row mask (deep-K filter): ONE THREAD per record walks the set bits
// of its mask.  With the 16-lanes-per-record form above only ~1 lane in 16 had a row to evaluate
// (the mask usually has a single bit), and at k = 128 a row is a serial chain of 128 subtract /
// multiply / add steps: 0.33 ms for C5's 640k records, most lanes idle.
template <int K>
__device__ __forceinline__ float v0_row_distance(const float *__restrict__ q, const float *__restrict__ r, int krt)
{
#pragma clang fp contract(off)
    const int k = K > 0 ? K : krt;
    float acc = 0.0f;
    int d = 0;
    for (; d + 16 <= k; d += 16) {  // chunks of 16 with all 32 loads in flight; order stays d = 0..k-1
        float qv[16], rv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            qv[j] = q[d + j];
            rv[j] = r[d + j];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float diff = qv[j] - rv[j];
            const float sq = diff * diff;
            acc = acc + sq;
        }
    }
    for (; d < k; ++d) {
        const float diff = q[d] - r[d];
        const float sq = diff * diff;
        acc = acc + sq;
    }
    return acc;
}

__global__ __launch_bounds__(KNN_BLOCK) void knn_rerank_rows_kernel(const float *__restrict__ Q,
                                                                    const float *__restrict__ R, int k,
                                                                    long long n, long long base,
                                                                    const u64 *__restrict__ rec,
                                                                    const unsigned short *__restrict__ rec_rows,
                                                                    const unsigned *__restrict__ counts,
                                                                    unsigned slice, unsigned *__restrict__ ctl,
                                                                    u64 *__restrict__ keys)
{
    if (ctl[KNN_CTL_FALLBACK] != 0u)
        return;
    const unsigned want = counts[blockIdx.x];
    const unsigned nrec = min(want, slice);
    if (threadIdx.x == 0 && want > slice)
        ctl[KNN_CTL_FALLBACK] = 1u;
    const u64 *__restrict__ list = rec + (size_t)blockIdx.x * slice;
    const unsigned short *__restrict__ rows = rec_rows + (size_t)blockIdx.x * slice;
    for (unsigned c = threadIdx.x; c < nrec; c += KNN_BLOCK) {
        const u64 e = list[c];
        unsigned rm = rows[c];
        const unsigned qi = (unsigned)(e >> 32);
        const unsigned lo = (unsigned)(e & 0xFFFFFFFFull);
        const float *__restrict__ q = Q + (size_t)qi * k;
        u64 key = ~0ull;
        while (rm) {
            const unsigned reg = (unsigned)__builtin_ctz(rm);
            rm &= rm - 1u;
            const long long ri = (long long)(lo >> 1) * 32 + 8 * (reg >> 2) + 4 * (lo & 1u) + (reg & 3u);
            if (ri < n) {
                const float acc = v0_row_distance<0>(q, R + (size_t)ri * k, k);
                if (acc < INFINITY) {  // false for NaN too: v0 never selects those
                    const u64 cand = pack_key(acc, (unsigned)(base + ri));
                    key = cand < key ? cand : key;
                }
            }
        }
        if (key != ~0ull && key < keys[qi])
            key_atomic_min(&keys[qi], key);
    }
}

// ------------------------------------------------------------------------------------------
// Small utility kernels.
// ------------------------------------------------------------------------------------------
__global__ void knn_keys_fill_kernel(u64 *keys, int m)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m)
        keys[i] = kKeyInit;
}

__global__ void knn_keys_unpack_kernel(const u64 *__restrict__ keys, int m, int *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m)
        out[i] = (int)(unsigned)(keys[i] & 0xFFFFFFFFull);
}

__global__ void knn_synth_fill_kernel(float *__restrict__ dst, long long count, u64 seed,
                                      long long first)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        u64 z = seed + ((u64)(first + i) + 1ull) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        dst[i] = (float)(unsigned)(z >> 40) * 0x1.0p-24f;  // < 2^24: exact
    }
}

// ------------------------------------------------------------------------------------------
// Host-side dispatch.
// ------------------------------------------------------------------------------------------
namespace {

struct SliceGeom {
    long long refs_per_block;
    unsigned nslices;
};

// Slice the references so the grid has ~20 blocks per CU (each block 4 waves) but a slice is
// never shorter than `min_refs` rows.  32 rows, not more: a block walks its slice serially at
// ~0.6 us per row (4 queries per lane), so 512-row slices put a 0.3 ms floor under every launch
// with n <= 2.6M — (16, 1024, 1024) took 0.42 ms, 0.10 ms with 32-row slices.
SliceGeom slice_refs(long long n, unsigned qgroups, int num_cu, long long min_refs)
{
    long long want = (long long)num_cu * 20 / (qgroups ? qgroups : 1);  // 4-5 rounds of blocks: small tail
    if (want < 1)
        want = 1;
    long long per = (n + want - 1) / want;
    if (per < min_refs)
        per = min_refs;
    SliceGeom g;
    g.refs_per_block = per;
    g.nslices = (unsigned)((n + per - 1) / per);
    return g;
}

// Blocks along the slice axis: one per slice, except for a gated (almost always no-op) launch, which
// gets what is resident at once (5 blocks per CU at these kernels' register counts) and strides.
unsigned slice_grid(const SliceGeom &g, unsigned qgroups, int num_cu, const unsigned *gate)
{
    if (!gate)
        return g.nslices;
    unsigned cap = (unsigned)num_cu * 5u / (qgroups ? qgroups : 1u);
    if (cap < 1u)
        cap = 1u;
    return g.nslices < cap ? g.nslices : cap;
}

template <int K>
hipError_t launch_qreg_k(int m, long long n, long long base, const float *q, const float *r,
                         u64 *keys, int num_cu, const unsigned *gate, hipStream_t s)
{
    // queries per block: 1024 (QP=2) when m fills most of it, else 512 (QP=1), else 256 (unpacked)
    if (m > 3 * KNN_WAVE * KNN_WAVES) {
        const unsigned qg = (unsigned)knn_divup(m, KNN_WAVES * 4 * KNN_WAVE);
        const SliceGeom g = slice_refs(n, qg, num_cu, 32);
        hipLaunchKernelGGL((knn_exact_qreg<K, 2>), dim3(slice_grid(g, qg, num_cu, gate), qg), dim3(KNN_BLOCK), 0, s, q, r,
                           m, n, base, keys, g.refs_per_block, gate);
    } else if (m > KNN_WAVE * KNN_WAVES) {
        const unsigned qg = (unsigned)knn_divup(m, KNN_WAVES * 2 * KNN_WAVE);
        const SliceGeom g = slice_refs(n, qg, num_cu, 32);
        hipLaunchKernelGGL((knn_exact_qreg<K, 1>), dim3(slice_grid(g, qg, num_cu, gate), qg), dim3(KNN_BLOCK), 0, s, q, r,
                           m, n, base, keys, g.refs_per_block, gate);
    } else {
        const unsigned qg = (unsigned)knn_divup(m, KNN_WAVES * KNN_WAVE);
        const SliceGeom g = slice_refs(n, qg, num_cu, 32);
        hipLaunchKernelGGL((knn_exact_qreg1<K>), dim3(slice_grid(g, qg, num_cu, gate), qg), dim3(KNN_BLOCK), 0, s, q, r, m,
                           n, base, keys, g.refs_per_block, gate);
    }
    return hipGetLastError();
}

template <int K>
hipError_t launch_rlane_k(int k, int m, long long n, long long base, const float *q, const float *r,
                          u64 *keys, int num_cu, const unsigned *gate, hipStream_t s)
{
    constexpr int QT = 4;
    if (K == 16 && m <= 2 * QT && n >= 4096 && ((uintptr_t)r & 15u) == 0) {
        // HBM-bound shape: LDS-staged, fully coalesced reference tiles
        // query tile sized to the batch: a lone query should not pay for four (the spare VALU work
        // costs power, hence clock, hence bandwidth)
        const int qt16 = m == 1 ? 1 : m == 2 ? 2 : QT;
        const unsigned gy = (unsigned)knn_divup(m, qt16);
        long long gx = (n + KNN_BLOCK - 1) / KNN_BLOCK;
        const long long capx = (long long)num_cu * 8 / gy;
        if (gx > capx)
            gx = capx;
        if (qt16 == 1)
            hipLaunchKernelGGL((knn_exact_rlane16<1>), dim3((unsigned)gx, gy), dim3(KNN_BLOCK), 0, s, q,
                               (const f4x *)r, m, n, base, keys, gate);
        else if (qt16 == 2)
            hipLaunchKernelGGL((knn_exact_rlane16<2>), dim3((unsigned)gx, gy), dim3(KNN_BLOCK), 0, s, q,
                               (const f4x *)r, m, n, base, keys, gate);
        else
            hipLaunchKernelGGL((knn_exact_rlane16<QT>), dim3((unsigned)gx, gy), dim3(KNN_BLOCK), 0, s, q,
                               (const f4x *)r, m, n, base, keys, gate);
        return hipGetLastError();
    }
    unsigned qt = (unsigned)knn_divup(m, QT);
    if (qt > 64u)
        qt = 64u;  // the kernel strides over the remaining query tiles
    long long blocks = (n + KNN_BLOCK - 1) / KNN_BLOCK;
    long long cap = (long long)num_cu * 8 / qt;
    if (cap < 4)
        cap = 4;
    if (blocks > cap)
        blocks = cap;
    if (blocks < 1)
        blocks = 1;
    hipLaunchKernelGGL((knn_exact_rlane<K, QT>), dim3((unsigned)blocks, qt), dim3(KNN_BLOCK), 0, s, q,
                       r, k, m, n, base, keys, gate, (const unsigned *)nullptr);
    return hipGetLastError();
}

}  // namespace

hipError_t knn_exact_launch(int k, int m, long long n, long long base, const float *q,
                            const float *r, u64 *keys, int num_cu, const unsigned *gate, hipStream_t s)
{
    if (n <= 0 || m <= 0)
        return hipSuccess;
    // Few queries: stream references once per 4 queries (HBM-bound).  Many: qreg (VALU-bound).
    const bool many = m >= 48;
    if (many) {
        switch (k) {
        case 1: return launch_qreg_k<1>(m, n, base, q, r, keys, num_cu, gate, s);
        case 2: return launch_qreg_k<2>(m, n, base, q, r, keys, num_cu, gate, s);
        case 3: return launch_qreg_k<3>(m, n, base, q, r, keys, num_cu, gate, s);
        case 4: return launch_qreg_k<4>(m, n, base, q, r, keys, num_cu, gate, s);
        case 8: return launch_qreg_k<8>(m, n, base, q, r, keys, num_cu, gate, s);
        case 16: return launch_qreg_k<16>(m, n, base, q, r, keys, num_cu, gate, s);
        default: break;
        }
        if (k <= 128 && n >= 2048) {  // no compile-time K: chunked run-time-k form of the same kernel
            // (below ~2k rows its query prologue, 16*KC loads per lane, costs more than the scan)
            const int kc = (k + 15) / 16;
            const bool pack = kc <= 4;
            const unsigned qg = (unsigned)knn_divup(m, KNN_WAVES * (pack ? 2 : 1) * KNN_WAVE);
            const SliceGeom g = slice_refs(n, qg, num_cu, 32);
            const dim3 grid(slice_grid(g, qg, num_cu, gate), qg), block(KNN_BLOCK);
#define KNN_QREGN(KCV, PK)                                                                                  \
    hipLaunchKernelGGL((knn_exact_qregn<KCV, PK>), grid, block, 0, s, q, r, k, m, n, base, keys, g.refs_per_block, gate)
            switch (kc) {
            case 1: KNN_QREGN(1, true); break;
            case 2: KNN_QREGN(2, true); break;
            case 3: KNN_QREGN(3, true); break;
            case 4: KNN_QREGN(4, true); break;
            case 5: KNN_QREGN(5, false); break;
            case 6: KNN_QREGN(6, false); break;
            case 7: KNN_QREGN(7, false); break;
            default: KNN_QREGN(8, false); break;
            }
#undef KNN_QREGN
            return hipGetLastError();
        }
    }
    switch (k) {
    case 1: return launch_rlane_k<1>(k, m, n, base, q, r, keys, num_cu, gate, s);
    case 2: return launch_rlane_k<2>(k, m, n, base, q, r, keys, num_cu, gate, s);
    case 3: return launch_rlane_k<3>(k, m, n, base, q, r, keys, num_cu, gate, s);
    case 4: return launch_rlane_k<4>(k, m, n, base, q, r, keys, num_cu, gate, s);
    case 8: return launch_rlane_k<8>(k, m, n, base, q, r, keys, num_cu, gate, s);
    case 16: return launch_rlane_k<16>(k, m, n, base, q, r, keys, num_cu, gate, s);
    default: return launch_rlane_k<0>(k, m, n, base, q, r, keys, num_cu, gate, s);
    }
}

hipError_t knn_exact_gather_launch(int k, int m, unsigned count, long long base, const float *q, const float *r,
                                   const unsigned *list, u64 *keys, int num_cu, const unsigned *gate, hipStream_t s)
{
    // exact scan of an explicit row list (listed rows in any order: the packed-key min does not care)
    if (count == 0 || m <= 0)
        return hipSuccess;
    constexpr int QT = 4;
    unsigned qt = (unsigned)knn_divup(m, QT);
    if (qt > 256u)
        qt = 256u;
    long long blocks = ((long long)count + KNN_BLOCK - 1) / KNN_BLOCK;
    long long cap = (long long)num_cu * 8 / qt;
    if (cap < 1)
        cap = 1;
    if (blocks > cap)
        blocks = cap;
    const dim3 grid((unsigned)blocks, qt);
    switch (k) {
    case 3: hipLaunchKernelGGL((knn_exact_rlane<3, QT>), grid, dim3(KNN_BLOCK), 0, s, q, r, k, m, (long long)count, base, keys, gate, list); break;
    case 8: hipLaunchKernelGGL((knn_exact_rlane<8, QT>), grid, dim3(KNN_BLOCK), 0, s, q, r, k, m, (long long)count, base, keys, gate, list); break;
    case 16: hipLaunchKernelGGL((knn_exact_rlane<16, QT>), grid, dim3(KNN_BLOCK), 0, s, q, r, k, m, (long long)count, base, keys, gate, list); break;
    default: hipLaunchKernelGGL((knn_exact_rlane<0, QT>), grid, dim3(KNN_BLOCK), 0, s, q, r, k, m, (long long)count, base, keys, gate, list); break;
    }
    return hipGetLastError();
}

hipError_t knn_rerank_launch(int k, long long n, const float *q, const float *r, long long base,
                             const u64 *rec, const unsigned short *rec_rows, const unsigned *counts,
                             unsigned nlists, unsigned slice, unsigned *ctl, u64 *keys, RerankPieces pieces,
                             hipStream_t s, const unsigned *perm, unsigned ovf_base, unsigned ovf_cap)
{
    if (nlists == 0)
        return hipSuccess;
    if (rec_rows && perm)
        return hipErrorInvalidValue;  // row masks belong to the deep-K scan, cell-sorted layouts to k <= 16
    if (rec_rows) {  // deep-K scans are never cut into pieces
        hipLaunchKernelGGL(knn_rerank_rows_kernel, dim3(nlists), dim3(KNN_BLOCK), 0, s, q, r, k, n, base, rec, rec_rows,
                           counts, slice, ctl, keys);
        return hipGetLastError();
    }
#define KNN_RERANK(KK)                                                                                     \
    hipLaunchKernelGGL(knn_rerank_kernel<KK>, dim3(nlists), dim3(KNN_BLOCK), 0, s, q, r, k, n, base, rec, rec_rows, \
                       counts, nlists, slice, ctl, keys, pieces, perm, ovf_base, ovf_cap)
#define KNN_RERANK_LPW(KK)                                                                                 \
    hipLaunchKernelGGL((knn_rerank_kernel<KK, true>), dim3((nlists + KNN_WAVES - 1) / KNN_WAVES), dim3(KNN_BLOCK), 0, s, q, r, k, n, \
                       base, rec, rec_rows, counts, nlists, slice, ctl, keys, pieces, perm, ovf_base, ovf_cap)
    if (perm) {   // cell-sorted layout = the pruned scan's records: few, spread thin
        switch (k) {
        case 3: KNN_RERANK_LPW(3); break;
        case 4: KNN_RERANK_LPW(4); break;
        case 8: KNN_RERANK_LPW(8); break;
        case 16: KNN_RERANK_LPW(16); break;
        default: KNN_RERANK_LPW(0); break;
        }
        return hipGetLastError();
    }
    switch (k) {
    case 3: KNN_RERANK(3); break;
    case 4: KNN_RERANK(4); break;
    case 8: KNN_RERANK(8); break;
    case 16: KNN_RERANK(16); break;
    default: KNN_RERANK(0); break;
    }
#undef KNN_RERANK
#undef KNN_RERANK_LPW
    return hipGetLastError();
}

hipError_t knn_keys_fill_launch(u64 *keys, int m, hipStream_t s)
{
    if (m <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(knn_keys_fill_kernel, dim3(knn_divup(m, 256)), dim3(256), 0, s, keys, m);
    return hipGetLastError();
}

hipError_t knn_keys_unpack_launch(const u64 *keys, int m, int *out, hipStream_t s)
{
    if (m <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(knn_keys_unpack_kernel, dim3(knn_divup(m, 256)), dim3(256), 0, s, keys, m, out);
    return hipGetLastError();
}

hipError_t knn_synth_fill_launch(float *dst, long long count, u64 seed, long long first, hipStream_t s)
{
    if (count <= 0)
        return hipSuccess;
    long long blocks = (count + 255) / 256;
    if (blocks > 8192)
        blocks = 8192;
    hipLaunchKernelGGL(knn_synth_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, count, seed,
                       first);
    return hipGetLastError();
}
