// knn_exact_dev.h — device code the exact kernels (knn_exact.hip) and the cell-pruned path (knn_cells.hip) share:
// packed keys, the exact v0 arithmetic on one (query, row) pair of a candidate record, and the exact evaluation of the
// listed (item, query) pairs.  Everything here follows the reference's serial path (sources/src/core.cu:44-54): d = q - r;
// p = d * d; acc = acc + p in dimension order, one rounding per operation (the translation units are compiled with
// -ffp-contract=off and a CPU test disassembles the kernels that include this file), unsigned min of
// (distance bits << 32 | index) = first strict minimum.
#pragma once

#include "knn_common.h"

#include <math.h>

#define KNN_BLOCK 256
#define KNN_WAVES (KNN_BLOCK / KNN_WAVE)

__device__ __forceinline__ u64 pack_key(float d2, unsigned idx)
{
    return ((u64)__float_as_uint(d2) << 32) | (u64)idx;
}

__device__ __forceinline__ void key_atomic_min(u64 *p, u64 key)
{
    // gfx950: global_atomic_umin_x2, device scope (all XCDs).
    __hip_atomic_fetch_min(p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ u64 wave_min_u64(u64 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const u64 o = __shfl_xor(v, off, KNN_WAVE);
        v = o < v ? o : v;
    }
    return v;
}

// One (record, row) pair: the packed key of row `reg` of record e for its query, ~0 when there is nothing to evaluate.
template <int K>
__device__ __forceinline__ u64 rerank_pair(const float *__restrict__ Q, const float *__restrict__ R, int k, long long n,
                                           long long base, u64 e, unsigned reg, unsigned rmask, unsigned qrow_base,
                                           const unsigned *__restrict__ perm, unsigned &qi)
{
#pragma clang fp contract(off)
    qi = (unsigned)(e >> 32) + qrow_base;
    const unsigned lo = (unsigned)(e & 0xFFFFFFFFull);
    long long ri = (long long)(lo >> 1) * 32 + 8 * (reg >> 2) + 4 * (lo & 1u) + (reg & 3u);
    // rows the filter already proved to be above the threshold are not the answer: skip them
    bool live = ri < n && ((rmask >> reg) & 1u);
    if (perm && live) {  // cell-sorted layout: position -> row, padding positions hold ~0u
        const unsigned row = perm[ri];
        live = row != 0xFFFFFFFFu;
        ri = (long long)row;
    }
    if (!live)
        return ~0ull;
    const float *__restrict__ q = Q + (size_t)qi * k;
    const float *__restrict__ r = R + (size_t)ri * k;
    float acc = 0.0f;
    if (K > 0) {
        float qv[K > 0 ? K : 1], rv[K > 0 ? K : 1];
#pragma unroll
        for (int d = 0; d < K; ++d) {
            qv[d] = q[d];
            rv[d] = r[d];
        }
#pragma unroll
        for (int d = 0; d < K; ++d) {
            const float diff = qv[d] - rv[d];
            const float sq = diff * diff;
            acc = acc + sq;
        }
    } else {
        // run-time k: chunks of 16 with all 32 loads of a chunk in flight together
        // (a plain scalar loop is one dependent round trip per dimension); the
        // accumulation order stays d = 0..k-1
        int d = 0;
        for (; d + 16 <= k; d += 16) {
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
    }
    if (acc < INFINITY)  // false for NaN too: v0 never selects those
        return pack_key(acc, (unsigned)(base + ri));
    return ~0ull;
}


// The body of knn_exact_qreg (knn_exact.hip): block (bx of gx, by) scans slices bx, bx + gx, ... of refs_per_block rows for
// the queries of group `by`.  Shared with the cell-pruned path's tail kernel, which runs it when a batch has a query nothing
// bounds (no separate gated launch for k = 16).
typedef float f2 __attribute__((ext_vector_type(2)));
template <int K, int QP>
__device__ __forceinline__ void exact_qreg_body(const float *__restrict__ Q, const float *__restrict__ R, int m, long long n,
                                                long long base, u64 *__restrict__ keys, long long refs_per_block,
                                                unsigned bx, unsigned gx, unsigned by)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x & (KNN_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int q0 = ((int)by * KNN_WAVES + wave) * (2 * QP * KNN_WAVE);
    if (q0 >= m)
        return;

    f2 q[QP][K];
    f2 best[QP];
    unsigned bidx[2 * QP];
#pragma unroll
    for (int p = 0; p < QP; ++p) {
        const int qa = min(q0 + (2 * p) * KNN_WAVE + lane, m - 1);
        const int qb = min(q0 + (2 * p + 1) * KNN_WAVE + lane, m - 1);
#pragma unroll
        for (int d = 0; d < K; ++d) {
            q[p][d].x = Q[(size_t)qa * K + d];
            q[p][d].y = Q[(size_t)qb * K + d];
        }
        best[p].x = INFINITY;
        best[p].y = INFINITY;
        bidx[2 * p] = 0u;
        bidx[2 * p + 1] = 0u;
    }

    // Slices are strided over: an ordinary launch has one block per slice; the GATED launch (the filter's
    // device-side fallback, almost always a no-op) has at most ~5 resident blocks per CU, so looking at
    // the flag costs a few hundred blocks instead of a full grid queued on every query.
    for (long long i0 = (long long)bx * refs_per_block; i0 < n; i0 += (long long)gx * refs_per_block) {
    const long long i1 = min(n, i0 + refs_per_block);
    unsigned gidx = (unsigned)(base + i0);  // global index of the current reference (low 32 bits)
    const float *__restrict__ r = R + (size_t)i0 * K;

#pragma unroll 2
    for (long long i = i0; i < i1; ++i, ++gidx, r += K) {
        float rv[K];
#pragma unroll
        for (int d = 0; d < K; ++d)
            rv[d] = r[d];  // wave-uniform address -> scalar load
#pragma unroll
        for (int p = 0; p < QP; ++p) {
            f2 acc = {0.0f, 0.0f};
#pragma unroll
            for (int d = 0; d < K; ++d) {
                const f2 rr = {rv[d], rv[d]};
                const f2 diff = q[p][d] - rr;
                const f2 sq = diff * diff;
                acc = acc + sq;
            }
            if (best[p].x > acc.x) {
                best[p].x = acc.x;
                bidx[2 * p] = gidx;
            }
            if (best[p].y > acc.y) {
                best[p].y = acc.y;
                bidx[2 * p + 1] = gidx;
            }
        }
    }
    }  // slices

#pragma unroll
    for (int p = 0; p < QP; ++p) {
        const int qa = q0 + (2 * p) * KNN_WAVE + lane;
        const int qb = q0 + (2 * p + 1) * KNN_WAVE + lane;
        // best == +INF means no reference beat +INF: leave the key alone (v0 keeps index 0).
        // (keys[] only decreases: a stale plain read can only cause a spare atomic, never skip one)
        if (qa < m && best[p].x < INFINITY) {
            const u64 key = pack_key(best[p].x, bidx[2 * p]);
            if (key < keys[qa])
                key_atomic_min(&keys[qa], key);
        }
        if (qb < m && best[p].y < INFINITY) {
            const u64 key = pack_key(best[p].y, bidx[2 * p + 1]);
            if (key < keys[qb])
                key_atomic_min(&keys[qb], key);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Exact evaluation of the cell-pruned path's listed (cell, query) pairs — what a batch falls back to when the fp16 scores
// cannot tell its candidates apart (rows of a cluster tighter than the fp16 step: every row of the query's cells passes the
// threshold and the record buffers overflow).  The geometry still rules out every cell it ruled out before (the lower
// bounds are exact arithmetic on the fp32 coordinates), so the exact v0 arithmetic runs over the same (item, listed query)
// pairs the MFMA scan visited instead of over the whole shard: for 64 clusters of 2^16 rows that is 1/64 of the pairs.
// Wave `wave` of `nwaves` takes items wave, wave + nwaves, ...; per pass its lanes hold CX_ROWS rows each (fetched through perm: layout position ->
// row) and walk the list — the query's coordinates are wave-uniform loads.  Called by knn_cells_tail_kernel (knn_cells.hip)
// when it finds the shared overflow area over-full.  K > 0: compile-time dimension; 0: run-time k <= 16.
// ------------------------------------------------------------------------------------------
#define CX_ROWS 4

// ---- the self-listing scan (round 5) -------------------------------------------------------------------------------
// A cell's list of queries — those of the batch that cannot rule the cell out — made by the wave that is about to score
// the cell instead of by a match launch in front of the scan: knn_cells_match_kernel's test with the same arithmetic
// (fl32(lo + hi) against Dup; lo, hi rounded down and Dup up with slack by the prep kernel) over ALL queries of the batch,
// 64 per step on the lanes, survivors compacted into the wave's own LDS room in query order.  The pruning tables come
// from the L2: a cell's row of the low table is contiguous (prep's `lo_by_entry` layout: [entry][query]) like its row of the
// high table, 2 x 4 KiB per cell at 1024 queries.  Saves the match launch (9 us at 2^13 cells, 18 at 2^16, one batch at a
// time), its lists' trip through memory and one dependent round trip per item (list length -> tiles).
#define CELL_SELF_CAP 256u     // list entries a wave's LDS room holds (with the Dup values the scan stays within 64 KiB of dynamic LDS at 1024 queries); a longer list: the cell is scored dense (every query)
#define CELL_SELF_GROUP 8      // steps of 64 queries whose table entries are in flight together (16 registers)
struct CellSelf {
    const float *lo_t;   // device [2^sa][m_padded]; null: the lists were made by knn_cells_match_kernel
    const float *hi;     // device [high entries][m_padded]
    const float *dup;    // device [m_padded]
    int sa, m_padded;
    const float *frame;  // per-cell frames (the centred scan): device [cells][KNN_CELL_FRAME_WORDS]; null otherwise
};

// -> the list's length; entries beyond CELL_SELF_CAP are counted, not stored.  `dupv`: the batch's Dup values (LDS in the
// scan, global memory in the tail kernel).  The caller orders the LDS writes against its reads (wave_lds_sync in knn_cells.hip).
__device__ __forceinline__ unsigned cell_self_list(const float *__restrict__ lo_t, const float *__restrict__ hi_t, int sa, int m_padded,
                                                   unsigned cell, int m, const float *__restrict__ dupv,
                                                   unsigned short *__restrict__ my_list, int lane)
{
    const float *__restrict__ lrow = lo_t + (size_t)(cell & ((1u << sa) - 1u)) * m_padded;
    const float *__restrict__ hrow = hi_t + (size_t)(cell >> sa) * m_padded;
    unsigned nq = 0u;
    for (int q0 = 0; q0 < m; q0 += 64 * CELL_SELF_GROUP) {   // (wave-uniform)
        float lo[CELL_SELF_GROUP], hi[CELL_SELF_GROUP];
#pragma unroll
        for (int u = 0; u < CELL_SELF_GROUP; ++u) {
            const int q = q0 + 64 * u + lane;
            lo[u] = q < m ? lrow[q] : 0.0f;
            hi[u] = q < m ? hrow[q] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < CELL_SELF_GROUP; ++u) {
            const int q = q0 + 64 * u + lane;
            if (q0 + 64 * u < m) {   // (wave-uniform)
                const float lb = lo[u] + hi[u];
                const bool pass = q < m && !(lb > dupv[q < m ? q : 0]);   // (a NaN keeps the cell, as in the match kernel)
                const u64 mask = __ballot(pass);
                if (mask != 0ull) {   // (wave-uniform)
                    const unsigned pos = nq + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    if (pass && pos < CELL_SELF_CAP)
                        my_list[pos] = (unsigned short)q;
                    nq += (unsigned)__popcll(mask);
                }
            }
        }
    }
    return nq;
}

// self.lo_t != null: the batch's lists are not in memory (the scan made its own) — this wave makes the list of every item it
// takes the same way, in `my_list` (CELL_SELF_CAP entries of LDS per wave); cell_counts / lists are then unused.
// KT (with K = 0): run-time k <= 16 KT
template <int K, int KT = 1>
__device__ __forceinline__ void cells_exact_items(
    const float *__restrict__ Q, const float *__restrict__ R, int krt, int m, long long base,
    const u64 *__restrict__ items, unsigned nitems, const unsigned *__restrict__ cell_counts,
    const unsigned short *__restrict__ lists, unsigned cap, const unsigned *__restrict__ perm,
    u64 *__restrict__ keys, unsigned wave, unsigned nwaves, const CellSelf &self, unsigned short *__restrict__ my_list)
{
#pragma clang fp contract(off)
    constexpr int KD = K > 0 ? K : 16 * KT;
    const int k = K > 0 ? K : krt;
    const int lane = threadIdx.x & (KNN_WAVE - 1);
    for (unsigned it = wave; it < nitems; it += nwaves) {
        const u64 item = items[it];
        const unsigned cell = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(item >> 48));
        const unsigned tb = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(item & 0xFFFFFFFFull));
        const unsigned nt = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(item >> 40) & 0xFFu));
        unsigned nq;
        const unsigned short *__restrict__ list;
        if (self.lo_t) {   // (kernel-uniform)
            __builtin_amdgcn_wave_barrier();   // the previous item's reads of the list are done
            nq = cell_self_list(self.lo_t, self.hi, self.sa, self.m_padded, cell, m, self.dup, my_list, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            list = my_list;
        } else {
            nq = (unsigned)__builtin_amdgcn_readfirstlane((int)cell_counts[cell]);
            list = lists + (size_t)cell * cap;
        }
        if (nq == 0u)
            continue;
        const bool dense = nq > cap;   // the list was cut short: every query of the batch (as in the MFMA scan)
        if (dense)
            nq = (unsigned)m;
        const unsigned p_end = (tb + nt) * 32u;
        for (unsigned p0 = tb * 32u; p0 < p_end; p0 += 64u * CX_ROWS) {
            float rv[CX_ROWS][KD];
            unsigned row[CX_ROWS];
#pragma unroll
            for (int j = 0; j < CX_ROWS; ++j) {
                const unsigned pos = p0 + (unsigned)j * 64u + (unsigned)lane;
                row[j] = pos < p_end ? perm[pos] : 0xFFFFFFFFu;   // ~0u: padding position
                const float *__restrict__ x = R + (size_t)(row[j] != 0xFFFFFFFFu ? row[j] : 0u) * k;
#pragma unroll
                for (int d = 0; d < KD; ++d)
                    rv[j][d] = d < k ? x[d] : 0.0f;
            }
            // the list 64 entries at a time on the lanes; the NEXT query's row and current key are requested before this
            // one's distances are computed (three dependent round trips per query — list entry, row, key — were 0.24 ms
            // for 64 clusters of 2^16 rows; the key may be stale: keys[] only ever decreases, a stale read costs a spare
            // reduction)
            for (unsigned e0 = 0u; e0 < nq; e0 += 64u) {
                const unsigned idx = min(e0 + (unsigned)lane, nq - 1u);
                const unsigned lq = dense ? idx : (unsigned)list[idx];
                const int cq = (int)min(64u, nq - e0);
                unsigned qid = (unsigned)__builtin_amdgcn_readlane((int)lq, 0);
                float qv[KD];
#pragma unroll
                for (int d = 0; d < KD; ++d)
                    qv[d] = d < k ? Q[(size_t)qid * k + d] : 0.0f;
                u64 cur = keys[qid];
                for (int j = 0; j < cq; ++j) {
                    const unsigned qid_n = (unsigned)__builtin_amdgcn_readlane((int)lq, min(j + 1, cq - 1));
                    float qn[KD];
#pragma unroll
                    for (int d = 0; d < KD; ++d)
                        qn[d] = d < k ? Q[(size_t)qid_n * k + d] : 0.0f;
                    const u64 cur_n = keys[qid_n];
                    u64 best = ~0ull;
#pragma unroll
                    for (int jr = 0; jr < CX_ROWS; ++jr) {
                        float acc = 0.0f;
#pragma unroll
                        for (int d = 0; d < KD; ++d)
                            if (d < k) {   // v0's order and operations: diff, square, add (no contraction)
                                const float diff = qv[d] - rv[jr][d];
                                const float sq = diff * diff;
                                acc = acc + sq;
                            }
                        if (row[jr] != 0xFFFFFFFFu && acc < INFINITY) {
                            const u64 key = pack_key(acc, (unsigned)(base + (long long)row[jr]));
                            best = key < best ? key : best;
                        }
                    }
                    if (__ballot(best < cur) != 0ull) {   // wave-uniform
                        const u64 wmin = wave_min_u64(best);
                        if (lane == 0)
                            key_atomic_min(&keys[qid], wmin);
                    }
#pragma unroll
                    for (int d = 0; d < KD; ++d)
                        qv[d] = qn[d];
                    cur = cur_n;
                    qid = qid_n;
                }
            }
        }
    }
}
