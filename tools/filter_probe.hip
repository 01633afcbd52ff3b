// filter_probe.hip — experiment bench for the filter's hot loop: variants of the per-tile
// epilogue on the production data path (same fragment layout, loads and pipelining as
// knn_filter_kernel), timed on synthetic data.  Results are NOT checked: this tool only prices
// instruction mixes.  Build (in tools/): hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o filter_probe filter_probe.hip
#define KNN_NO_POOL
#include "../multicore_hw2_amd/csrc/knn_filter.hip"
#include "../multicore_hw2_amd/csrc/knn_exact.hip"
#include "../multicore_hw2_amd/csrc/knn_cells.hip"   // (knn_filter.hip links against the cell-pruned form since round 2)
#include <algorithm>

// VAR 0: production epilogue (8 min3 incl. thr, cmp, branch)
// VAR 1: running minimum only (8 min3, no cmp/branch)
// VAR 2: half tree (4 min3 + cmp + branch)                      [slope check, wrong results]
// VAR 3: production + s_setprio(1) around the MFMA issue
// VAR 4: xor/or accumulate, one cmp + branch per 4 tiles
// VAR 5: no epilogue at all (MFMA stream only, result kept alive per tile)
// VAR 6: MFMA accumulate chains (4 rotating accumulators, srcC = vdst), nothing else per tile
// VAR 7: VAR 6 + an 8-op min3 tree per tile on registers the MFMAs do not write (pure issue contention)
// VAR 8: VAR 6 with C taken from the fixed c tile (srcC != vdst), results folded once per ref tile
template <int VAR>
__global__ __launch_bounds__(FILTER_BLOCK, 2) void probe_kernel(const h8 *__restrict__ rf, const float *__restrict__ rn,
                                                                 const h8 *__restrict__ qfg, const float *__restrict__ thrg,
                                                                 long long ntiles, float *__restrict__ sink,
                                                                 unsigned long long *__restrict__ stamps)
{
    constexpr int KT = 1, QT = 32;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {   // where this wave runs: HW_REG_HW_ID (4) and HW_REG_XCC_ID (20)
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        stamps[4096 + blockIdx.x * 4 + (threadIdx.x >> 6)] = ((unsigned long long)xcc << 32) | hw;
        stamps[8192 + blockIdx.x * 4 + (threadIdx.x >> 6)] = r0;
    }
    __shared__ float s_thr[QT * 32];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < QT * 32; i += FILTER_BLOCK)
        s_thr[i] = thrg[i];
    __syncthreads();
    const long long wave = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long nwaves = (long long)gridDim.x * 4;
    const long long t0 = ntiles * wave / nwaves, t1 = ntiles * (wave + 1) / nwaves;
    h8 qf[QT][KT];
#pragma unroll
    for (int t = 0; t < QT; ++t)
        qf[t][0] = qfg[(size_t)t * 64 + lane];
    float um = 3e38f;
    unsigned hits = 0, acc = 0;
    h8 a[KT];
    f16v c;
    load_ref_tile<KT>(rf, rn, t0, lane, a, c);
    if (VAR == 11) {   // MFMAs issued in pairs: 2 in flight while the previous pair's trees run
        for (long long tile = t0; tile < t1; ++tile) {
            h8 an[KT];
            f16v cn;
            load_ref_tile<KT>(rf, rn, min(tile + 1, t1 - 1), lane, an, cn);
            f16v d[4];
            d[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], qf[0][0], c, 0, 0, 0);
            d[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], qf[1][0], c, 0, 0, 0);
            const float *vthr = s_thr + (lane & 31);
#pragma unroll
            for (int t = 0; t < QT; t += 2) {
                asm volatile("" ::: "memory");
                const float th0 = vthr[t * 32], th1 = vthr[(t + 1) * 32];
                const int cur = t & 2, nxt = cur ^ 2;
                if (t + 2 < QT) {
                    d[nxt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], qf[t + 2][0], c, 0, 0, 0);
                    d[nxt + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], qf[t + 3][0], c, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f16v &x = d[cur + u];
                    const float th = u ? th1 : th0;
                    const float m0 = min3f(x[0], x[1], x[2]);
                    const float m1 = min3f(x[3], x[4], x[5]);
                    const float m2 = min3f(x[6], x[7], x[8]);
                    const float m3 = min3f(x[9], x[10], x[11]);
                    const float m4 = min3f(x[12], x[13], x[14]);
                    const float m5 = min3f(m0, m1, m2);
                    const float m6 = min3f(m3, m4, x[15]);
                    const float mn = min3f(m5, m6, th);
                    if (__builtin_expect(__ballot(mn < th) != 0ull, 0)) { ++hits; um = mn; }
                }
            }
            a[0] = an[0];
            c = cn;
        }
        if (um == 1.2345f || hits == 0xFFFFFFFFu)
            sink[threadIdx.x] = um + hits;
        if (lane == 0) {
            stamps[2 * wave] = __builtin_amdgcn_s_memtime() - c0;
            stamps[2 * wave + 1] = __builtin_amdgcn_s_memrealtime() - r0;
        }
        return;
    }
    if (VAR == 9 || VAR == 10) {   // two reference tiles in flight
        h8 a1[KT], a2[KT];
        f16v c1, c2;
        load_ref_tile<KT>(rf, rn, min(t0 + 1, t1 - 1), lane, a1, c1);
        for (long long tile = t0; tile < t1; ++tile) {
            load_ref_tile<KT>(rf, rn, min(tile + 2, t1 - 1), lane, a2, c2);
            if (VAR == 9) {
                f16v e[4];
                e[0] = c; e[1] = c; e[2] = c; e[3] = c;
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    e[t & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], qf[t][0], e[t & 3], 0, 0, 0);
                um = fminf(um, e[0][0] + e[1][1] + e[2][2] + e[3][3]);
            } else {
                f16v d[2];
                d[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], qf[0][0], c, 0, 0, 0);
                const float *vthr = s_thr + (lane & 31);
                float th_next = vthr[0];
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    const float th = th_next;
                    asm volatile("" ::: "memory");
                    if (t + 1 < QT) {
                        th_next = vthr[(t + 1) * 32];
                        d[(t + 1) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], qf[t + 1][0], c, 0, 0, 0);
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
                    if (__builtin_expect(__ballot(mn < th) != 0ull, 0)) { ++hits; um = mn; }
                }
            }
            a[0] = a1[0]; c = c1;
            a1[0] = a2[0]; c1 = c2;
        }
        if (um == 1.2345f || hits == 0xFFFFFFFFu)
            sink[threadIdx.x] = um + hits;
        if (lane == 0) {
            stamps[2 * wave] = __builtin_amdgcn_s_memtime() - c0;
            stamps[2 * wave + 1] = __builtin_amdgcn_s_memrealtime() - r0;
        }
        return;
    }
    for (long long tile = t0; tile < t1; ++tile) {
        h8 an[KT];
        f16v cn;
        load_ref_tile<KT>(rf, rn, min(tile + 1, t1 - 1), lane, an, cn);
        if (VAR >= 6) {
            f16v e[4];
            e[0] = c; e[1] = c; e[2] = c; e[3] = c;
            f16v y = cn;
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                if (VAR == 8)
                    e[t & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], qf[t][0], c, 0, 0, 0);
                else
                    e[t & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], qf[t][0], e[t & 3], 0, 0, 0);
                if (VAR == 7) {
                    asm volatile("" : "+v"(y));
                    const float m0 = min3f(y[0], y[1], y[2]);
                    const float m1 = min3f(y[3], y[4], y[5]);
                    const float m2 = min3f(y[6], y[7], y[8]);
                    const float m3 = min3f(y[9], y[10], y[11]);
                    const float m4 = min3f(y[12], y[13], y[14]);
                    const float m5 = min3f(m0, m1, m2);
                    const float m6 = min3f(m3, m4, y[15]);
                    um = min3f(m5, m6, um);
                }
                if (VAR == 8 && (t & 3) == 3) {
                    asm volatile("" ::"v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]));
                }
            }
            um = fminf(um, e[0][0] + e[1][1] + e[2][2] + e[3][3]);
            a[0] = an[0];
            c = cn;
            continue;
        }
        f16v d[2];
        d[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], qf[0][0], c, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const float th = s_thr[t * 32 + (lane & 31)];
            if (t + 1 < QT) {
                if (VAR == 3) __builtin_amdgcn_s_setprio(1);
                d[(t + 1) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], qf[t + 1][0], c, 0, 0, 0);
                if (VAR == 3) __builtin_amdgcn_s_setprio(0);
            }
            const f16v &x = d[t & 1];
            if (VAR == 5) {
                asm volatile("" ::"v"(x));
                continue;
            }
            const float m0 = min3f(x[0], x[1], x[2]);
            const float m1 = min3f(x[3], x[4], x[5]);
            if (VAR == 2) {
                const float m2 = min3f(x[6], x[7], x[8]);
                const float mn = min3f(min3f(m0, m1, m2), x[15], th);
                if (__builtin_expect(mn < th, 0)) { ++hits; um = mn; }
                continue;
            }
            const float m2 = min3f(x[6], x[7], x[8]);
            const float m3 = min3f(x[9], x[10], x[11]);
            const float m4 = min3f(x[12], x[13], x[14]);
            const float m5 = min3f(m0, m1, m2);
            const float m6 = min3f(m3, m4, x[15]);
            if (VAR == 1) {
                um = min3f(m5, m6, um);
            } else if (VAR == 4) {
                const float mn = min3f(m5, m6, th);
                acc |= __float_as_uint(mn) ^ __float_as_uint(th);
                if ((t & 3) == 3) {
                    if (__builtin_expect(__ballot(acc != 0u) != 0ull, 0)) { ++hits; um = mn; }
                    acc = 0u;
                }
            } else {
                const float mn = min3f(m5, m6, th);
                const bool hit = mn < th;
                if (__builtin_expect(__ballot(hit) != 0ull, 0)) { ++hits; um = mn; }
            }
        }
        a[0] = an[0];
        c = cn;
    }
    if (VAR == 1)
        sink[1024 + (size_t)wave * 64 + lane] = um;
    if (um == 1.2345f || hits == 0xFFFFFFFFu)
        sink[threadIdx.x] = um + hits;
    if (lane == 0) {   // diagnostic only: shader cycles and 100 MHz ticks this wave ran
        stamps[2 * wave] = __builtin_amdgcn_s_memtime() - c0;
        stamps[2 * wave + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}


// ---- round 2 variants: per-query-tile running minima (no compare, no threshold read in the hot loop),
// NACC accumulators with the MFMA pipeline rotated across reference tiles (no bubble at a tile
// boundary), NSET operand sets (reference tiles in flight), WPS waves per SIMD.
// The MFMA and the min3 tree are inline asm: issue order = program order, accumulators pinned to arch
// VGPRs (left to itself hipcc puts them in AGPRs under a 512-register budget and copies every result
// back with 16 v_accvgpr_read per tile pair), query fragments pinned to AGPRs when QAGPR.
// Hazard: the tree of step t reads the accumulator NACC-1 steps after its MFMA was issued
// (>= 8 (NACC-1) VALU + NACC-1 MFMA issue slots in between; 8-pass XDL write -> VALU read needs 12 wait
// states: NACC >= 3 has them by construction, NACC = 2 pads with s_nop).
//   THRREG: thresholds in registers (1 wave per SIMD has the room), else read from LDS at the check.
// The check (run[t] < thr[t], once per NSET reference tiles) only counts hits here.
#define MFMA_ASM_V(D, A, B, C) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(D) : "v"(A), "v"(B), "v"(C))
#define MFMA_ASM_A(D, A, B, C) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(D) : "v"(A), "a"(B), "v"(C))
// One step = the MFMA of step t + NACC - 1 and the tree of step t in ONE statement (hipcc pads every asm
// boundary with an s_nop: 4 issue cycles).  The five temporaries are "+v": live across the whole loop, so
// they own their registers (as plain outputs hipcc parks them in whatever is dead at that point — e.g.
// the C tile an MFMA issued one instruction earlier is still reading).
#define TREE8 "v_min3_f32 %1, %10, %11, %12\n\t" "v_min3_f32 %2, %13, %14, %15\n\t" "v_min3_f32 %3, %16, %17, %18\n\t" \
              "v_min3_f32 %4, %19, %20, %21\n\t" "v_min3_f32 %5, %22, %23, %24\n\t" "v_min3_f32 %1, %1, %2, %3\n\t"       \
              "v_min3_f32 %4, %4, %5, %25\n\t" "v_min3_f32 %6, %1, %4, %6"
#define TREE4 "v_min3_f32 %1, %10, %11, %12\n\t" "v_min3_f32 %2, %13, %14, %15\n\t" "v_min3_f32 %3, %16, %17, %18\n\t" \
              "v_min3_f32 %6, %1, %2, %3"
#if defined(__HIP_DEVICE_COMPILE__)
#define KEEP_V(X) asm volatile("" ::"v"(X))
#define KEEP_A(X) asm volatile("" ::"a"(X))
#else
#define KEEP_V(X) (void)(X)
#define KEEP_A(X) (void)(X)
#endif
// Round 3 (`filter_probe r2 35` ended in a GPU memory fault when run as a process of its own).
// Cause, read off the ISA (tools/mfma_hazard_audit.py flags it: 4 / 5 / 11 wait states): an asm MFMA's output that no C++
// statement reads is dead at ASMEND for the register allocator, while the matrix core writes it 8 passes LATER.  KEEP_V
// kept one of the three rotating accumulators alive per step; the other two were handed out again at once — v[16:17] as
// the loop bound's compare operand four instructions after the MFMA that was going to overwrite them, and v[0:1] as the
// ADDRESS of the kernel's final store, computed 5 - 11 wait states after the last MFMA had been issued: when the late
// write-back won the race the store went to whatever the accumulator held.  Whether it did depended on timing (cold
// instruction cache in a single-variant process), which is why the full sweep got through.  Fix: every accumulator stays
// live at every MFMA-only step (KEEP_ALL), and the loop is followed by a drain that holds them until the last MFMA has
// retired (DRAIN: 2 x s_nop 15 with all accumulators as operands).
#if defined(__HIP_DEVICE_COMPILE__)
#define KEEP_ALL_V(D) do { if constexpr (NACC == 2) asm volatile("" ::"v"(D[0]), "v"(D[1])); else if constexpr (NACC == 3) asm volatile("" ::"v"(D[0]), "v"(D[1]), "v"(D[2])); else asm volatile("" ::"v"(D[0]), "v"(D[1]), "v"(D[2]), "v"(D[3])); } while (0)
#define KEEP_ALL_A(D) do { if constexpr (NACC == 2) asm volatile("" ::"a"(D[0]), "a"(D[1])); else if constexpr (NACC == 3) asm volatile("" ::"a"(D[0]), "a"(D[1]), "a"(D[2])); else asm volatile("" ::"a"(D[0]), "a"(D[1]), "a"(D[2]), "a"(D[3])); } while (0)
#define DRAIN_V(D) do { if constexpr (NACC == 2) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(D[0]), "+v"(D[1])); else if constexpr (NACC == 3) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(D[0]), "+v"(D[1]), "+v"(D[2])); else asm volatile("s_nop 15\n\ts_nop 15" : "+v"(D[0]), "+v"(D[1]), "+v"(D[2]), "+v"(D[3])); } while (0)
#define DRAIN_A(D) do { if constexpr (NACC == 2) asm volatile("s_nop 15\n\ts_nop 15" : "+a"(D[0]), "+a"(D[1])); else if constexpr (NACC == 3) asm volatile("s_nop 15\n\ts_nop 15" : "+a"(D[0]), "+a"(D[1]), "+a"(D[2])); else asm volatile("s_nop 15\n\ts_nop 15" : "+a"(D[0]), "+a"(D[1]), "+a"(D[2]), "+a"(D[3])); } while (0)
#else
#define KEEP_ALL_V(D) (void)(D)
#define KEEP_ALL_A(D) (void)(D)
#define DRAIN_V(D) (void)(D)
#define DRAIN_A(D) (void)(D)
#endif
#define TREE0 ""
#define TREE6 "v_min3_f32 %1, %10, %11, %12\n\t" "v_min3_f32 %2, %13, %14, %15\n\t" "v_min3_f32 %3, %16, %17, %18\n\t" \
              "v_min3_f32 %4, %19, %20, %21\n\t" "v_min3_f32 %1, %1, %2, %3\n\t" "v_min3_f32 %6, %1, %4, %6"
#define TREE12 TREE8 "\n\t" "v_min3_f32 %1, %10, %11, %12\n\t" "v_min3_f32 %2, %13, %14, %15\n\t" "v_min3_f32 %3, %16, %17, %18\n\t" \
              "v_min3_f32 %6, %1, %2, %6"
#define TREE16 TREE8 "\n\t" TREE8
// 16 two-operand ops (VOP2) instead of 8 three-operand ones
#define TREEV2 "v_min_f32 %1, %10, %11\n\t" "v_min_f32 %2, %12, %13\n\t" "v_min_f32 %3, %14, %15\n\t" "v_min_f32 %4, %16, %17\n\t"   \
               "v_min_f32 %5, %18, %19\n\t" "v_min_f32 %1, %1, %20\n\t" "v_min_f32 %2, %2, %21\n\t" "v_min_f32 %3, %3, %22\n\t"     \
               "v_min_f32 %4, %4, %23\n\t" "v_min_f32 %5, %5, %24\n\t" "v_min_f32 %1, %1, %25\n\t" "v_min_f32 %2, %2, %3\n\t"       \
               "v_min_f32 %4, %4, %5\n\t" "v_min_f32 %1, %1, %2\n\t" "v_min_f32 %4, %4, %6\n\t" "v_min_f32 %6, %1, %4"
// MFMA forms (no tree): literal-zero C; C and D both in AGPRs; accumulate chain (srcC = vdst)
// (the accumulators of the MFMA-only forms are "+v" / "+a": in-out, so their registers stay theirs from one statement to the
// next.  As plain outputs they were dead between their last KEEP and their redefinition, and the allocator used v[16:17]
// there as the scratch of the loop-bound compare while the MFMA issued three statements earlier was still writing them.)
#define STEP_ASM_C0(DN, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "+v"(DN) : "v"(A), "a"(B))
#define STEP_ASM_CH(DN, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(DN) : "v"(A), "a"(B))
#define STEP_ASM_AA(DN, A, B, C) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "+a"(DN) : "v"(A), "v"(B), "a"(C))
// 8 ops, the last one (which needs the two before it) moved to the front: it folds the PREVIOUS step's partial minima
// (kept in %1 / %4 of the other temporary set, passed as %5 <- unused here) -- see STEP_ASM_D
#define STEP_ASM(BCLS, CCLS, DN, A, B, C, X, RUN, T, PAD, TREE)                                                        \
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %7, %8, %9\n\t" PAD TREE                                                 \
                 : "=&v"(DN), "+v"(T[0]), "+v"(T[1]), "+v"(T[2]), "+v"(T[3]), "+v"(T[4]), "+v"(RUN)                    \
                 : "v"(A), BCLS(B), CCLS(C), "v"(X[0]), "v"(X[1]), "v"(X[2]), "v"(X[3]), "v"(X[4]), "v"(X[5]),         \
                   "v"(X[6]), "v"(X[7]), "v"(X[8]), "v"(X[9]), "v"(X[10]), "v"(X[11]), "v"(X[12]), "v"(X[13]),         \
                   "v"(X[14]), "v"(X[15]))
// deferred fold: this step computes its two partial minima into T[0] / T[3]; the fold of the PREVIOUS step's
// partials (TP[0], TP[3], the other temporary set) into its running minimum RUNP sits in the middle, so no
// instruction needs a result produced one or two instructions earlier
#define STEP_ASM_D(BCLS, CCLS, DN, A, B, C, X, RUNP, T, TP)                                                            \
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %9, %10, %11\n\t"                                                        \
                 "v_min3_f32 %1, %12, %13, %14\n\t" "v_min3_f32 %2, %15, %16, %17\n\t" "v_min3_f32 %3, %18, %19, %20\n\t" \
                 "v_min3_f32 %4, %21, %22, %23\n\t" "v_min3_f32 %5, %24, %25, %26\n\t"                                   \
                 "v_min3_f32 %6, %7, %8, %6\n\t"                                                                       \
                 "v_min3_f32 %1, %1, %2, %3\n\t" "v_min3_f32 %4, %4, %5, %27"                                          \
                 : "=&v"(DN), "+v"(T[0]), "+v"(T[1]), "+v"(T[2]), "+v"(T[3]), "+v"(T[4]), "+v"(RUNP)                   \
                 : "v"(TP[0]), "v"(TP[3]), "v"(A), BCLS(B), CCLS(C), "v"(X[0]), "v"(X[1]), "v"(X[2]), "v"(X[3]),       \
                   "v"(X[4]), "v"(X[5]), "v"(X[6]), "v"(X[7]), "v"(X[8]), "v"(X[9]), "v"(X[10]), "v"(X[11]),           \
                   "v"(X[12]), "v"(X[13]), "v"(X[14]), "v"(X[15]))

template <int NACC, int WPS, int NSET, bool THRREG, bool QAGPR, int TREE = 8, bool CAGPR = false>
__global__ __launch_bounds__(FILTER_BLOCK, WPS) void probe_run_kernel(const h8 *__restrict__ rf, const float *__restrict__ rn,
                                                                       const h8 *__restrict__ qfg, const float *__restrict__ thrg,
                                                                       long long ntiles, float *__restrict__ sink,
                                                                       unsigned long long *__restrict__ stamps)
{
    constexpr int QT = 32;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __shared__ float s_thr[QT * 32];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < QT * 32; i += FILTER_BLOCK)
        s_thr[i] = thrg[i];
    __syncthreads();
    const long long wave = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long nwaves = (long long)gridDim.x * 4;
    const long long t0 = ntiles * wave / nwaves, t1 = ntiles * (wave + 1) / nwaves;
    h8 qf[QT];
    float run[QT], thv[THRREG ? QT : 1];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        qf[t] = qfg[(size_t)t * 64 + lane];
        run[t] = INFINITY;
        if (THRREG)
            thv[t] = s_thr[t * 32 + (lane & 31)];
    }
    unsigned hits = 0;
    f16v dummy;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        dummy[i] = 1.0f + (float)(lane + i);
    asm volatile("" : "+v"(dummy));
    float tmp[2][5] = {{INFINITY, INFINITY, INFINITY, INFINITY, INFINITY}, {INFINITY, INFINITY, INFINITY, INFINITY, INFINITY}};   // alternate: a statement never reads its predecessor's outputs (no boundary s_nop)
    h8 a[NSET][1];
    f16v c[NSET];
#pragma unroll
    for (int s = 0; s < NSET; ++s)
        load_ref_tile<1>(rf, rn, min(t0 + s, t1 - 1), lane, a[s], c[s]);
    f16v d[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
        d[j] = c[0];
#pragma unroll
    for (int j = 0; j + 1 < NACC; ++j) {
        if (TREE == 22)
            STEP_ASM_AA(d[j], a[0][0], qf[j], c[0]);
        else if (QAGPR)
            MFMA_ASM_A(d[j], a[0][0], qf[j], c[0]);
        else
            MFMA_ASM_V(d[j], a[0][0], qf[j], c[0]);
    }
    // (hipcc pads nothing in front of an asm statement it cannot see into: the moves that set up d[] / c[] just above may
    // end one instruction before the first MFMA of the loop reads them — tools/mfma_hazard_audit.py, operand rule)
    asm volatile("s_nop 1");
    for (long long tile = t0; tile < t1; tile += NSET) {
#pragma unroll
        for (int s = 0; s < NSET; ++s) {
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const int tt = t + NACC - 1;
                const int su = tt < QT ? s : (s + 1) % NSET;   // the first NACC-1 steps of the next tile start early
                f16v &dn = d[tt % NACC];
                const f16v &x = TREE == 9 ? dummy : d[t % NACC];
                if constexpr (TREE == 10) {   // deferred fold (the very first fold uses tmp = +INF partials: harmless)
                    STEP_ASM_D("a", "v", dn, a[su][0], qf[tt % QT], c[su], x, run[(t + QT - 1) % QT], tmp[t & 1], tmp[(t + 1) & 1]);
                } else if constexpr (TREE == 0) {
                    STEP_ASM("a", "v", dn, a[su][0], qf[tt % QT], c[su], x, run[t], tmp[t & 1], "", TREE0);
                } else if constexpr (TREE == 6) {
                    STEP_ASM("a", "v", dn, a[su][0], qf[tt % QT], c[su], x, run[t], tmp[t & 1], "", TREE6);
                } else if constexpr (TREE == 12) {
                    STEP_ASM("a", "v", dn, a[su][0], qf[tt % QT], c[su], x, run[t], tmp[t & 1], "", TREE12);
                } else if constexpr (TREE == 16) {
                    STEP_ASM("a", "v", dn, a[su][0], qf[tt % QT], c[su], x, run[t], tmp[t & 1], "", TREE16);
                } else if constexpr (TREE == 17) {
                    STEP_ASM("a", "v", dn, a[su][0], qf[tt % QT], c[su], x, run[t], tmp[t & 1], "", TREEV2);
                } else if constexpr (TREE == 20) {
                    // (an asm output counts as written at ASMEND: a result nobody reads is dead at once, its
                    // registers get reused — e.g. for an address — while the MFMA is still going to write them.
                    // The empty statement keeps the accumulator issued NACC-1 steps ago live until here.)
                    STEP_ASM_C0(dn, a[su][0], qf[tt % QT]);
                    KEEP_ALL_V(d);
                } else if constexpr (TREE == 21) {
                    STEP_ASM_CH(dn, a[su][0], qf[tt % QT]);
                    KEEP_ALL_V(d);
                } else if constexpr (TREE == 22) {
                    STEP_ASM_AA(dn, a[su][0], qf[tt % QT], c[su]);
                    KEEP_ALL_A(d);
                } else if constexpr (TREE == 23) {   // literal-zero C + the full tree
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %7, %8, 0\n\t" TREE8
                                 : "=&v"(dn), "+v"(tmp[t & 1][0]), "+v"(tmp[t & 1][1]), "+v"(tmp[t & 1][2]), "+v"(tmp[t & 1][3]),
                                   "+v"(tmp[t & 1][4]), "+v"(run[t])
                                 : "v"(a[su][0]), "a"(qf[tt % QT]), "v"(c[su]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]),
                                   "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]),
                                   "v"(x[13]), "v"(x[14]), "v"(x[15]));
                } else if constexpr (TREE == 4) {
                    STEP_ASM("a", "v", dn, a[su][0], qf[tt % QT], c[su], x, run[t], tmp[t & 1], "", TREE4);
                } else if constexpr (!QAGPR) {
                    if (NACC == 2)
                        STEP_ASM("v", "v", dn, a[su][0], qf[tt % QT], c[su], x, run[t], tmp[t & 1], "s_nop 1\n\t", TREE8);
                    else
                        STEP_ASM("v", "v", dn, a[su][0], qf[tt % QT], c[su], x, run[t], tmp[t & 1], "", TREE8);
                } else {
                    if (NACC == 2)
                        STEP_ASM("a", "v", dn, a[su][0], qf[tt % QT], c[su], x, run[t], tmp[t & 1], "s_nop 1\n\t", TREE8);
                    else
                        STEP_ASM("a", "v", dn, a[su][0], qf[tt % QT], c[su], x, run[t], tmp[t & 1], "", TREE8);
                }
                if (t == QT - NACC + 1)   // every MFMA reading set s has been issued: refill it
                    load_ref_tile<1>(rf, rn, min(tile + s + NSET, t1 - 1), lane, a[s], c[s]);
            }
        }
        if (TREE == 10) {
            run[QT - 1] = min3f(tmp[(QT - 1) & 1][0], tmp[(QT - 1) & 1][3], run[QT - 1]);
            tmp[(QT - 1) & 1][0] = INFINITY;
            tmp[(QT - 1) & 1][3] = INFINITY;
        }
        u64 any = 0ull;
#pragma unroll
        for (int t = 0; t < QT; ++t)
            any |= __ballot(run[t] < (THRREG ? thv[t] : s_thr[t * 32 + (lane & 31)]));
        if (__builtin_expect(any != 0ull, 0))
            ++hits;
    }
    if constexpr (TREE == 22)
        DRAIN_A(d);
    else
        DRAIN_V(d);
    float um = run[0];
#pragma unroll
    for (int t = 1; t < QT; ++t)
        um = fminf(um, run[t]);
    sink[1024 + (size_t)wave * 64 + lane] = um;   // checked on the host: global minimum equal across variants
    if (hits == 0xFFFFFFFFu)
        sink[threadIdx.x] = um + hits;
    if (lane == 0) {
        stamps[2 * wave] = __builtin_amdgcn_s_memtime() - c0;
        stamps[2 * wave + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static unsigned long long *g_stamps;
typedef void (*probe_fn)(const h8 *, const float *, const h8 *, const float *, long long, float *, unsigned long long *);
template <int VAR>
static int run_k(const char *name, probe_fn fn, unsigned blocks, const h8 *rf, const float *rn, const h8 *qf, const float *thr, long long ntiles, float *sink);
template <int VAR>
static int run(const char *name, const h8 *rf, const float *rn, const h8 *qf, const float *thr, long long ntiles, float *sink)
{
    return run_k<VAR>(name, probe_kernel<VAR>, 512u, rf, rn, qf, thr, ntiles, sink);
}
template <int VAR>
static int run_k(const char *name, probe_fn fn, unsigned blocks, const h8 *rf, const float *rn, const h8 *qf, const float *thr, long long ntiles, float *sink)
{
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHK(hipEventRecord(a));
        if (rep == 0) {
            CHK(hipMemset(g_stamps, 0, 3 * 4096 * 8));
            CHK(hipMemset(sink + 1024, 0x7f, 2048 * 64 * 4));   // 0x7f7f7f7f = 3.4e38
        }
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(FILTER_BLOCK), 0, 0, rf, rn, qf, thr, ntiles, sink, g_stamps);
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float ms;
        CHK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double tiles = (double)ntiles * 32;
    std::vector<unsigned long long> hs(3 * 4096);
    CHK(hipMemcpy(hs.data(), g_stamps, 3 * 4096 * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk;
    for (int w = 0; w < 2048; ++w)
        if (hs[2 * w + 1])
            clk.push_back((double)hs[2 * w] / (double)hs[2 * w + 1] * 0.1);   // GHz
    std::sort(clk.begin(), clk.end());
    const double ghz = clk.empty() ? 0.0 : clk[clk.size() / 2];
    {   // per-wave wall time (s_memrealtime, 10 ns ticks): how uneven are equal shares of tiles?
        std::vector<double> dur;
        for (int w = 0; w < 2048; ++w)
            if (hs[2 * w + 1])
                dur.push_back((double)hs[2 * w + 1] * 0.01);   // us
        std::sort(dur.begin(), dur.end());
        if (!dur.empty())
        if (VAR == 0) {   // by XCC, and by how many of the kernel's waves share the SIMD
            double sum[8] = {0}, cnt[8] = {0};
            std::vector<unsigned long long> key(2048);
            for (int w = 0; w < 2048; ++w) {
                const unsigned hw = (unsigned)hs[4096 + w], xcc = (unsigned)(hs[4096 + w] >> 32) & 7u;
                sum[xcc] += (double)hs[2 * w + 1] * 0.01;
                cnt[xcc] += 1;
                // se[15:13] sh[12] cu[11:8] simd[5:4]
                key[w] = ((unsigned long long)xcc << 16) | (hw & 0xFF30u);
            }
            printf("    by XCC (avg us / waves):");
            for (int x = 0; x < 8; ++x)
                printf(" %.0f/%.0f", cnt[x] ? sum[x] / cnt[x] : 0.0, cnt[x]);
            printf("\n");
            double s1 = 0, n1 = 0, s2 = 0, n2 = 0, s3 = 0, n3 = 0;
            for (int w = 0; w < 2048; ++w) {
                int same = 0;
                for (int v = 0; v < 2048; ++v)
                    same += key[v] == key[w];
                const double d = (double)hs[2 * w + 1] * 0.01;
                if (same == 1) { s1 += d; n1 += 1; } else if (same == 2) { s2 += d; n2 += 1; } else { s3 += d; n3 += 1; }
            }
            {   // deciles, and the two waves of each SIMD: first to finish vs second
                std::vector<double> dd;
                for (int w = 0; w < 2048; ++w)
                    dd.push_back((double)hs[2 * w + 1] * 0.01);
                std::sort(dd.begin(), dd.end());
                printf("    deciles us:");
                for (int q = 0; q <= 10; ++q)
                    printf(" %.0f", dd[std::min<size_t>(dd.size() - 1, dd.size() * q / 10)]);
                printf("\n");
                double fs = 0, ss = 0, np = 0;
                for (int w = 0; w < 2048; ++w)
                    for (int v = w + 1; v < 2048; ++v)
                        if (key[v] == key[w]) {
                            const double a = (double)hs[2 * w + 1] * 0.01, b = (double)hs[2 * v + 1] * 0.01;
                            fs += std::min(a, b);
                            ss += std::max(a, b);
                            np += 1;
                        }
                printf("    SIMD pairs: %.0f, first to finish avg %.0f us, second avg %.0f us\n", np, np ? fs / np : 0.0, np ? ss / np : 0.0);
            }
            unsigned long long rmin = ~0ull, rmax = 0;
            for (int w = 0; w < 2048; ++w) { rmin = std::min(rmin, hs[8192 + w]); rmax = std::max(rmax, hs[8192 + w]); }
            printf("    waves alone on their SIMD: %.0f (avg %.0f us), two per SIMD: %.0f (avg %.0f us), more: %.0f (avg %.0f us); start spread %.1f us\n",
                   n1, n1 ? s1 / n1 : 0.0, n2, n2 ? s2 / n2 : 0.0, n3, n3 ? s3 / n3 : 0.0, (double)(rmax - rmin) * 0.01);
        }
            printf("    per-wave us: min %.0f  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f  (%zu waves)\n", dur.front(),
                   dur[dur.size() / 10], dur[dur.size() / 2], dur[dur.size() * 9 / 10], dur.back(), dur.size());
    }
    float gmin = 3.4e38f;
    {
        std::vector<float> hm(2048 * 64);
        CHK(hipMemcpy(hm.data(), sink + 1024, hm.size() * 4, hipMemcpyDeviceToHost));
        for (float v : hm)
            gmin = std::min(gmin, v);
    }
    printf("%-44s %8.3f ms  %7.1f TFLOP/s  %6.1f ns = %5.1f cycles per tile per SIMD at the in-kernel clock %.2f GHz  (global min score %.6g)\n", name, best,
           tiles * 32768 / (best * 1e-3) / 1e12, best * 1e6 / (tiles / 1024.0), best * 1e6 / (tiles / 1024.0) * ghz, ghz, gmin);
    return 0;
}

int main(int argc, char **argv)
{
    const long long n = 1ll << 24, ntiles = n / 32;
    float *refs, *rn, *thr, *sink, *q;
    h8 *rf, *qf;
    CHK(hipMalloc(&refs, n * 16 * 4));
    CHK(hipMalloc(&q, 1024 * 16 * 4));
    CHK(hipMalloc(&rf, ntiles * 1024));
    CHK(hipMalloc(&qf, 32 * 1024));
    CHK(hipMalloc(&rn, n * 4));
    CHK(hipMalloc(&thr, 1024 * 4));
    CHK(hipMalloc(&sink, (1024 + 2048 * 64) * 4));
    CHK(hipMalloc(&g_stamps, 3 * 4096 * 8));
    CHK(hipMemset(g_stamps, 0, 3 * 4096 * 8));
    CHK(knn_synth_fill_launch(refs, n * 16, 1001, 0, 0));
    CHK(knn_synth_fill_launch(q, 1024 * 16, 1000, 0, 0));
    float *center;
    unsigned *out;
    CHK(hipMalloc(&center, 64));
    CHK(hipMalloc(&out, 64));
    CHK(hipMemset(out, 0, 64));
    std::vector<float> hc(16, 0.5f), hthr(1024, -5.0f);   // thresholds no score can reach: no hits
    CHK(hipMemcpy(center, hc.data(), 64, hipMemcpyHostToDevice));
    CHK(hipMemcpy(thr, hthr.data(), 4096, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(knn_frag16_kernel, dim3((unsigned)(n / 256)), dim3(256), 0, 0, (const f4v *)refs, n, n, center, 2.0f, rf, rn, out, (unsigned *)sink, 0u);
    hipLaunchKernelGGL(knn_frag_kernel, dim3(4), dim3(256), 0, 0, q, 1024ll, 1024ll, 16, 1, center, 2.0f, -2.0f, 0.0f, qf, sink + 0, out, 0, (unsigned *)nullptr, (float *)nullptr, (unsigned *)nullptr, 0u);
    CHK(hipDeviceSynchronize());
    setvbuf(stdout, nullptr, _IONBF, 0);
    const bool r2 = argc > 1 && !strcmp(argv[1], "r2");
    const int only = argc > 2 ? atoi(argv[2]) : -1;   // run a single variant
    for (int round = 0; r2 && round < 3; ++round) {
        if (only < 0 || only == 0) if (run<0>("0 production (8 min3 + cmp + branch)", rf, rn, qf, thr, ntiles, sink)) return 1;
        if (only < 0 || only == 1) if (run<1>("1 running min (8 min3)", rf, rn, qf, thr, ntiles, sink)) return 1;
        if (only < 0 || only == 5) if (run<5>("5 MFMA stream only", rf, rn, qf, thr, ntiles, sink)) return 1;
        if (only < 0 || only == 21) if (run_k<21>("21 run[t] NACC 3, 1 wave/SIMD, 3 sets", probe_run_kernel<3, 1, 3, true, true>, 256u, rf, rn, qf, thr, ntiles, sink)) return 1;
        if (only < 0 || only == 25) if (run_k<25>("25 as 21, no tree (MFMA only)", probe_run_kernel<3, 1, 3, true, true, 0>, 256u, rf, rn, qf, thr, ntiles, sink)) return 1;
        if (only < 0 || only == 26) if (run_k<26>("26 as 21, 4-op tree", probe_run_kernel<3, 1, 3, true, true, 4>, 256u, rf, rn, qf, thr, ntiles, sink)) return 1;
        if (only < 0 || only == 31) if (run_k<31>("31 as 21, 6-op tree", probe_run_kernel<3, 1, 3, true, true, 6>, 256u, rf, rn, qf, thr, ntiles, sink)) return 1;
        if (only < 0 || only == 32) if (run_k<32>("32 as 21, 12-op tree", probe_run_kernel<3, 1, 3, true, true, 12>, 256u, rf, rn, qf, thr, ntiles, sink)) return 1;
        if (only < 0 || only == 35) if (run_k<35>("35 MFMA only, srcC = literal 0", probe_run_kernel<3, 1, 3, true, true, 20>, 256u, rf, rn, qf, thr, ntiles, sink)) return 1;
        if (only < 0 || only == 36) if (run_k<36>("36 MFMA only, chain srcC = vdst", probe_run_kernel<3, 1, 3, true, true, 21>, 256u, rf, rn, qf, thr, ntiles, sink)) return 1;
        if (only < 0 || only == 37) if (run_k<37>("37 MFMA only, C and D in AGPRs, B VGPR", probe_run_kernel<3, 1, 3, true, false, 22>, 256u, rf, rn, qf, thr, ntiles, sink)) return 1;
        if (only < 0 || only == 38) if (run_k<38>("38 MFMA only (as 25), 2 waves/SIMD NACC 2", probe_run_kernel<2, 2, 2, false, true, 0>, 512u, rf, rn, qf, thr, ntiles, sink)) return 1;
        if (only < 0 || only == 39) if (run_k<39>("39 as 21, srcC = literal 0 + full tree", probe_run_kernel<3, 1, 3, true, true, 23>, 256u, rf, rn, qf, thr, ntiles, sink)) return 1;
    }
    for (int round = 0; !r2 && round < 2; ++round) {
        if (run<0>("0 production (8 min3 + cmp + branch)", rf, rn, qf, thr, ntiles, sink)) return 1;
        if (run<1>("1 running min (8 min3)", rf, rn, qf, thr, ntiles, sink)) return 1;
        if (run<2>("2 half tree (4 min3 + cmp + branch)", rf, rn, qf, thr, ntiles, sink)) return 1;
        if (run<3>("3 production + setprio around MFMA", rf, rn, qf, thr, ntiles, sink)) return 1;
        if (run<4>("4 xor/or accumulate, branch per 4 tiles", rf, rn, qf, thr, ntiles, sink)) return 1;
        if (run<5>("5 MFMA stream only", rf, rn, qf, thr, ntiles, sink)) return 1;
        if (run<6>("6 MFMA accumulate chains only", rf, rn, qf, thr, ntiles, sink)) return 1;
        if (run<7>("7 MFMA chains + independent 8 min3", rf, rn, qf, thr, ntiles, sink)) return 1;
        if (run<8>("8 MFMA srcC=c, folded per 4 tiles", rf, rn, qf, thr, ntiles, sink)) return 1;
        if (run<11>("11 production epilogue, MFMAs issued in pairs", rf, rn, qf, thr, ntiles, sink)) return 1;
        if (run<9>("9 MFMA chains, 2 ref tiles prefetched", rf, rn, qf, thr, ntiles, sink)) return 1;
        if (run<10>("10 production epilogue, 2 ref tiles prefetched", rf, rn, qf, thr, ntiles, sink)) return 1;
    }
    return 0;
}
