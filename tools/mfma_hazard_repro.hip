// mfma_hazard_repro.hip — smallest form of the code shape that lost filter survivors:
// an MFMA whose result is first read in a LATER basic block.  Compile to ISA and audit:
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o repro.s mfma_hazard_repro.hip
//   python tools/mfma_hazard_audit.py repro.s
// With ROCm 7.2 (AMD clang 22) `with_branch` leaves 6 wait states between the second MFMA and the
// v_min3 that reads it when the branch is not taken (12 are required, and are inserted in
// `single_block`).  The production kernels keep MFMA and reader in one block.
#include <hip/hip_runtime.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float tree(const f16v &x, float th)
{
    float m = __builtin_fminf(__builtin_fminf(x[0], x[1]), x[2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2)
        m = __builtin_fminf(__builtin_fminf(m, x[i]), x[i + 1]);
    return __builtin_fminf(__builtin_fminf(m, x[15]), th);
}

__global__ void with_branch(const h8 *a, const h8 *b, const f16v *c, float th, float *out, unsigned *hits)
{
    const int lane = threadIdx.x;
    const h8 av = a[lane], b0 = b[lane], b1 = b[64 + lane];
    const f16v cv = c[lane];
    f16v d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b0, cv, 0, 0, 0);
    f16v d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b1, cv, 0, 0, 0);
    const float m0 = tree(d0, th);
    if (__builtin_expect(__ballot(m0 < th) != 0ull, 0))   // rare, wave-uniform: ends the basic block
        atomicAdd(hits, 1u);
    out[lane] = m0 + tree(d1, th);                          // d1 is read in the block after the branch
}

__global__ void single_block(const h8 *a, const h8 *b, const f16v *c, float th, float *out, unsigned long long *masks)
{
    const int lane = threadIdx.x;
    const h8 av = a[lane], b0 = b[lane], b1 = b[64 + lane];
    const f16v cv = c[lane];
    f16v d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b0, cv, 0, 0, 0);
    f16v d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b1, cv, 0, 0, 0);
    const float m0 = tree(d0, th);
    const unsigned long long k0 = __ballot(m0 < th);        // parked, looked at after the last reader
    const float m1 = tree(d1, th);
    out[lane] = m0 + m1;
    if (__builtin_expect((k0 | __ballot(m1 < th)) != 0ull, 0))
        masks[0] = k0;
}

// The production shape before the fix: a software-pipelined loop over query tiles (MFMA of step t+1
// issued before the reduction of step t) with a rare wave-uniform branch per step.
template <int QT>
__global__ __launch_bounds__(256, 2) void pipelined_with_branch(const h8 *a, const h8 *b, const f16v *c,
                                                                 const float *thr, unsigned long long *rec,
                                                                 unsigned *count, int tiles)
{
    const int lane = threadIdx.x & 63;
    h8 qf[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t)
        qf[t] = b[t * 64 + lane];
    unsigned cnt = 0;
    for (int tile = 0; tile < tiles; ++tile) {
        const h8 av = a[tile * 64 + lane];
        const f16v cv = c[tile * 64 + lane];
        f16v d[2];
        d[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, qf[0], cv, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const float th = thr[t * 32 + (lane & 31)];
            if (t + 1 < QT)
                d[(t + 1) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, qf[t + 1], cv, 0, 0, 0);
            const float mn = tree(d[t & 1], th);
            const unsigned long long mask = __ballot(mn < th);
            if (__builtin_expect(mask != 0ull, 0)) {
                if (mn < th)
                    rec[cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u))] =
                        ((unsigned long long)t << 32) | (unsigned)tile;
                cnt += (unsigned)__popcll(mask);
            }
        }
    }
    if (lane == 0)
        count[blockIdx.x] = cnt;
}
template __global__ void pipelined_with_branch<16>(const h8 *, const h8 *, const f16v *, const float *,
                                                   unsigned long long *, unsigned *, int);
template __global__ void pipelined_with_branch<32>(const h8 *, const h8 *, const f16v *, const float *,
                                                   unsigned long long *, unsigned *, int);
