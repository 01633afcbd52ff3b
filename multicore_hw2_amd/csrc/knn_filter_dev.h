// knn_filter_dev.h — what knn_filter.hip and knn_cells.hip share (gfx950 only): fragment vector types, the
// monotone-atomic helpers, the error-bound constants and the per-query threshold, small host macros.
#pragma once

#include "knn_common.h"

#include <math.h>
#include <string.h>

// Device buffers of an index come from the library's pool (knn_api.cpp): a one-shot cudaCallback that
// builds the filter layouts makes ~20 allocations, and hipMalloc + hipFree (a device-wide sync and
// ~0.2 ms each) cost more than its kernels.  Stand-alone tools that include this file define KNN_NO_POOL.
#ifdef KNN_NO_POOL
#define KNN_DEV_ALLOC(p, bytes) hipMalloc(p, bytes)
#define KNN_DEV_FREE(p) hipFree(p)
#else
#define KNN_DEV_ALLOC(p, bytes) knn_dev_alloc((void **)(p), bytes)
#define KNN_DEV_FREE(p) knn_dev_free((void *)(p))
#endif

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define FILTER_BLOCK 256

// order-preserving map float -> uint (for atomic min/max over signed floats)
__device__ __forceinline__ unsigned f2ord(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}
static inline float ord2f_host(unsigned o)
{
    const unsigned u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// Monotone accumulators on a few hot words: a plain (possibly stale) read first.  The target only
// moves one way, so a stale value can cost a spare atomic, never lose an update; without the
// guard half a million atomics on one word serialise at ~88/us (6 ms on a 2^24-row shard).
__device__ __forceinline__ void guarded_atomic_max(unsigned *p, unsigned v)
{
    if (v > __builtin_nontemporal_load(p))
        atomicMax(p, v);
}
__device__ __forceinline__ void guarded_atomic_min(unsigned *p, unsigned v)
{
    if (v < __builtin_nontemporal_load(p))
        atomicMin(p, v);
}

__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v = fmaxf(v, __shfl_xor(v, off, KNN_WAVE));
    return v;
}


// ------------------------------------------------------------------------------------------
// Per-query pruning threshold (double arithmetic; see the header comment for the derivation).
// ------------------------------------------------------------------------------------------
struct BoundConsts {
    double eta, eta2, rho, g2, tau, gam, sigma2;
};

__host__ __device__ inline BoundConsts knn_bound_consts(int k, int kt, double sigma, double amax,
                                                        double bmax, double nmax)
{
    const double u = 0x1p-24;
    const double theta = 0x1p-11 + 0x1p-23;          // fp32 centring + fp16 rounding, relative
    const double thp = theta / (1.0 - theta);
    const double nu0 = 0x1p-14 * 1.001;               // fp16 subnormal rounding or flush-to-zero
    const double kp = 16.0 * kt;
    const double emax = thp * (amax + bmax) + 2.0 * nu0;
    BoundConsts c;
    c.eta2 = k * emax * emax;
    c.eta = sqrt(c.eta2);
    // MFMA internal accumulation of one score, relative to the sum of the magnitudes it adds (norm + k products), per chained
    // K-step.  The matrix core's adder tree is not documented, so kt 2^-18 is an allowance, not a derivation; it is pinned by
    // MEASURED WORST CASES on gfx950 (tests/test_parity_gpu.py::test_mfma_accumulation_error_on_adversarial_operands, round 5:
    // operands exactly representable in fp16, float64 reference exact): one product per K-step 2^10 times the others
    // 2^-22.6 (kt 1) .. 2^-20.0 (kt 256); fp16-subnormal operands 2^-21.9 (kt 1: they are NOT flushed); magnitudes 2^0..2^-12
    // mixed 2^-22.9 (kt 1) .. 2^-19.9 (kt 256); alternating +P, -P products: exact; gaussian data <= 2^-20 (kt <= 64).
    // The worst of all, 2^-19.9 at k = 4096, is 2^9.9 inside its allowance (2^-10); at kt = 1 the margin is 2^3.9.
    const double omega = kt * 0x1p-18;
    c.gam = (kp + 2.0) * u;
    const double mmax = kp * amax * amax;
    c.rho = (omega + 2.0 * c.gam) * 2.0 * (nmax + mmax) + kp * 0x1p-27;
    // the cell-pruned scan takes the C operand from norms kept as two fp16 halves (knn_cells.hip): off by at most
    // 2^-22 N, or 2^-25 where the low half is flushed; twice that is allowed for (always: it is a 2^-4 of omega's share)
    c.rho += 0x1p-21 * nmax + 0x1p-24;
    c.g2 = (k + 3.0) * u * 1.0001;
    c.tau = k * 0x1p-125;
    c.sigma2 = sigma * sigma;
    return c;
}

// Threshold implied by a filter score `u` = S of SOME real reference j0 of the shard (the minimum
// over the sample pass), for a query whose fp16 row has computed squared norm mq:
//   D~_j0 <= u + mq(1+g) + rho;  (sqrt(D_j0) - eta)^2 <= D~_j0 + 2 eta^2  =>  D_j0 <= D0up
//   the winner j* has E_j* <= E_j0 (v0 values), hence D_j* <= D0up (1+g2)^2 + sigma^2 tau =: Dup
//   and its own score obeys S_j* <= Dup + 2 eta sqrt(Dup) + eta^2 + rho - mq(1-g).
// Monotone in u, so any upper bound of the true sample minimum is safe too.
__host__ __device__ inline float knn_threshold(const BoundConsts &c, double u, double mq, double *dup_out = nullptr)
{
    double dt = u + mq * (1.0 + 1.01 * c.gam) + c.rho;
    if (dt < 0.0)
        dt = 0.0;
    const double sq0 = c.eta + sqrt(dt + 2.0 * c.eta2);
    const double dup = sq0 * sq0 * (1.0 + c.g2) * (1.0 + c.g2) + c.sigma2 * c.tau;
    if (dup_out)
        *dup_out = dup;  // real scaled squared distance no candidate for the answer can exceed (cell pruning)
    double thr = dup + 2.0 * c.eta * sqrt(dup) + c.eta2 + c.rho - mq * (1.0 - c.gam);
    thr += fabs(thr) * 1e-6 + 1e-30;                  // slack for the double arithmetic above
    float tf = (float)thr;
    if ((double)tf < thr)
        tf = nextafterf(tf, INFINITY);
    return nextafterf(tf, INFINITY);                  // the kernel tests S < thr (strict)
}


__device__ __forceinline__ float min3f(float a, float b, float c)
{
    return __builtin_fminf(__builtin_fminf(a, b), c);
}

#define FTRY(call)                       \
    do {                                 \
        hipError_t e_ = (call);          \
        if (e_ != hipSuccess)            \
            return e_;                   \
    } while (0)

static const float kAmaxLimit = 1024.0f;           // queries far outside the references' box
