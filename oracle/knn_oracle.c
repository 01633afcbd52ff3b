/*
 * knn_oracle.c — CPU oracle (test infrastructure; see knn_oracle.h).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp (oracle/Makefile).
 * -ffp-contract=off is load-bearing: the reference binary's v0 loop is
 * strictly ordered scalar subss/mulss/addss, i.e. one
 * IEEE-754 binary32 rounding per operation and no fused multiply-add.
 */
#include "knn_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

float knn_oracle_dist2(int k, const float *q, const float *r)
{
    /* core.cu:44-49: squareSum = 0; for kInd: diff = q - r; squareSum += diff*diff */
    float acc = 0.0f;
    for (int d = 0; d < k; ++d) {
        const float diff = q[d] - r[d];
        const float sq = diff * diff;
        acc = acc + sq;
    }
    return acc;
}

static inline uint32_t f32_bits(float x)
{
    uint32_t u;
    memcpy(&u, &x, sizeof u);
    return u;
}

/* One query against refs [0,n): returns index, writes best distance. */
static int scan_one(int k, long long n, const float *q, const float *R, float *best_out)
{
    /* core.cu:39-40: minSquareSum = INFINITY, minIndex = 0 */
    float best = INFINITY;
    long long best_i = 0;
    for (long long i = 0; i < n; ++i) {
        const float ss = knn_oracle_dist2(k, q, R + (size_t)i * (size_t)k);
        /* core.cu:50-54: strict '>' — first minimum wins, NaN never wins */
        if (best > ss) {
            best = ss;
            best_i = i;
        }
    }
    *best_out = best;
    return (int)best_i;
}

void knn_oracle_v0(int k, int m, long long n, const float *Q, const float *R, int *out)
{
    for (int j = 0; j < m; ++j) {
        float best;
        out[j] = scan_one(k, n, Q + (size_t)j * (size_t)k, R, &best);
    }
}

int knn_oracle_v0_range(int k, int m0, int m1, long long n, const float *Q, const float *R,
                        int *out, int threads)
{
    int used = 1;
#ifdef _OPENMP
    if (threads <= 0)
        threads = omp_get_max_threads();
    used = threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int j = m0; j < m1; ++j) {
        float best;
        out[j] = scan_one(k, n, Q + (size_t)j * (size_t)k, R, &best);
    }
    return used;
}

void knn_oracle_v0_keys(int k, int m, long long n_local, const float *Q, const float *R_shard,
                        long long base, uint64_t *keys, int threads)
{
#ifdef _OPENMP
    if (threads <= 0)
        threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int j = 0; j < m; ++j) {
        float best;
        const int li = scan_one(k, n_local, Q + (size_t)j * (size_t)k, R_shard, &best);
        if (best == INFINITY) /* nothing beat +INF: global answer stays index 0 */
            keys[j] = (uint64_t)0x7F800000u << 32;
        else
            keys[j] = ((uint64_t)f32_bits(best) << 32) | (uint64_t)(uint32_t)(base + li);
    }
}

/* ---------------------------------------------------------------------- */
/* glibc TYPE_3 rand(): degree 31, separation 3 additive feedback generator.
 * State seeding: r[0] = seed (0 -> 1); r[i] = 16807 * r[i-1] mod (2^31 - 1)
 * computed with Schrage's split (q = 127773, rem = 2836), negative results
 * wrapped by +2^31-1; then 310 outputs are discarded.  Each draw adds the
 * word three places back into the front word and returns it >> 1. */
static int32_t ta_state[31];
static int ta_f = 3, ta_r = 0;

static int ta_step(void)
{
    uint32_t v = (uint32_t)ta_state[ta_f] + (uint32_t)ta_state[ta_r];
    ta_state[ta_f] = (int32_t)v;
    if (++ta_f >= 31) ta_f = 0;
    if (++ta_r >= 31) ta_r = 0;
    return (int)(v >> 1);
}

void ta_srand(unsigned int seed)
{
    if (seed == 0)
        seed = 1;
    int32_t w = (int32_t)seed;
    ta_state[0] = w;
    for (int i = 1; i < 31; ++i) {
        const int32_t hi = w / 127773;
        const int32_t lo = w % 127773;
        w = 16807 * lo - 2836 * hi;
        if (w < 0)
            w += 2147483647;
        ta_state[i] = w;
    }
    ta_f = 3;
    ta_r = 0;
    for (int i = 0; i < 310; ++i)
        (void)ta_step();
}

int ta_rand(void) { return ta_step(); }

void ta_get_sample(int k, int m, int n, float *Q, float *R)
{
    /* generator.h:14-19: getRandNum() = rand() / double(RAND_MAX), narrowed to float */
    const double rmax = 2147483647.0;
    for (long long i = 0; i < (long long)k * m; ++i)
        Q[i] = (float)(ta_rand() / rmax);
    for (long long i = 0; i < (long long)k * n; ++i)
        R[i] = (float)(ta_rand() / rmax);
}

/* ---------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void knn_synth_fill(float *x, long long count, uint64_t seed, long long first)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (long long i = 0; i < count; ++i) {
        const uint64_t ctr = (uint64_t)(first + i);
        const uint64_t z = mix64(seed + (ctr + 1) * 0x9E3779B97F4A7C15ull);
        x[i] = (float)(z >> 40) * 0x1.0p-24f;
    }
}
