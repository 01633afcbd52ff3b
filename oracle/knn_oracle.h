/*
 * knn_oracle.h — CPU oracle for the brute-force 1-NN hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 * The shipped library (libknn_mi355x.so) never links, loads or calls it.
 *
 * Parity pin: the oracle + TA sample generator below reproduce all 8 index
 * lines of the reference's golden file /root/reference/results.csv (odd
 * lines 1..15) — checked by oracle/make_golden.py at fixture-generation time
 * and by tests/test_oracle.py against tests/golden/ta_indices.txt on every
 * run.  Shapes beyond the TA samples (n > 65536, multi-shard) have no golden
 * vector in the reference: there the oracle itself is the pin.
 */
#ifndef KNN_ORACLE_H
#define KNN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Restates v0::cudaCallback's loop nest (reference sources/src/core.cu:35-57):
 * for each query, scan refs in index order, squared L2 accumulated in fp32 in
 * dimension order (d = q - r; acc = acc + d*d, one rounding per operation, no
 * FMA), strict `best > ss` update so the first minimum wins; best starts at
 * +INF with index 0.  `out` is caller-allocated int[m]. */
void knn_oracle_v0(int k, int m, long long n, const float *Q, const float *R, int *out);

/* Same arithmetic, queries [m0, m1) only, spread over OpenMP threads (each
 * query's scan is untouched, so results are bit-identical to knn_oracle_v0).
 * threads <= 0 means omp_get_max_threads().  Returns the thread count used. */
int knn_oracle_v0_range(int k, int m0, int m1, long long n, const float *Q, const float *R,
                        int *out, int threads);

/* Shard form used to check the multi-GPU scheme (the partition of
 * core.cu:875-883 done right): scans refs [0, n_local) of a shard whose first
 * point has global index `base`, and returns for each query the packed key
 * (float_bits(best) << 32) | global_index, or 0x7F800000_00000000 when
 * nothing beats +INF.  Unsigned min over shards of these keys equals
 * knn_oracle_v0 on the concatenated set. */
void knn_oracle_v0_keys(int k, int m, long long n_local, const float *Q, const float *R_shard,
                        long long base, uint64_t *keys, int threads);

/* The fp32 squared distance exactly as v0 computes it (core.cu:44-49). */
float knn_oracle_dist2(int k, const float *q, const float *r);

/* ---- TA sample generator (reference sources/src/generator.h:14-50) ----
 * glibc's TYPE_3 additive-feedback rand(), restated from its published
 * algorithm (r[i] = r[i-3] + r[i-31], 310 warm-up draws, output >> 1) so the
 * GPU box needs no particular libc.  ta_rand_* is one global stream like the
 * reference's srand()/rand() pair. */
void ta_srand(unsigned int seed);
int ta_rand(void);
/* getSample (generator.h:32-50): fills Q[k*m] then R[k*n] from the stream,
 * each value (float)(rand() / (double)RAND_MAX). */
void ta_get_sample(int k, int m, int n, float *Q, float *R);

/* ---- Large-shape synthetic inputs ----
 * x[i] = (float)((splitmix64(seed ^ i) >> 40) * 2^-24), uniform in [0,1). */
void knn_synth_fill(float *x, long long count, uint64_t seed, long long first);

#ifdef __cplusplus
}
#endif
#endif
