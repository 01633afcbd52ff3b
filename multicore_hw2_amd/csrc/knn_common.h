// knn_common.h — shared declarations of the knn_mi355x library internals (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define KNN_WAVE 64

typedef unsigned long long u64;

static constexpr u64 kKeyInit = 0x7F80000000000000ull;  // (+INF, index 0)

static inline int knn_divup(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- exact VALU kernels (knn_exact.hip) -----------------------------------
// Folds the nearest reference of refs[0..n_local) (global index base + i) for each of the m
// queries into keys[] with a 64-bit unsigned atomic min.  Every distance is computed with the
// v0 arithmetic (reference core.cu:44-49).  Returns hipSuccess or the launch error.
hipError_t knn_exact_launch(int k, int m, long long n_local, long long base, const float *q_dev,
                            const float *r_dev, u64 *keys_dev, int num_cu, hipStream_t stream);

// Exact re-rank of an explicit candidate list: cand[c] = (query << 32) | local_ref_index.
hipError_t knn_rerank_launch(int k, const float *q_dev, const float *r_dev, long long base,
                             const u64 *cand_dev, const unsigned *count_dev, unsigned capacity,
                             u64 *keys_dev, hipStream_t stream);

hipError_t knn_keys_fill_launch(u64 *keys_dev, int m, hipStream_t stream);
hipError_t knn_keys_unpack_launch(const u64 *keys_dev, int m, int *out_dev, hipStream_t stream);
hipError_t knn_synth_fill_launch(float *dst, long long count, u64 seed, long long first,
                                 hipStream_t stream);
