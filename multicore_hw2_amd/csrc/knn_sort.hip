// knn_sort.hip — the one library call of the index build: a stable device radix sort of
// (norm bits, row number) pairs (rocPRIM).  Kept in its own translation unit: the template
// instantiation costs ~13 s of compile time.
#include <hip/hip_runtime.h>
#include <string.h>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "knn_common.h"

// keys: bit patterns of non-negative floats (+INF included) order like unsigned integers
hipError_t knn_sort_pairs_u32(void *tmp, size_t *tmp_bytes, const unsigned *keys_in, unsigned *keys_out,
                              const unsigned *vals_in, unsigned *vals_out, size_t count, hipStream_t stream)
{
    return rocprim::radix_sort_pairs(tmp, *tmp_bytes, keys_in, keys_out, vals_in, vals_out, count, 0, 32, stream);
}
