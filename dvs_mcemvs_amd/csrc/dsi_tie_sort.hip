// dsi_tie_sort.hip -- the one library call of the exact tie resolver (dsi_mapper_resolve_near_ties): a device radix sort
// of (key, weight) pairs.  Kept in its own translation unit so that the rocPRIM templates stay out of dsi_kernels.hip.
// Off the throughput path: the resolver is an optional exactness pass over a few thousand near-tie columns.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "dsi_kernels.h"

namespace dsi {

hipError_t tie_sort_pairs(hipStream_t s, unsigned long long* keys_in, unsigned long long* keys_out, float* w_in, float* w_out,
                          size_t n, unsigned key_bits, void* tmp, size_t* tmp_bytes)
{
    if (key_bits > 64u) key_bits = 64u;
    return rocprim::radix_sort_pairs(tmp, *tmp_bytes, keys_in, keys_out, w_in, w_out, n, 0u, key_bits, s);
}

}  // namespace dsi
