// plx_sort.hip -- stable radix sort of (vertex id, entry index) pairs for the
// splat CSR.  A plain library sort (rocPRIM) on the build path; it is kept in
// its own translation unit because the rocPRIM headers dominate compile time.
#include <cstring>

#include "plx_internal.h"

#include <rocprim/rocprim.hpp>

namespace plx {

int sort_pairs_temp_bytes(int64_t n, int end_bit, size_t *bytes)
{
    size_t tb = 0;
    PLX_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                          (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, 0,
                                          (unsigned)end_bit, (hipStream_t)0));
    *bytes = tb;
    return PLX_OK;
}

int sort_pairs(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out,
               const uint32_t *vals_in, uint32_t *vals_out, int64_t n, int end_bit, hipStream_t stream)
{
    PLX_HIP_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0,
                                          (unsigned)end_bit, stream));
    return PLX_OK;
}

int sort_pairs64_temp_bytes(int64_t n, int end_bit, size_t *bytes)
{
    size_t tb = 0;
    PLX_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                          (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, 0,
                                          (unsigned)end_bit, (hipStream_t)0));
    *bytes = tb;
    return PLX_OK;
}

int sort_pairs64(void *temp, size_t temp_bytes, const uint64_t *keys_in, uint64_t *keys_out,
                 const uint32_t *vals_in, uint32_t *vals_out, int64_t n, int end_bit, hipStream_t stream)
{
    PLX_HIP_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0,
                                          (unsigned)end_bit, stream));
    return PLX_OK;
}

}  // namespace plx
