// Device-wide sort / scan primitives for the one-time symbolic phases
// (node ordering, incidence lists, CSR pattern).  rocPRIM does the radix
// sorts and scans; every hot-loop kernel is hand-written in solver.hip.
// Kept in its own translation unit so the heavy rocPRIM templates are
// compiled once and solver.hip rebuilds in seconds.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "primitives.h"

namespace magp {

// tmp == nullptr: only *tmp_bytes is written (size query), as in rocPRIM.

hipError_t sort_pairs_u32(void *tmp, size_t *tmp_bytes, const uint32_t *kin, uint32_t *kout,
                          const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit, int end_bit,
                          hipStream_t s)
{
    return rocprim::radix_sort_pairs(tmp, *tmp_bytes, kin, kout, vin, vout, n, (unsigned)begin_bit,
                                     (unsigned)end_bit, s);
}

hipError_t sort_pairs_u64(void *tmp, size_t *tmp_bytes, const uint64_t *kin, uint64_t *kout,
                          const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit, int end_bit,
                          hipStream_t s)
{
    return rocprim::radix_sort_pairs(tmp, *tmp_bytes, kin, kout, vin, vout, n, (unsigned)begin_bit,
                                     (unsigned)end_bit, s);
}

hipError_t sort_keys_u64(void *tmp, size_t *tmp_bytes, const uint64_t *kin, uint64_t *kout, size_t n, int begin_bit,
                         int end_bit, hipStream_t s)
{
    return rocprim::radix_sort_keys(tmp, *tmp_bytes, kin, kout, n, (unsigned)begin_bit, (unsigned)end_bit, s);
}

hipError_t exclusive_scan_i32(void *tmp, size_t *tmp_bytes, const int32_t *in, int32_t *out, size_t n,
                              hipStream_t s)
{
    return rocprim::exclusive_scan(tmp, *tmp_bytes, in, out, (int32_t)0, n, rocprim::plus<int32_t>(), s);
}

hipError_t exclusive_scan_i64(void *tmp, size_t *tmp_bytes, const int64_t *in, int64_t *out, size_t n,
                              hipStream_t s)
{
    return rocprim::exclusive_scan(tmp, *tmp_bytes, in, out, (int64_t)0, n, rocprim::plus<int64_t>(), s);
}

} // namespace magp
