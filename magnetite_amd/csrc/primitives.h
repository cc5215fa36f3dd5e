// Internal: device-wide sort/scan wrappers (primitives.hip).
#pragma once
#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

namespace magp {
hipError_t sort_pairs_u32(void *tmp, size_t *tmp_bytes, const uint32_t *kin, uint32_t *kout,
                          const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit, int end_bit,
                          hipStream_t s);
hipError_t sort_pairs_u64(void *tmp, size_t *tmp_bytes, const uint64_t *kin, uint64_t *kout,
                          const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit, int end_bit,
                          hipStream_t s);
hipError_t sort_keys_u64(void *tmp, size_t *tmp_bytes, const uint64_t *kin, uint64_t *kout, size_t n, int begin_bit,
                         int end_bit, hipStream_t s);
hipError_t exclusive_scan_i32(void *tmp, size_t *tmp_bytes, const int32_t *in, int32_t *out, size_t n,
                              hipStream_t s);
hipError_t exclusive_scan_i64(void *tmp, size_t *tmp_bytes, const int64_t *in, int64_t *out, size_t n,
                              hipStream_t s);
} // namespace magp
