// sort.h -- device radix sort of (key, index) pairs, the glue behind ray sorting (K7).
// Implemented in sort.hip on top of rocPRIM (the vendor library; SURVEY.md kernel table K7 names it acceptable
// glue).  Kept in its own translation unit because the rocPRIM headers are slow to compile.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace rt {
// bytes of temporary storage rocPRIM needs for `n` pairs
hipError_t sort_pairs_temp_bytes(size_t n, uint32_t key_bits, size_t *out_bytes);
// sorts by the low `key_bits` bits of the keys; all pointers are device pointers; enqueued on `stream`
hipError_t sort_pairs(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                      uint32_t *vals_out, size_t n, uint32_t key_bits, hipStream_t stream);
// the same for 64-bit keys (all 64 bits), and an exclusive prefix sum of uint32 -- the two collective steps of the linear BVH
// builder (lbvh.hip.h).  With temp == nullptr both only report the temporary storage they need in *temp_bytes.
hipError_t sort_pairs_u64(void *temp, size_t *temp_bytes, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in,
                          uint32_t *vals_out, size_t n, hipStream_t stream);
hipError_t exclusive_scan_u32(void *temp, size_t *temp_bytes, const uint32_t *in, uint32_t *out, size_t n, hipStream_t stream);
} // namespace rt
