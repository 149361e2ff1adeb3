// sort.hip -- rocPRIM radix sort wrapper (see sort.h)
#include "sort.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace rt {

hipError_t sort_pairs_temp_bytes(size_t n, uint32_t key_bits, size_t *out_bytes) {
    *out_bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, *out_bytes, static_cast<const uint32_t *>(nullptr),
                                     static_cast<uint32_t *>(nullptr), static_cast<const uint32_t *>(nullptr),
                                     static_cast<uint32_t *>(nullptr), n, 0u, key_bits, hipStream_t(nullptr));
}

hipError_t sort_pairs(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                      uint32_t *vals_out, size_t n, uint32_t key_bits, hipStream_t stream) {
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, key_bits, stream);
}

hipError_t sort_pairs_u64(void *temp, size_t *temp_bytes, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in,
                          uint32_t *vals_out, size_t n, hipStream_t stream) {
    return rocprim::radix_sort_pairs(temp, *temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, 64u, stream);
}

hipError_t exclusive_scan_u32(void *temp, size_t *temp_bytes, const uint32_t *in, uint32_t *out, size_t n, hipStream_t stream) {
    return rocprim::exclusive_scan(temp, *temp_bytes, in, out, uint32_t(0), n, rocprim::plus<uint32_t>(), stream);
}

} // namespace rt
