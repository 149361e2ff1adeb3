// sort.hip -- rocPRIM radix sort wrapper (see sort.h)
#include "sort.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

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

} // namespace rt
