// pm_sort.hip -- stable LSD radix sort of (63-bit Morton key, photon index) pairs on the key's top 30 bits (four
// 8-bit passes instead of eight): the octree is never walked below level 10 (pm_grid.h: cells of 1/1024 of the map's
// extent), so the order of photons inside a level-10 cell is free.
// Sorting is the one library primitive of the index build (rocPRIM, ROCm's own device library);
// the hot kernels (trace, shadow, gather) are hand written.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>

namespace pm {
hipError_t sort_pairs_u64(void *temp, size_t &temp_bytes, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in,
                          uint32_t *vals_out, size_t n, hipStream_t stream) {
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 33, 63, stream);
}
}  // namespace pm
