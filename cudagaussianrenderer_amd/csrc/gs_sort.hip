// gs_sort.hip -- stable LSD radix sort of the (tile|depth key, splat index) pairs.
//
// Replaces the cub::DeviceRadixSort::SortPairs call of sortTileList (GaussianRender.cu:806-855) with
// rocPRIM's device radix sort on the same bit range [0, 32 + tile bits) and the same double-buffer
// protocol (the caller learns which buffer holds the result).  Pure HBM streaming: per pair and per
// digit pass 12 B read + 12 B written plus the 8 B histogram read.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "gs_internal.h"

namespace gs
{

int sortEndBit(int totalTiles)
{
    // GaussianRender.cu:818-819: 32 depth bits + the bits of the largest tile index (12 at 4096 tiles).
    int bits = 1;
    while ((1 << bits) < totalTiles) ++bits;
    return 32 + bits;
}

size_t sortTempBytes(int size, int endBit)
{
    rocprim::double_buffer<uint64_t> keys(nullptr, nullptr);
    rocprim::double_buffer<int32_t> values(nullptr, nullptr);
    size_t bytes = 0;
    GS_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys, values, (unsigned int) size, 0u, (unsigned int) endBit));
    return bytes;
}

int launchSortPairs(hipStream_t stream, void* temp, size_t tempBytes, uint64_t* keys0, uint64_t* keys1, int32_t* values0,
                    int32_t* values1, int selector, int size, int endBit)
{
    uint64_t* kc = selector ? keys1 : keys0;
    uint64_t* ka = selector ? keys0 : keys1;
    int32_t* vc = selector ? values1 : values0;
    int32_t* va = selector ? values0 : values1;
    rocprim::double_buffer<uint64_t> keys(kc, ka);
    rocprim::double_buffer<int32_t> values(vc, va);
    GS_CHECK(rocprim::radix_sort_pairs(temp, tempBytes, keys, values, (unsigned int) size, 0u, (unsigned int) endBit, stream));
    return (keys.current() == keys0) ? 0 : 1;
}

} // namespace gs
