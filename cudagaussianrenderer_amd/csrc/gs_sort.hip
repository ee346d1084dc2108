// gs_sort.hip -- stable LSD radix sorts (rocPRIM device radix sort, double-buffer protocol).
//
//   * 64-bit keys: the (tile|depth key, splat index) pair sort of sortTileList (GaussianRender.cu:806-855),
//     same bit range [0, 32 + tile bits) and same "which buffer holds the result" contract as the
//     cub::DeviceRadixSort::SortPairs call it replaces;
//   * 32-bit keys: the renderer's depth sort of the splats (quantised depth -> splat index);
//   * 16/32-bit keys: the renderer's stable sort of the depth-ordered pairs by tile index.
// Pure HBM streaming: per element and digit pass (key+value) read + written, plus the histogram read.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "gs_internal.h"

namespace gs
{

int sortEndBit(int totalTiles)
{
    // GaussianRender.cu:818-819: 32 depth bits + the bits of the largest tile index (12 at 4096 tiles).
    return 32 + tileBits(totalTiles);
}

int tileBits(int totalTiles)
{
    int bits = 1;
    while ((1 << bits) < totalTiles) ++bits;
    return bits;
}

namespace
{
template <typename Key>
size_t tempBytesT(int size, int beginBit, int endBit)
{
    rocprim::double_buffer<Key> keys(nullptr, nullptr);
    rocprim::double_buffer<int32_t> values(nullptr, nullptr);
    size_t bytes = 0;
    GS_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys, values, (unsigned int) size, (unsigned int) beginBit, (unsigned int) endBit));
    return bytes;
}

template <typename Key>
int sortT(hipStream_t stream, void* temp, size_t tempBytes, void* keys0, void* keys1, int32_t* values0, int32_t* values1, int selector,
          int size, int beginBit, int endBit)
{
    Key* k0 = (Key*) keys0;
    Key* k1 = (Key*) keys1;
    rocprim::double_buffer<Key> keys(selector ? k1 : k0, selector ? k0 : k1);
    rocprim::double_buffer<int32_t> values(selector ? values1 : values0, selector ? values0 : values1);
    GS_CHECK(rocprim::radix_sort_pairs(temp, tempBytes, keys, values, (unsigned int) size, (unsigned int) beginBit, (unsigned int) endBit,
                                       stream));
    return (keys.current() == k0) ? 0 : 1;
}
} // namespace

size_t sortTempBytes(int keyBytes, int size, int beginBit, int endBit)
{
    switch (keyBytes)
    {
    case 2: return tempBytesT<uint16_t>(size, beginBit, endBit);
    case 4: return tempBytesT<uint32_t>(size, beginBit, endBit);
    default: return tempBytesT<uint64_t>(size, beginBit, endBit);
    }
}

int launchSortPairs(hipStream_t stream, void* temp, size_t tempBytes, int keyBytes, void* keys0, void* keys1, int32_t* values0,
                    int32_t* values1, int selector, int size, int beginBit, int endBit)
{
    switch (keyBytes)
    {
    case 2: return sortT<uint16_t>(stream, temp, tempBytes, keys0, keys1, values0, values1, selector, size, beginBit, endBit);
    case 4: return sortT<uint32_t>(stream, temp, tempBytes, keys0, keys1, values0, values1, selector, size, beginBit, endBit);
    default: return sortT<uint64_t>(stream, temp, tempBytes, keys0, keys1, values0, values1, selector, size, beginBit, endBit);
    }
}

} // namespace gs
