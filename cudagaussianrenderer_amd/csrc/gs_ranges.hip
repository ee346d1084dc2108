// gs_ranges.hip -- per-tile [first, end) ranges of the sorted pair list.
//
// Replaces evaluateTileRangesKernel (GaussianRender.cu:857-888): tileRange[2t] = first index of tile
// t, tileRange[2t+1] = one past its last; tiles that own no pair keep the caller's 0xFF fill (-1,-1)
// and therefore count 0.  HBM streaming over the keys (8 B per pair): each lane reads two adjacent
// keys with one 16-byte load and one extra neighbour key.
#include "gs_internal.h"

namespace gs
{
namespace
{

__global__ __launch_bounds__(256) void tileRangesKernel(int size, const uint64_t* __restrict__ keys, int32_t* __restrict__ tileRange)
{
    const int pairIndex = (blockIdx.x * blockDim.x + threadIdx.x) * 2; // this lane owns list entries pairIndex, pairIndex+1
    if (pairIndex >= size) return;

    const ulonglong2 two = (pairIndex + 1 < size) ? *reinterpret_cast<const ulonglong2*>(keys + pairIndex)
                                                  : make_ulonglong2(keys[pairIndex], 0ull);
    const uint32_t tile0 = (uint32_t) (two.x >> 32);
    if (pairIndex == 0)
    {
        tileRange[tile0 * 2] = 0; // GaussianRender.cu:864-866
    }
    else
    {
        const uint32_t prev = (uint32_t) (keys[pairIndex - 1] >> 32);
        if (prev != tile0) // GaussianRender.cu:882-886
        {
            tileRange[prev * 2 + 1] = pairIndex;
            tileRange[tile0 * 2] = pairIndex;
        }
    }
    if (pairIndex + 1 < size)
    {
        const uint32_t tile1 = (uint32_t) (two.y >> 32);
        if (tile1 != tile0)
        {
            tileRange[tile0 * 2 + 1] = pairIndex + 1;
            tileRange[tile1 * 2] = pairIndex + 1;
        }
        if (pairIndex + 2 == size) tileRange[tile1 * 2 + 1] = size; // GaussianRender.cu:868-870
    }
    else
    {
        tileRange[tile0 * 2 + 1] = size;
    }
}

} // namespace

void launchTileRanges(hipStream_t stream, int size, const uint64_t* keys, int32_t* tileRange)
{
    if (size <= 0) return; // Demo.cpp:469
    const int lanes = (size + 1) / 2;
    hipLaunchKernelGGL(tileRangesKernel, dim3((lanes + 255) / 256), dim3(256), 0, stream, size, keys, tileRange);
}

} // namespace gs
