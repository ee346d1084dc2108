// gs_ranges.hip -- per-tile [first, end) ranges of the sorted pair list.
//
// Replaces evaluateTileRangesKernel (GaussianRender.cu:857-888): tileRange[2t] = first index of tile
// t, tileRange[2t+1] = one past its last; tiles that own no pair keep the caller's 0xFF fill (-1,-1)
// and therefore count 0.  HBM streaming over the keys: each lane owns 16 bytes of keys (two 64-bit
// keys, four 32-bit or eight 16-bit tile indices) read with one load, plus one neighbour key.
#include "gs_internal.h"

namespace gs
{
namespace
{

template <typename Key>
__device__ __forceinline__ uint32_t tileOf(Key k)
{
    if constexpr (sizeof(Key) == 8) return (uint32_t) (k >> 32); // GaussianRender.cu:865
    else return (uint32_t) k;
}

template <typename Key>
__global__ __launch_bounds__(256) void tileRangesKernel(int sizeHost, const int* __restrict__ sizeDevice, const Key* __restrict__ keys,
                                                        int32_t* __restrict__ tileRange)
{
    const int size = sizeDevice ? min(*sizeDevice, sizeHost) : sizeHost;
    constexpr int PER = 16 / sizeof(Key);
    const int first = (blockIdx.x * blockDim.x + threadIdx.x) * PER; // this lane owns list entries [first, first+PER)
    if (first >= size) return;

    Key local[PER];
    if (first + PER <= size)
    {
        const uint4 raw = *reinterpret_cast<const uint4*>(keys + first);
        __builtin_memcpy(local, &raw, 16);
    }
    else
    {
#pragma unroll
        for (int i = 0; i != PER; ++i) local[i] = (first + i < size) ? keys[first + i] : Key(0);
    }
    uint32_t prev;
    if (first == 0)
    {
        prev = tileOf(local[0]);
        tileRange[prev * 2] = 0; // GaussianRender.cu:864-866
    }
    else prev = tileOf(keys[first - 1]);
#pragma unroll
    for (int i = 0; i != PER; ++i)
    {
        const int index = first + i;
        if (index < size)
        {
            const uint32_t tile = tileOf(local[i]);
            if (tile != prev) // GaussianRender.cu:882-886
            {
                tileRange[prev * 2 + 1] = index;
                tileRange[tile * 2] = index;
            }
            if (index == size - 1) tileRange[tile * 2 + 1] = size; // GaussianRender.cu:868-870
            prev = tile;
        }
    }
}

template <typename Key>
void launchT(hipStream_t stream, int size, const void* keys, int32_t* tileRange, const int* sizeDevice)
{
    constexpr int PER = 16 / sizeof(Key);
    const int lanes = (size + PER - 1) / PER;
    hipLaunchKernelGGL(tileRangesKernel<Key>, dim3((lanes + 255) / 256), dim3(256), 0, stream, size, sizeDevice, (const Key*) keys, tileRange);
}

// History of the range-table clear: in round 1 a hipMemsetAsync NODE of a captured band frame was seen to leave bytes that were not
// 0xFF, and the clear was moved into the clip kernel.  Round 2 looked for the cause: a standalone 28-node graph with a memcpy node,
// a memset node and by-value 96-byte kernel arguments (tools/micro/graph_memset.hip, 400 replays, both node orders) never shows it,
// and neither does the library itself built with the memset node back in place (-DGS_DEBUG_MEMSET_NODE: all captured / band / cull
// tests pass).  So it was not a ROCm defect; whatever wrote into the table then is gone.  The clear stays in the clip kernel
// because it is one graph node less, not as a workaround.

} // namespace

void launchTileRanges(hipStream_t stream, int size, int keyBytes, const void* keys, int32_t* tileRange, const int* sizeDevice)
{
    if (size <= 0) return; // Demo.cpp:469
    switch (keyBytes)
    {
    case 2: launchT<uint16_t>(stream, size, keys, tileRange, sizeDevice); break;
    case 4: launchT<uint32_t>(stream, size, keys, tileRange, sizeDevice); break;
    default: launchT<uint64_t>(stream, size, keys, tileRange, sizeDevice); break;
    }
}

} // namespace gs
