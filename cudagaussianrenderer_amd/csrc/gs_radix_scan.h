// gs_radix_scan.h -- the column scan of a radix pass as a device routine: one workgroup per digit.  Shared by the sort's own scan kernel
// (gs_radix.hip) and by the kernel that lets the clip stage share the launch of the depth sort's FIRST scan (gs_clip.hip).  Not installed.
#pragma once

#include "gs_internal.h"
#include "gs_wave_match.h"

namespace gs
{

// Exclusive scan of digit `digit`'s row of table[digit][tile] over the tiles (contiguous: 16 bytes per lane and round), row total to
// totals[digit].  256 lanes for rows of up to 4096 tiles (one to four rounds), 1024 lanes beyond: the routine is pure latency, and
// with 1024 lanes per digit its 256-512 workgroups took every wave slot of the chip, so that in the pipelined frame loop nothing of
// the other frames could run beside it (17 us per launch there against 5 us alone).
template <int kScanThreads>
__device__ __forceinline__ void radixColumnScanRow(int digit, int nHost, const int* __restrict__ nDevice, int tileElements, int stride,
                                                   const int32_t* __restrict__ table, int32_t* __restrict__ scanned, int32_t* __restrict__ totals)
{
    constexpr int kScanWaves = kScanThreads / kWave;
    __shared__ int waveSums[2][kScanWaves];
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
    const int n = nDevice ? min(*nDevice, nHost) : nHost;
    const int numTiles = (n + tileElements - 1) / tileElements;
    const int32_t* in = table + (size_t) digit * stride;
    int32_t* out = scanned + (size_t) digit * stride;
    int carry = 0, flip = 0;
    for (int base = 0; base < numTiles; base += kScanThreads * 4, flip ^= 1)
    {
        const int tile = base + t * 4; // stride is a multiple of 4: a quad never leaves the row
        int4 v = make_int4(0, 0, 0, 0);
        if (tile < numTiles) v = *reinterpret_cast<const int4*>(in + tile);
        if (tile + 1 >= numTiles) v.y = 0; // entries past the live tiles hold stale counts
        if (tile + 2 >= numTiles) v.z = 0;
        if (tile + 3 >= numTiles) v.w = 0;
        const int own = v.x + v.y + v.z + v.w;
        const int incl = waveInclusiveSum(own);
        if (lane == kWave - 1) waveSums[flip][wave] = incl;
        __syncthreads(); // the other half of waveSums is free for the next round: one barrier per round
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w != kScanWaves; ++w)
        {
            const int ws = waveSums[flip][w];
            before += (w < wave) ? ws : 0;
            all += ws;
        }
        const int p0 = carry + before + incl - own;
        if (tile < numTiles) *reinterpret_cast<int4*>(out + tile) = make_int4(p0, p0 + v.x, p0 + v.x + v.y, p0 + v.x + v.y + v.z);
        carry += all;
    }
    if (t == 0) totals[digit] = carry;
}

} // namespace gs
