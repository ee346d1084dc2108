// gs_tile_order.h -- the start order of a frame's raster tiles from the previous frame's consumption (gs_renderer_set_tile_start_order):
// the table builder, shared by its own small kernel (gs_raster.hip) and by the clip kernel, whose first eight workgroups run it where
// they can (gs_clip.hip) -- at the start of a frame, hidden behind 22 us of streaming.  See gs_raster.hip for what it is for.
#pragma once

#include "gs_internal.h"
#include "gs_wave_match.h"

#ifndef GS_RASTER_BLOCK_W
#define GS_RASTER_BLOCK_W 8
#define GS_RASTER_BLOCK_H 4
#endif

namespace gs
{

constexpr int kOrderBuckets = 64;
constexpr int kOrderWorkgroups = 8; // one per XCD

// Workgroup `xcd` of eight orders the tiles of the blocks XCD `xcd` renders (position p of XCD x is grid slot 8 p + x, the product's own
// map).  A frame's counters live in one of two buffers by frame parity: `now` holds the PREVIOUS frame's consumption (the frame that just
// ended), `prev` the one before -- which every workgroup clears for its own tiles once it has looked at them: the frame that is starting
// writes there, and a tile without work writes nothing.  The neighbourhood maximum reads other XCDs' tiles of `now`, which nobody
// modifies.  The workgroup that finishes last adds up the eight partial sums and decides whether the table is in force for this frame.
template <int THREADS>
__device__ __forceinline__ void tileStartOrderBody(int xcd, const Screen& scr, const TileOrderArgs& a)
{
    __shared__ int hist[kOrderBuckets];
    __shared__ int cursor[kOrderBuckets];
    __shared__ unsigned long long reduce[2][THREADS / kWave];
    __shared__ int count;
    // The builder's waves share their SIMDs with up to seven waves of the clip kernel each (40 VGPRs, VALU-heavy): at an eighth of the issue
    // slots its ~2 000 instructions per wave took 33 us -- 10 us longer than the kernel they ride in (rocprofv3: 22.8 -> 32.7 us).  Raised
    // priority gives them the issue slots they need for a few microseconds.
    __builtin_amdgcn_s_setprio(3);
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
    const int32_t* __restrict__ now = a.now;
    int32_t* __restrict__ prev = a.prev;
    int32_t* __restrict__ order = a.order;
    int32_t* __restrict__ state = a.state;
    const int firstTile = scr.rowBegin * scr.tilesX, rows = scr.rowEnd - scr.rowBegin, slots = a.slots;
    if (t < kOrderBuckets) hist[t] = 0;
    constexpr int BW = GS_RASTER_BLOCK_W;
    const int bh = rows < GS_RASTER_BLOCK_H ? rows : GS_RASTER_BLOCK_H;
    const int blocksX = (scr.tilesX + BW - 1) / BW;
    const int perBlock = BW * bh, perXcd = slots >> 3;
    // position p of this XCD -> the tile the product's map gives grid slot 8 p + xcd (gs_raster.hip rasterizeTilesKernel), -1 outside the band
    const auto tileAt = [&](int p, int& tx, int& ty) -> int {
        const int j = (p / perBlock) * 8 + xcd, inside = p % perBlock;
        tx = (j % blocksX) * BW + inside % BW;
        ty = (j / blocksX) * bh + inside / BW;
        return (tx < scr.tilesX && ty < rows) ? ty * scr.tilesX + tx : -1;
    };
    // Four tiles per lane and trip, their loads in flight together.  The key is the tile's own consumption in vote chunks of 32 (a first
    // version took the largest of the 3 x 3 neighbourhood -- a 1-degree step moves the image by about one tile -- but the ten loads per
    // tile made the builder the longest workgroup of the clip kernel it rides in, and the plain key predicts as well at 1-degree steps:
    // 48.1 against 46.5 us with the frame's own counts, tools/tile_order_experiment.py).
    constexpr int K = 4;
    const auto keysOf = [&](int base, int (&tile)[K], int (&bucket)[K]) {
        int key[K];
#pragma unroll
        for (int k = 0; k != K; ++k)
        {
            const int p = base + k * THREADS + t;
            int tx, ty;
            tile[k] = p < perXcd ? tileAt(p, tx, ty) : -1;
            key[k] = tile[k] >= 0 ? now[firstTile + tile[k]] : 0;
        }
#pragma unroll
        for (int k = 0; k != K; ++k) bucket[k] = kOrderBuckets - 1 - min(key[k] >> 5, kOrderBuckets - 1); // bucket 0 = the longest tiles
    };
    // The host saw the camera jump (gs_renderer.hip cameraJumped: the view's rotation changed by more than ~3 degrees since the previous
    // frame): the previous frame predicts nothing, and the builder's ten microseconds would be spent for a table nobody wants -- only the
    // counters this frame writes are cleared, and the table is withdrawn.
    if (a.skip != 0 || (a.skipFlagDevice != nullptr && *a.skipFlagDevice != 0.0f))
    {
        for (int p = t; p < perXcd; p += THREADS)
        {
            int tx, ty;
            const int i = tileAt(p, tx, ty);
            if (i >= 0) prev[firstTile + i] = 0;
        }
        if (xcd == 0 && t == 0)
        {
            state[0] = 0;
            state[1] = state[2] = 0;
            state[3] += 1;
        }
        return;
    }
    unsigned long long diff = 0, sum = 0;
    __syncthreads();
    // (a band of up to THREADS * K tiles per XCD -- 1920x1080 with 256 lanes: exactly -- is one trip, whose tiles and buckets stay in
    // registers for the second pass; larger screens make several trips and look their buckets up again)
    const bool oneTrip = perXcd <= THREADS * K;
    int keptTile[K], keptBucket[K];
    for (int base = 0; base < perXcd; base += THREADS * K)
    {
        int tile[K], bucket[K], c[K], b[K];
#pragma unroll
        for (int k = 0; k != K; ++k)
        {
            const int p = base + k * THREADS + t;
            int tx, ty;
            const int i = p < perXcd ? tileAt(p, tx, ty) : -1;
            c[k] = i >= 0 ? now[firstTile + i] : 0;
            b[k] = i >= 0 ? prev[firstTile + i] : 0;
        }
        keysOf(base, tile, bucket);
#pragma unroll
        for (int k = 0; k != K; ++k)
        {
            const bool valid = tile[k] >= 0;
            if (valid)
            {
                prev[firstTile + tile[k]] = 0; // (this frame's raster writes here)
                diff += (unsigned long long) (c[k] > b[k] ? c[k] - b[k] : b[k] - c[k]);
                sum += (unsigned long long) c[k];
            }
            // (most tiles share a handful of buckets: the lanes of a wave that hold the same bucket are found by ballots, the radix
            // scatter's primitive, and their first lane adds the group's size -- not a thousand LDS atomics on a few addresses)
            const unsigned long long same = matchDigit<6>(bucket[k], valid);
            if (valid && lanesBelow(same) == 0) atomicAdd(&hist[bucket[k]], __popcll(same));
            keptTile[k] = tile[k];
            keptBucket[k] = bucket[k];
        }
    }
#pragma unroll
    for (int off = 32; off != 0; off >>= 1)
    {
        diff += __shfl_xor(diff, off);
        sum += __shfl_xor(sum, off);
    }
    if (lane == 0)
    {
        reduce[0][wave] = diff;
        reduce[1][wave] = sum;
    }
    __syncthreads();
    if (wave == 0)
    {
        // cursor = exclusive prefix of the 64 bucket counts (one wave, six DPP adds), count = their total
        const int mine = hist[lane], incl = waveInclusiveSum(mine);
        cursor[lane] = incl - mine;
        if (lane == kWave - 1) count = incl;
    }
    if (t == 0)
    {
        // This XCD's part of the two sums and its arrival in ONE 64-bit word: sum in bits 0..27, difference in bits 28..56, arrivals in
        // bits 57..60 (per XCD the sum is capped at 2^25 - 1 and the difference at 2^26 - 1, so eight of them fit their fields: a 3840x2160
        // frame's XCD holds 4 050 tiles x at most 2 047 counted entries = 8.3e6).  One returning atomic
        // and no fence: the workgroup that finds seven arrivals before its own holds all the totals in the value it got back, decides,
        // and puts the word back to zero for the next frame (nobody else touches it before).  The builder rides in the clip kernel, whose
        // streaming loads the memory system: every dependent round trip saved here is several microseconds.
        unsigned long long dSum = 0, sSum = 0;
        for (int w = 0; w != THREADS / kWave; ++w)
        {
            dSum += reduce[0][w];
            sSum += reduce[1][w];
        }
        dSum = min(dSum, (1ull << 26) - 1ull);
        sSum = min(sSum, (1ull << 25) - 1ull);
        const unsigned long long mine = sSum | (dSum << 28) | (1ull << 57);
        const unsigned long long all = atomicAdd((unsigned long long*) (state + 4), mine) + mine;
        if ((all >> 57) == (unsigned long long) kOrderWorkgroups)
        {
            const unsigned long long sAll = all & ((1ull << 28) - 1ull), dAll = (all >> 28) & ((1ull << 29) - 1ull);
            *(volatile unsigned long long*) (state + 4) = 0ull;
            state[0] = (sAll > 0 && (float) dAll <= a.maxMismatch * (float) sAll) ? 1 : 0;
            state[1] = (int32_t) dAll;
            state[2] = (int32_t) sAll;
            state[3] += 1;
        }
    }
    __syncthreads();
    for (int base = 0; base < perXcd; base += THREADS * K)
    {
        int tile[K], bucket[K];
        if (oneTrip)
        {
#pragma unroll
            for (int k = 0; k != K; ++k)
            {
                tile[k] = keptTile[k];
                bucket[k] = keptBucket[k];
            }
        }
        else keysOf(base, tile, bucket);
#pragma unroll
        for (int k = 0; k != K; ++k)
        {
            const bool valid = tile[k] >= 0;
            const unsigned long long same = matchDigit<6>(bucket[k], valid);
            int first = 0;
            if (valid && lanesBelow(same) == 0) first = atomicAdd(&cursor[bucket[k]], __popcll(same)); // one returning atomic per bucket and wave
            first = __shfl(first, valid ? __ffsll((unsigned long long) same) - 1 : 0);
            if (valid) order[xcd + 8 * (first + lanesBelow(same))] = firstTile + tile[k];
        }
    }
    for (int p = count + t; p < perXcd; p += THREADS) order[xcd + 8 * p] = -1; // fewer tiles than slots: idle workgroups
}

} // namespace gs
