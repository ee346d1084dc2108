// gs_radix.hip -- stable LSD radix sort of (16/32-bit key, 32-bit value) pairs for the renderer's two sorts
// (splats by the 27-bit depth code; depth-ordered pairs by 13-15 tile bits).
//
// rocPRIM's device radix sort stays behind the drop-in sortTileList (64-bit keys, gs_sort.hip).  For the
// renderer's element counts (1 M splats, ~5 M pairs) it spends most of its time outside the streaming
// passes (merge-sort fallback at 1 M elements, two memsets + look-back state per onesweep pass), so the
// renderer uses this plain three-kernel pass instead; every kernel is barrier-free across workgroups (no
// look-back, no spinning), so there is nothing that can hang:
//   1. histogram   -- one workgroup per 1024/2048-element tile counts its digits (up to 9 bits: 512 bins, two per lane) in LDS
//                     -> table[digit][tile];
//   2. column scan -- one workgroup per digit: exclusive scan of that digit's column over the tiles + its total;
//   3. scatter     -- the tile is re-read (keys and values in registers), ranked stably (ballot match + per-wave
//                     running counters in LDS), reordered by digit in LDS, and written out as runs of consecutive
//                     addresses (base = exclusive scan of the 256 digit totals + the scanned table entry).
// HBM traffic per pass and element: key read twice, value read once, key + value written once.
#include <type_traits>

#include "gs_depth_key.h"
#include "gs_internal.h"
#include "gs_radix_scan.h"
#include "gs_wave_match.h"

namespace gs
{
namespace
{

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / kWave;
// Elements per lane: 8 (2048-element tiles) for long lists, 4 (1024-element tiles) for short ones so that a
// 1 M-element sort still puts several workgroups on every CU.
constexpr int kItemsLarge = 8;
constexpr int kItemsSmall = 4;
constexpr int kSmallLimit = 2 * 1024 * 1024;
constexpr int kDigits = 256;
constexpr int kMaxDigitBits = 9; // 512 bins: two per lane in the histogram and scatter kernels

// Tiles counted by one histogram workgroup: its four counts of a digit are one 16-byte store into the digit's row of
// table[digit][tile] (a single tile per workgroup would scatter 256 four-byte stores a row apart).
constexpr int kHistTiles = 4;

// KEYGEN (first pass of the depth sort of a renderer with Morton storage): the keys do not exist yet -- element i = v * N + c is the
// splat with CALLER index c in view v; its depth code is computed here from the caller-ordered copy of the positions
// (gs_depth_key.h: the very routine the clip kernel uses) and stored to `keysOut` for the scatter of this pass.  The sort's
// input is thereby in the caller's order without a single gathered or scattered access, and a stable sort leaves equal depth
// codes in input order: ties blend in the caller's index order however the splats are stored.
// A plain histogram workgroup (256 lanes) walks its kHistTiles tiles in turn.  Counting them side by side, one 256-lane quarter each,
// was measured in round 4: the kernel is at its floor either way (5.6 against 5.7 us at 1 M keys) and the 1024-lane workgroups cost
// the pipelined frame loop 3 % (profiles/r04_ab_experiments.txt) -- kept only for the key-generating variant, whose per-element
// arithmetic (a projection and an IEEE division) would otherwise run four tiles deep on a wave that has its SIMD to itself.
#ifndef GS_KEYGEN_PARALLEL_TILES
#define GS_KEYGEN_PARALLEL_TILES 1 // A/B builds: 0 = the key-generating histogram on 256-lane workgroups too
#endif
struct NoKeySource
{
};
template <bool KEYGEN>
constexpr int kHistThreadsFor = (KEYGEN && GS_KEYGEN_PARALLEL_TILES) ? kThreads * 4 : kThreads;
template <typename SOURCE>
constexpr bool kIsKeySource = !std::is_same<SOURCE, NoKeySource>::value;

template <typename K, int BITS, int ITEMS, typename SOURCE>
__global__ __launch_bounds__(kHistThreadsFor<kIsKeySource<SOURCE>>) void radixHistogramKernel(
    const K* __restrict__ keys, SOURCE source, K* __restrict__ keysOut, int nHost, const int* __restrict__ nDevice, int shift, int skewed,
    int32_t* __restrict__ table, int stride)
{
    constexpr bool KEYGEN = kIsKeySource<SOURCE>;
    constexpr int kHistThreads = kHistThreadsFor<KEYGEN>;
    constexpr bool kParallelTiles = KEYGEN && GS_KEYGEN_PARALLEL_TILES;
    constexpr int kTile = kThreads * ITEMS;
    constexpr int DIGITS = BITS > 8 ? 2 * kDigits : kDigits; // 9-bit digits: every lane owns two bins
    static_assert(kHistTiles == 4, "four tiles per workgroup: one 16-byte table store per digit");
    __shared__ int hist[kHistTiles][DIGITS];
    const int t = threadIdx.x & (kThreads - 1); // lane within the tile's quarter
    // The element count may live on the device (the renderer's band-culled splat list): nHost is then only the
    // upper bound the grid was sized for; tiles past the real count do nothing.
    const int n = nDevice ? min(*nDevice, nHost) : nHost;
    const int firstTile = blockIdx.x * kHistTiles;
    if (firstTile * kTile >= n) return;
#pragma unroll
    for (int d = threadIdx.x; d < kHistTiles * DIGITS; d += kHistThreads) (&hist[0][0])[d] = 0;
    __syncthreads();
    constexpr int kTilesPerQuarter = kParallelTiles ? 1 : kHistTiles;
    for (int turn = 0; turn != kTilesPerQuarter; ++turn)
    {
        const int j = kParallelTiles ? (int) (threadIdx.x / kThreads) : turn;
        const int base = (firstTile + j) * kTile;
        if (base >= n) break;
        int* h = hist[j];
        K key[ITEMS];
        // Counting does not care which lane sees which key: full tiles are read 16 bytes per lane (ITEMS consecutive keys
        // when they fill a multiple of 16 bytes), the ragged last tile element by element.
        constexpr bool kWide = (sizeof(K) * ITEMS) % 16 == 0;
        const bool wide = !KEYGEN && kWide && base + kTile <= n;
        if constexpr (kWide && !KEYGEN)
        {
            if (wide)
            {
                constexpr int kVecs = (int) (sizeof(K) * ITEMS / 16);
                const uint4* src = reinterpret_cast<const uint4*>(keys + base) + t * kVecs;
                uint4 v[kVecs];
#pragma unroll
                for (int q = 0; q != kVecs; ++q) v[q] = src[q];
                __builtin_memcpy(key, v, sizeof(key));
            }
        }
        if constexpr (KEYGEN)
        {
            float4 pos[ITEMS];
            int view[ITEMS];
            const auto viewOf = [&](int i) {
                int v = 0;
#pragma unroll
                for (int w = 1; w != kMaxViews; ++w) v += (i >= w * source.splatsPerView) ? 1 : 0;
                return v;
            };
#pragma unroll
            for (int r = 0; r != ITEMS; ++r) // all loads in flight before the first use
            {
                const int i = base + r * kThreads + t;
                view[r] = viewOf(i);
                pos[r] = (i < n) ? source.positionCaller[i - view[r] * source.splatsPerView] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            // the camera: a kernel argument, or the device-resident one of the tile's view (captured frames) through scalar loads --
            // only a tile that straddles two views of a chain reads it lane by lane
            const int viewFirst = viewOf(base), viewLast = viewOf(min(base + kTile, n) - 1);
            if (source.params == nullptr || viewFirst == viewLast)
            {
                const ClipCamera cam = source.params != nullptr ? source.params[__builtin_amdgcn_readfirstlane(viewFirst)].camera : source.camera;
#pragma unroll
                for (int r = 0; r != ITEMS; ++r) key[r] = (K) depthKeyOfSplat(pos[r], cam);
            }
            else
            {
#pragma unroll
                for (int r = 0; r != ITEMS; ++r) key[r] = (K) depthKeyOfSplat(pos[r], source.params[view[r]].camera);
            }
#pragma unroll
            for (int r = 0; r != ITEMS; ++r)
                if (base + r * kThreads + t < n) keysOut[base + r * kThreads + t] = key[r];
        }
        else if (!wide)
        {
#pragma unroll
            for (int r = 0; r != ITEMS; ++r) // all loads in flight before the first use
            {
                const int i = base + r * kThreads + t;
                key[r] = (i < n) ? keys[i] : K(0);
            }
        }
        if (skewed && wide)
        {
            // Top digit of a multi-pass tile sort: the lane's ITEMS consecutive keys mostly share it -- one add per run.
            int run = 1;
#pragma unroll
            for (int r = 0; r != ITEMS; ++r)
            {
                const int digit = (int) ((key[r] >> shift) & ((1 << BITS) - 1));
                const int next = (r + 1 != ITEMS) ? (int) ((key[(r + 1) % ITEMS] >> shift) & ((1 << BITS) - 1)) : -1;
                if (next == digit) ++run;
                else
                {
                    atomicAdd(&h[digit], run);
                    run = 1;
                }
            }
        }
        else
        {
#pragma unroll
            for (int r = 0; r != ITEMS; ++r)
            {
                const bool valid = wide || base + r * kThreads + t < n;
                const int digit = (int) ((key[r] >> shift) & ((1 << BITS) - 1));
                if (BITS == 8 || !skewed)
                {
                    // Well-spread digits: plain LDS atomics (lanes sharing a digit serialise inside the LDS).
                    if (valid) atomicAdd(&h[digit], 1);
                }
                else
                {
                    // skewed digit, ragged tile: one add per distinct digit per wave
                    const unsigned long long m = matchDigit<BITS>(digit, valid);
                    if (valid && lanesBelow(m) == 0) atomicAdd(&h[digit], __popcll(m));
                }
            }
        }
    }
    __syncthreads();
    // [digit][tile]: every digit's row is contiguous for the scan; stride is a multiple of 4, so the quad is aligned
#pragma unroll
    for (int d = threadIdx.x; d < DIGITS; d += kHistThreads)
        *reinterpret_cast<int4*>(table + (size_t) d * stride + firstTile) = make_int4(hist[0][d], hist[1][d], hist[2][d], hist[3][d]);
}

// One workgroup per digit: exclusive scan of that digit's row of table[digit][tile] over the tiles (gs_radix_scan.h).
template <int kScanThreads>
__global__ __launch_bounds__(kScanThreads) void radixColumnScanKernel(int nHost, const int* __restrict__ nDevice, int tileElements, int stride,
                                                                      const int32_t* __restrict__ table, int32_t* __restrict__ scanned,
                                                                      int32_t* __restrict__ totals)
{
    radixColumnScanRow<kScanThreads>((int) blockIdx.x, nHost, nDevice, tileElements, stride, table, scanned, totals);
}

// CARRY (the staged buildTileList, gs_staged.hip): a second 32-bit payload -- the pair's quantised depth -- travels with the value;
// the last pass writes it as the low word of the reference's 64-bit key (tile << 32 | depth, GaussianRender.cu:446-454) into keys64Out
// instead of writing the tile index.
template <typename K, int BITS, int ITEMS, bool CARRY = false>
__global__ __launch_bounds__(kThreads) void radixScatterKernel(const K* __restrict__ keysIn, const int32_t* __restrict__ valuesIn,
                                                                K* __restrict__ keysOut, int32_t* __restrict__ valuesOut, int nHost, const int* __restrict__ nDevice,
                                                                int shift, const int32_t* __restrict__ scanned,
                                                                const int32_t* __restrict__ totals, int stride, int32_t* __restrict__ rangeTable,
                                                                const uint32_t* __restrict__ carryIn, uint32_t* __restrict__ carryOut,
                                                                uint64_t* __restrict__ keys64Out)
{
    constexpr int kTile = kThreads * ITEMS;
    constexpr int kDigitMask = (1 << BITS) - 1;
    constexpr int DIGITS = BITS > 8 ? 2 * kDigits : kDigits; // 9-bit digits: 512 bins, lane t owns bins OWN*t .. OWN*t + OWN-1
    constexpr int OWN = DIGITS / kThreads;
    __shared__ int cnt[kWaves][DIGITS]; // per-wave running digit counters, then per-wave exclusive prefixes
    __shared__ int localBase[DIGITS];   // first slot of each digit inside the tile
    __shared__ int globalBase[DIGITS];  // first output index of each digit for this tile
    __shared__ int waveSums[2][kWaves];
    __shared__ K sKeys[kTile];
    __shared__ int32_t sVals[kTile];
    __shared__ uint32_t sCarry[CARRY ? kTile : 1];

    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
    const int n = nDevice ? min(*nDevice, nHost) : nHost;
    // Workgroups are dealt to the 8 XCDs round-robin; give every XCD a CONTIGUOUS eighth of the tiles, so that the
    // output runs of neighbouring tiles (adjacent addresses, often the two halves of one 64-byte sector) meet in the
    // same L2 instead of leaving two partial lines in two L2s.  Any bijection works: the tables are indexed by tile.
    const int numTiles = (n + kTile - 1) / kTile;
    const int perXcd = (numTiles + 7) >> 3;
    const int slot = (int) blockIdx.x >> 3;
    const int tile = ((int) blockIdx.x & 7) * perXcd + slot;
    const int base = tile * kTile;
    if (slot >= perXcd || base >= n) return; // whole workgroup (uniform): the grid was sized for the host-side upper bound
    const int tileCount = min(kTile, n - base);

    // ---- load the tile: element (wave, row, lane) is tile index wave*64*ITEMS + row*64 + lane ---------
    K key[ITEMS];
    int32_t val[ITEMS];
    uint32_t carried[CARRY ? ITEMS : 1];
#pragma unroll
    for (int r = 0; r != ITEMS; ++r)
    {
        const int i = base + wave * (kWave * ITEMS) + r * kWave + lane;
        const bool valid = i < n;
        key[r] = valid ? keysIn[i] : K(0);
        val[r] = valid ? valuesIn[i] : 0;
        if constexpr (CARRY) carried[r] = valid ? carryIn[i] : 0u;
    }
    int digitTotal[OWN], tileEntry[OWN];
#pragma unroll
    for (int o = 0; o != OWN; ++o)
    {
        digitTotal[o] = totals[OWN * t + o];
        tileEntry[o] = scanned[(size_t) (OWN * t + o) * stride + tile];
    }
#pragma unroll
    for (int w = 0; w != kWaves; ++w)
#pragma unroll
        for (int o = 0; o != OWN; ++o) cnt[w][OWN * t + o] = 0;
    __syncthreads();

    // ---- stable rank inside the wave: ballot match, then running per-digit counters ----------------------
    unsigned long long match[ITEMS];
#pragma unroll
    for (int r = 0; r != ITEMS; ++r) // independent of each other: issued back to back
    {
        const bool valid = base + wave * (kWave * ITEMS) + r * kWave + lane < n;
        match[r] = matchDigit<BITS>((int) ((key[r] >> shift) & kDigitMask), valid);
    }
    int rank[ITEMS];
#pragma unroll
    for (int r = 0; r != ITEMS; ++r)
    {
        const unsigned long long m = match[r];
        const int digit = (int) ((key[r] >> shift) & kDigitMask);
        const bool valid = m != 0ull;
        // every lane of a digit group reads the running count (one broadcast read), then the group's first lane -- the one with no
        // lane of the group below it -- bumps it: LDS instructions of a wave execute in order, so the next row sees the bumped value
        const int below = lanesBelow(m);
        const int old = valid ? cnt[wave][digit] : 0;
        if (valid && below == 0) cnt[wave][digit] = old + __popcll(m);
        rank[r] = old + below;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();

    // ---- per digit (lane t owns OWN consecutive digits): prefixes over waves, tile-local base, global base ----------
    {
        int sum[OWN], sumAll = 0, totalAll = 0;
#pragma unroll
        for (int o = 0; o != OWN; ++o)
        {
            const int d = OWN * t + o;
            int acc = 0;
#pragma unroll
            for (int w = 0; w != kWaves; ++w)
            {
                const int c = cnt[w][d];
                cnt[w][d] = acc;
                acc += c;
            }
            sum[o] = acc;
            sumAll += acc;
            totalAll += digitTotal[o];
        }
        const int inclLocal = waveInclusiveSum(sumAll), inclGlobal = waveInclusiveSum(totalAll);
        if (lane == kWave - 1)
        {
            waveSums[0][wave] = inclLocal;
            waveSums[1][wave] = inclGlobal;
        }
        __syncthreads();
        int beforeLocal = 0, beforeGlobal = 0;
#pragma unroll
        for (int w = 0; w != kWaves; ++w)
        {
            beforeLocal += (w < wave) ? waveSums[0][w] : 0;
            beforeGlobal += (w < wave) ? waveSums[1][w] : 0;
        }
        int local = beforeLocal + inclLocal - sumAll, global = beforeGlobal + inclGlobal - totalAll;
#pragma unroll
        for (int o = 0; o != OWN; ++o)
        {
            localBase[OWN * t + o] = local;
            globalBase[OWN * t + o] = global + tileEntry[o];
            local += sum[o];
            global += digitTotal[o];
        }
    }
    __syncthreads();

    // ---- reorder by digit in LDS ----------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r != ITEMS; ++r)
    {
        if (match[r] != 0ull)
        {
            const int digit = (int) ((key[r] >> shift) & kDigitMask);
            const int slot = localBase[digit] + cnt[wave][digit] + rank[r];
            sKeys[slot] = key[r];
            sVals[slot] = val[r];
            if constexpr (CARRY) sCarry[slot] = carried[r];
        }
    }
    __syncthreads();

    // ---- write runs ---------------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r != ITEMS; ++r)
    {
        const int j = r * kThreads + t;
        if (j < tileCount)
        {
            const K k = sKeys[j];
            const int digit = (int) ((k >> shift) & kDigitMask);
            const int pos = globalBase[digit] + (j - localBase[digit]);
            valuesOut[pos] = sVals[j];
            if constexpr (CARRY)
            {
                // (the staged buildTileList: the last pass leaves the reference's key, tile | quantised depth, instead of the tile index)
                if (keys64Out != nullptr) keys64Out[pos] = ((uint64_t) k << 32) | (uint64_t) sCarry[j];
                else
                {
                    keysOut[pos] = k;
                    carryOut[pos] = sCarry[j];
                }
            }
            else keysOut[pos] = k;
            if (rangeTable != nullptr)
            {
                // The last pass of the renderer's tile sort finds the tile ranges itself (GaussianRender.cu:857-888; no range kernel):
                // the workgroup's elements of one key sit together in LDS with ascending output positions, so the first of them
                // bids for the range's begin (unsigned min against the preset -1 = 0xFFFFFFFF) and the last for its end (signed max
                // against -1).  ~0.7 M fire-and-forget atomics per frame at C3; tiles without pairs keep (-1, -1) (Demo.cpp:403).
                if (j == 0 || sKeys[j - 1] != k) atomicMin((unsigned*) &rangeTable[2 * (int) k], (unsigned) pos);
                if (j == tileCount - 1 || sKeys[j + 1] != k) atomicMax(&rangeTable[2 * (int) k + 1], pos + 1);
            }
        }
    }
}

int itemsPerLane(int n) { return n <= kSmallLimit ? kItemsSmall : kItemsLarge; }

template <typename K, int BITS, int ITEMS>
void runPassItems(hipStream_t stream, const K* keysIn, const int32_t* valuesIn, const DepthKeySource* keySource, K* keysOut, int32_t* valuesOut, int n,
                  int shift, bool skewed, RadixScratch& s, int32_t* lastPassRanges)
{

    constexpr int kTile = kThreads * ITEMS;
    const int numTiles = (n + kTile - 1) / kTile;
    const int stride = (numTiles + 3) & ~3; // tiles per digit row of the tables
    const dim3 histGrid((numTiles + kHistTiles - 1) / kHistTiles);

    if constexpr (sizeof(K) == 4 && BITS == kMaxDigitBits)
    {
        // the depth sort's first pass may have to make its keys itself (Morton storage): they land in keysIn, where the scatter reads them
        if (keySource != nullptr)
            hipLaunchKernelGGL((radixHistogramKernel<K, BITS, ITEMS, DepthKeySource>), histGrid, dim3(kHistThreadsFor<true>), 0, stream, keysIn, *keySource,
                               const_cast<K*>(keysIn), n, s.countDevice, shift, 0, s.table, stride);
    }
    if (keySource == nullptr || !(sizeof(K) == 4 && BITS == kMaxDigitBits))
        hipLaunchKernelGGL((radixHistogramKernel<K, BITS, ITEMS, NoKeySource>), histGrid, dim3(kHistThreadsFor<false>), 0, stream, keysIn, NoKeySource{},
                           (K*) nullptr, n, s.countDevice, shift, skewed ? 1 : 0, s.table, stride);
#ifndef GS_SCAN_ROW_LIMIT
#define GS_SCAN_ROW_LIMIT 4096 // A/B builds: 0 = always 1024 lanes
#endif
    if (s.firstScanCompanion.launch != nullptr && numTiles <= GS_SCAN_ROW_LIMIT)
    {
        // the caller has a kernel that does not depend on this sort and shares the launch of its first column scan (gs_internal.h)
        const RadixScanArgs scan{BITS > 8 ? 2 * kDigits : kDigits, n, s.countDevice, kTile, stride, s.table, s.scanned, s.chunkTotals};
        const RadixScanCompanion companion = s.firstScanCompanion;
        if (--s.firstScanCompanion.shots <= 0) s.firstScanCompanion = RadixScanCompanion{};
        companion.launch(stream, scan, companion.user);
    }
    else if (numTiles <= GS_SCAN_ROW_LIMIT)
        hipLaunchKernelGGL(radixColumnScanKernel<256>, dim3(BITS > 8 ? 2 * kDigits : kDigits), dim3(256), 0, stream, n, s.countDevice, kTile, stride, s.table, s.scanned, s.chunkTotals);
    else
        hipLaunchKernelGGL(radixColumnScanKernel<1024>, dim3(BITS > 8 ? 2 * kDigits : kDigits), dim3(1024), 0, stream, n, s.countDevice, kTile, stride, s.table, s.scanned, s.chunkTotals);
    const dim3 scatterGrid(8 * ((numTiles + 7) / 8));
    if (s.carry[0] != nullptr)
    {
        // the staged builder's pairs: depths travel with the values; the last pass turns them into 64-bit keys
        hipLaunchKernelGGL((radixScatterKernel<K, BITS, ITEMS, true>), scatterGrid, dim3(kThreads), 0, stream, keysIn, valuesIn, keysOut, valuesOut, n,
                           s.countDevice, shift, s.scanned, s.chunkTotals, stride, lastPassRanges, (const uint32_t*) s.carry[s.carrySel],
                           s.carry[s.carrySel ^ 1], s.lastPassIsNow ? s.lastPassKeys64 : (uint64_t*) nullptr);
        s.carrySel ^= 1;
        return;
    }
    hipLaunchKernelGGL((radixScatterKernel<K, BITS, ITEMS, false>), scatterGrid, dim3(kThreads), 0, stream, keysIn, valuesIn, keysOut, valuesOut, n,
                       s.countDevice, shift, s.scanned, s.chunkTotals, stride, lastPassRanges, (const uint32_t*) nullptr, (uint32_t*) nullptr,
                       (uint64_t*) nullptr);
}

template <typename K, int BITS>
void runPass(hipStream_t stream, const K* keysIn, const int32_t* valuesIn, const DepthKeySource* keySource, K* keysOut, int32_t* valuesOut, int n,
             int shift, bool skewed, RadixScratch& s, int32_t* lastPassRanges)
{
    if (n <= kSmallLimit) runPassItems<K, BITS, kItemsSmall>(stream, keysIn, valuesIn, keySource, keysOut, valuesOut, n, shift, skewed, s, lastPassRanges);
    else runPassItems<K, BITS, kItemsLarge>(stream, keysIn, valuesIn, keySource, keysOut, valuesOut, n, shift, skewed, s, lastPassRanges);
}

template <typename K>
void runPassBits(hipStream_t stream, int bits, const K* keysIn, const int32_t* valuesIn, const DepthKeySource* keySource, K* keysOut, int32_t* valuesOut,
                 int n, int shift, bool skewed, RadixScratch& s, int32_t* lastPassRanges)
{
    switch (bits)
    {
    case 1: runPass<K, 1>(stream, keysIn, valuesIn, keySource, keysOut, valuesOut, n, shift, skewed, s, lastPassRanges); break;
    case 2: runPass<K, 2>(stream, keysIn, valuesIn, keySource, keysOut, valuesOut, n, shift, skewed, s, lastPassRanges); break;
    case 3: runPass<K, 3>(stream, keysIn, valuesIn, keySource, keysOut, valuesOut, n, shift, skewed, s, lastPassRanges); break;
    case 4: runPass<K, 4>(stream, keysIn, valuesIn, keySource, keysOut, valuesOut, n, shift, skewed, s, lastPassRanges); break;
    case 5: runPass<K, 5>(stream, keysIn, valuesIn, keySource, keysOut, valuesOut, n, shift, skewed, s, lastPassRanges); break;
    case 6: runPass<K, 6>(stream, keysIn, valuesIn, keySource, keysOut, valuesOut, n, shift, skewed, s, lastPassRanges); break;
    case 7: runPass<K, 7>(stream, keysIn, valuesIn, keySource, keysOut, valuesOut, n, shift, skewed, s, lastPassRanges); break;
    case 8: runPass<K, 8>(stream, keysIn, valuesIn, keySource, keysOut, valuesOut, n, shift, skewed, s, lastPassRanges); break;
    default: runPass<K, 9>(stream, keysIn, valuesIn, keySource, keysOut, valuesOut, n, shift, skewed, s, lastPassRanges); break;
    }
}

template <typename K>
int sortT(hipStream_t stream, K* keys0, K* keys1, int32_t* values0, int32_t* values1, int n, int beginBit, int endBit, RadixScratch& s,
          const int32_t* inputValues, const DepthKeySource* keySource)
{
    // As few passes as 9-bit digits allow (512 bins, two per lane: the 27-bit compact depth keys sort in three passes), the bits
    // spread evenly over them (13 tile bits -> 7 + 6, not 8 + 5): fewer digits per pass means longer runs of consecutive
    // output addresses per workgroup in BOTH passes.  9 bits are only used where they save a pass (32 bits stay 4 x 8).
    const int passes = radixSortPasses(beginBit, endBit);
    int sel = 0, shift = beginBit;
    for (int pass = 0; pass != passes; ++pass)
    {
        const int left = endBit - shift;
        const int bits = (left + (passes - pass) - 1) / (passes - pass);
        K* kin = sel ? keys1 : keys0;
        K* kout = sel ? keys0 : keys1;
        // (first pass: the values may come from a constant list of the caller's -- it is only read)
        const int32_t* vin = sel ? values1 : (pass == 0 && inputValues != nullptr ? inputValues : values0);
        int32_t* vout = sel ? values0 : values1;
        // the top digit of a multi-pass sort of tile indices is shared by most neighbours: count it through ballots
        // (beginBit > 0: the passes below it were done elsewhere -- the emit pass of the list builder carries the first one)
        const bool skewed = sizeof(K) == 2 && (passes > 1 || beginBit > 0) && pass == passes - 1 && bits < 8;
        s.lastPassIsNow = pass == passes - 1;
        runPassBits<K>(stream, bits, kin, vin, pass == 0 ? keySource : nullptr, kout, vout, n, shift, skewed, s,
                       pass == passes - 1 ? s.rangeTableForLastPass : nullptr);
        s.lastPassIsNow = false;
        shift += bits;
        sel ^= 1;
    }
    return sel;
}

__global__ __launch_bounds__(256) void gatherSortInputKernel(int n, const uint32_t* __restrict__ keysIn, const int32_t* __restrict__ inputOrder,
                                                             uint32_t* __restrict__ keysOut, int32_t* __restrict__ valuesOut)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int src = inputOrder[i];
    keysOut[i] = keysIn[src];
    valuesOut[i] = src;
}

} // namespace

void launchGatherSortInput(hipStream_t stream, int n, const uint32_t* keysIn, const int32_t* inputOrder, uint32_t* keysOut, int32_t* valuesOut)
{
    if (n > 0) hipLaunchKernelGGL(gatherSortInputKernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, keysIn, inputOrder, keysOut, valuesOut);
}

void launchRadixColumnScan(hipStream_t stream, int digits, int nHost, const int* nDevice, int tileElements, int stride, const int32_t* table,
                           int32_t* scanned, int32_t* totals)
{
    const int numTiles = (nHost + tileElements - 1) / tileElements;
    // (a device-side count is a culled list: a fraction of the rows the host-side bound would make -- 256 lanes walk it in a few rounds)
    if (numTiles <= GS_SCAN_ROW_LIMIT || nDevice != nullptr)
        hipLaunchKernelGGL(radixColumnScanKernel<256>, dim3(digits), dim3(256), 0, stream, nHost, nDevice, tileElements, stride, table, scanned, totals);
    else
        hipLaunchKernelGGL(radixColumnScanKernel<1024>, dim3(digits), dim3(1024), 0, stream, nHost, nDevice, tileElements, stride, table, scanned, totals);
}

int radixSortPasses(int beginBit, int endBit) { return endBit > beginBit ? (endBit - beginBit + kMaxDigitBits - 1) / kMaxDigitBits : 0; }

bool radixSortSupported(int keyBytes, int n) { return (keyBytes == 2 || keyBytes == 4) && n >= 0; }

void ensureRadixScratch(RadixScratch& s, int n)
{
    const int tile = kThreads * itemsPerLane(n);
    const int numTiles = (n + tile - 1) / tile;
    const int entries = 2 * kDigits * ((std::max(numTiles, 1) + 3) & ~3); // up to 512 digit rows
    if (entries > s.capacityEntries)
    {
        if (s.table) GS_CHECK(hipFree(s.table));
        if (s.scanned) GS_CHECK(hipFree(s.scanned));
        GS_CHECK(hipMalloc((void**) &s.table, sizeof(int32_t) * (size_t) entries));
        GS_CHECK(hipMalloc((void**) &s.scanned, sizeof(int32_t) * (size_t) entries));
        s.capacityEntries = entries;
    }
    if (s.chunkTotals == nullptr) GS_CHECK(hipMalloc((void**) &s.chunkTotals, sizeof(int32_t) * 2 * kDigits));
}

void freeRadixScratch(RadixScratch& s)
{
    if (s.table) (void) hipFree(s.table);
    if (s.scanned) (void) hipFree(s.scanned);
    if (s.chunkTotals) (void) hipFree(s.chunkTotals);
    s = RadixScratch{};
}

int launchRadixSortPairs(hipStream_t stream, RadixScratch& s, int keyBytes, void* keys0, void* keys1, int32_t* values0, int32_t* values1,
                         int n, int beginBit, int endBit, const int32_t* inputValues, const DepthKeySource* keySource)
{
    if (n <= 0 || endBit <= beginBit) return 0;
    ensureRadixScratch(s, n);
    if (keyBytes == 2) return sortT<uint16_t>(stream, (uint16_t*) keys0, (uint16_t*) keys1, values0, values1, n, beginBit, endBit, s, inputValues, nullptr);
    // (key generation is the depth sort's: 32-bit keys whose first pass is a 9-bit one)
    if (keySource != nullptr && !(beginBit == 0 && endBit == kDepthKeyBits)) keySource = nullptr;
    return sortT<uint32_t>(stream, (uint32_t*) keys0, (uint32_t*) keys1, values0, values1, n, beginBit, endBit, s, inputValues, keySource);
}

} // namespace gs
