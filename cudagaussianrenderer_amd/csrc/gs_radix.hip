// gs_radix.hip -- stable LSD radix sort of (16/32-bit key, 32-bit value) pairs for the renderer's two sorts
// (splats by 32-bit depth key; depth-ordered pairs by 13-15 tile bits).
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
#include "gs_internal.h"

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

// Lanes of the wave whose (valid) digit equals this lane's digit.  Written on the two 32-bit halves of the mask with
// xnor/and against the ballot of each digit bit: seven VALU instructions per bit and no scalar temporaries beyond the
// ballot itself (the obvious `m &= b ? bal : ~bal` makes the compiler keep every ballot alive and spill SGPRs).
template <int BITS>
__device__ __forceinline__ unsigned long long matchDigit(int digit, bool valid)
{
    const unsigned long long all = __ballot(valid);
    unsigned lo = (unsigned) all, hi = (unsigned) (all >> 32);
#pragma unroll
    for (int bit = 0; bit != BITS; ++bit)
    {
        // lanes outside `valid` may vote anything: they were removed from the mask above and stay removed
        const unsigned long long bal = __ballot(((unsigned) digit & (1u << bit)) != 0u);
        const unsigned mine = 0u - (((unsigned) digit >> bit) & 1u); // all ones when my bit is set
        lo &= ~((unsigned) bal ^ mine);
        hi &= ~((unsigned) (bal >> 32) ^ mine);
    }
    const unsigned long long m = ((unsigned long long) hi << 32) | lo;
    return valid ? m : 0ull;
}

// Tiles counted by one histogram workgroup: its four counts of a digit are one 16-byte store into the digit's row of
// table[digit][tile] (a single tile per workgroup would scatter 256 four-byte stores a row apart).
constexpr int kHistTiles = 4;

template <typename K, int BITS, int ITEMS>
__global__ __launch_bounds__(kThreads) void radixHistogramKernel(const K* __restrict__ keys, int nHost, const int* __restrict__ nDevice, int shift,
                                                                  int skewed, int32_t* __restrict__ table, int stride)
{
    constexpr int kTile = kThreads * ITEMS;
    constexpr int DIGITS = BITS > 8 ? 2 * kDigits : kDigits; // 9-bit digits: every lane owns two bins
    __shared__ int hist[kHistTiles][DIGITS];
    const int t = threadIdx.x;
    // The element count may live on the device (the renderer's band-culled splat list): nHost is then only the
    // upper bound the grid was sized for; tiles past the real count do nothing.
    const int n = nDevice ? min(*nDevice, nHost) : nHost;
    const int firstTile = blockIdx.x * kHistTiles;
    if (firstTile * kTile >= n) return;
#pragma unroll
    for (int j = 0; j != kHistTiles; ++j)
#pragma unroll
        for (int d = t; d < DIGITS; d += kThreads) hist[j][d] = 0;
    __syncthreads();
    for (int j = 0; j != kHistTiles; ++j)
    {
        const int base = (firstTile + j) * kTile;
        if (base >= n) break;
        int* h = hist[j];
        K key[ITEMS];
        // Counting does not care which lane sees which key: full tiles are read 16 bytes per lane (ITEMS consecutive keys
        // when they fill a multiple of 16 bytes), the ragged last tile element by element.
        constexpr bool kWide = (sizeof(K) * ITEMS) % 16 == 0;
        const bool wide = kWide && base + kTile <= n;
        if constexpr (kWide)
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
        if (!wide)
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
                    const int lane = t & (kWave - 1);
                    const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (kWave - lane));
                    const unsigned long long m = matchDigit<BITS>(digit, valid);
                    if (valid && (m & lower) == 0ull) atomicAdd(&h[digit], __popcll(m));
                }
            }
        }
    }
    __syncthreads();
    // [digit][tile]: every digit's row is contiguous for the scan; stride is a multiple of 4, so the quad is aligned
#pragma unroll
    for (int d = t; d < DIGITS; d += kThreads)
        *reinterpret_cast<int4*>(table + (size_t) d * stride + firstTile) = make_int4(hist[0][d], hist[1][d], hist[2][d], hist[3][d]);
}

// One workgroup per digit: exclusive scan of that digit's row of table[digit][tile] over the tiles (contiguous: 16 bytes per
// lane and round), row total to totals[digit].  256 lanes for rows of up to 4096 tiles (one to four rounds), 1024 lanes beyond:
// the kernel is pure latency, and with 1024 lanes per digit its 256-512 workgroups took every wave slot of the chip, so that in
// the pipelined frame loop nothing of the other frames could run beside it (17 us per launch there against 5 us alone).
template <int kScanThreads>
__global__ __launch_bounds__(kScanThreads) void radixColumnScanKernel(int nHost, const int* __restrict__ nDevice, int tileElements, int stride,
                                                                      const int32_t* __restrict__ table, int32_t* __restrict__ scanned,
                                                                      int32_t* __restrict__ totals)
{
    constexpr int kScanWaves = kScanThreads / kWave;
    __shared__ int waveSums[2][kScanWaves];
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
    const int digit = blockIdx.x;
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
        int incl = own;
#pragma unroll
        for (int delta = 1; delta < kWave; delta <<= 1)
        {
            const int o = __shfl_up(incl, delta);
            if (lane >= delta) incl += o;
        }
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

template <typename K, int BITS, int ITEMS>
__global__ __launch_bounds__(kThreads) void radixScatterKernel(const K* __restrict__ keysIn, const int32_t* __restrict__ valuesIn,
                                                                K* __restrict__ keysOut, int32_t* __restrict__ valuesOut, int nHost, const int* __restrict__ nDevice,
                                                                int shift, const int32_t* __restrict__ scanned,
                                                                const int32_t* __restrict__ totals, int stride)
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
#pragma unroll
    for (int r = 0; r != ITEMS; ++r)
    {
        const int i = base + wave * (kWave * ITEMS) + r * kWave + lane;
        const bool valid = i < n;
        key[r] = valid ? keysIn[i] : K(0);
        val[r] = valid ? valuesIn[i] : 0;
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
    const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (kWave - lane));
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
        const int leader = valid ? (__ffsll((long long) m) - 1) : lane;
        // every lane of a digit group reads the running count (one broadcast read), then the group's first lane
        // bumps it: LDS instructions of a wave execute in order, so the next row sees the bumped value
        const int old = valid ? cnt[wave][digit] : 0;
        if (valid && lane == leader) cnt[wave][digit] = old + __popcll(m);
        rank[r] = old + __popcll(m & lower);
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
        int inclLocal = sumAll, inclGlobal = totalAll;
#pragma unroll
        for (int delta = 1; delta < kWave; delta <<= 1)
        {
            const int a = __shfl_up(inclLocal, delta);
            const int b = __shfl_up(inclGlobal, delta);
            if (lane >= delta)
            {
                inclLocal += a;
                inclGlobal += b;
            }
        }
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
            keysOut[pos] = k;
            valuesOut[pos] = sVals[j];
        }
    }
}

int itemsPerLane(int n) { return n <= kSmallLimit ? kItemsSmall : kItemsLarge; }

template <typename K, int BITS, int ITEMS>
void runPassItems(hipStream_t stream, const K* keysIn, const int32_t* valuesIn, K* keysOut, int32_t* valuesOut, int n, int shift,
                  bool skewed, RadixScratch& s)
{
    constexpr int kTile = kThreads * ITEMS;
    const int numTiles = (n + kTile - 1) / kTile;
    const int stride = (numTiles + 3) & ~3; // tiles per digit row of the tables
    hipLaunchKernelGGL((radixHistogramKernel<K, BITS, ITEMS>), dim3((numTiles + kHistTiles - 1) / kHistTiles), dim3(kThreads), 0, stream, keysIn, n, s.countDevice, shift, skewed ? 1 : 0, s.table, stride);
#ifndef GS_SCAN_ROW_LIMIT
#define GS_SCAN_ROW_LIMIT 4096 // A/B builds: 0 = always 1024 lanes
#endif
    if (numTiles <= GS_SCAN_ROW_LIMIT)
        hipLaunchKernelGGL(radixColumnScanKernel<256>, dim3(BITS > 8 ? 2 * kDigits : kDigits), dim3(256), 0, stream, n, s.countDevice, kTile, stride, s.table, s.scanned, s.chunkTotals);
    else
        hipLaunchKernelGGL(radixColumnScanKernel<1024>, dim3(BITS > 8 ? 2 * kDigits : kDigits), dim3(1024), 0, stream, n, s.countDevice, kTile, stride, s.table, s.scanned, s.chunkTotals);
    hipLaunchKernelGGL((radixScatterKernel<K, BITS, ITEMS>), dim3(8 * ((numTiles + 7) / 8)), dim3(kThreads), 0, stream, keysIn, valuesIn, keysOut,
                       valuesOut, n, s.countDevice, shift, s.scanned, s.chunkTotals, stride);
}

template <typename K, int BITS>
void runPass(hipStream_t stream, const K* keysIn, const int32_t* valuesIn, K* keysOut, int32_t* valuesOut, int n, int shift,
             bool skewed, RadixScratch& s)
{
    if (n <= kSmallLimit) runPassItems<K, BITS, kItemsSmall>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, skewed, s);
    else runPassItems<K, BITS, kItemsLarge>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, skewed, s);
}

template <typename K>
void runPassBits(hipStream_t stream, int bits, const K* keysIn, const int32_t* valuesIn, K* keysOut, int32_t* valuesOut, int n,
                 int shift, bool skewed, RadixScratch& s)
{
    switch (bits)
    {
    case 1: runPass<K, 1>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, skewed, s); break;
    case 2: runPass<K, 2>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, skewed, s); break;
    case 3: runPass<K, 3>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, skewed, s); break;
    case 4: runPass<K, 4>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, skewed, s); break;
    case 5: runPass<K, 5>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, skewed, s); break;
    case 6: runPass<K, 6>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, skewed, s); break;
    case 7: runPass<K, 7>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, skewed, s); break;
    case 8: runPass<K, 8>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, skewed, s); break;
    default: runPass<K, 9>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, skewed, s); break;
    }
}

template <typename K>
int sortT(hipStream_t stream, K* keys0, K* keys1, int32_t* values0, int32_t* values1, int n, int beginBit, int endBit, RadixScratch& s)
{
    // As few passes as 9-bit digits allow (512 bins, two per lane: the 27-bit compact depth keys sort in three passes), the bits
    // spread evenly over them (13 tile bits -> 7 + 6, not 8 + 5): fewer digits per pass means longer runs of consecutive
    // output addresses per workgroup in BOTH passes.  9 bits are only used where they save a pass (32 bits stay 4 x 8).
    const int totalBits = endBit - beginBit;
    const int passes = (totalBits + kMaxDigitBits - 1) / kMaxDigitBits;
    int sel = 0, shift = beginBit;
    for (int pass = 0; pass != passes; ++pass)
    {
        const int left = endBit - shift;
        const int bits = (left + (passes - pass) - 1) / (passes - pass);
        K* kin = sel ? keys1 : keys0;
        K* kout = sel ? keys0 : keys1;
        int32_t* vin = sel ? values1 : values0;
        int32_t* vout = sel ? values0 : values1;
        // the top digit of a multi-pass sort of tile indices is shared by most neighbours: count it through ballots
        const bool skewed = sizeof(K) == 2 && passes > 1 && pass == passes - 1 && bits < 8;
        runPassBits<K>(stream, bits, kin, vin, kout, vout, n, shift, skewed, s);
        shift += bits;
        sel ^= 1;
    }
    return sel;
}

// Ties of the depth sort, re-ordered by the CALLER's splat index.  The sort is stable, so splats with equal depth codes come out in
// the order they were stored in; a renderer that stores its splats along a Morton curve (gs_renderer_set_storage_order) must still
// break ties the way a renderer with the caller's order does -- by the caller's index -- or two overlapping splats of equal depth
// would blend in another order.  Ties are the rule, not the exception: getKey's (z + 1) / 2 in float leaves ~0.8 M distinct codes
// for the 1 M splats of the bench scene, 38 % of the entries sit in a run of two to about twenty equal keys.
//
// One lane per sorted position.  A workgroup stages its 256 (key, entry) pairs and kTieHalo neighbours on both sides in LDS; every
// staged entry that has an equal neighbour fetches its caller index ONCE (one gather per tied entry instead of one per pair of run
// members) and leaves it in LDS too; a lane then finds its run in the window, counts the members that precede its own entry in the
// caller's order and stores its entry at that rank.  A run that reaches beyond the halo takes a loop over global memory, in slices
// whose loads are in flight together (the lane that walks the longest run sets the kernel's duration); runs of more than kMaxTieRun
// entries (a scene with thousands of splats at exactly one depth) keep the storage order -- every lane of a run comes to the same
// verdict.  No atomics, no order-dependent step.  Measured at 1 M entries: 15 us, of which 6 are the launch floor and the rest four
// dependent memory round trips per wave, two rounds of waves (profiles/r03_storage_order.txt).  Key 0 is the code of the splats outside the frustum (clip depth -128), which emit
// no pairs: their run -- the longest of every frame -- is copied as it is.  Several views per chain: a key's entries of view 0 come
// first, then view 1's, ... (stable sort, view-major input), and only entries of ONE view can meet in a tile: the run that matters is
// (key, view).
constexpr int kMaxTieRun = 1024;
constexpr int kTieHalo = 16; // a run of up to 16 entries is always resolved from the LDS window
__global__ __launch_bounds__(256) void depthTieFixKernel(int nHost, const int* __restrict__ nDevice, const uint32_t* __restrict__ keys,
                                                         const int32_t* __restrict__ orderIn, int32_t* __restrict__ orderOut,
                                                         const int32_t* __restrict__ storageToSplat, int splatsPerView, int views)
{
    constexpr int kWindow = 256 + 2 * kTieHalo;
    __shared__ uint32_t sKey[kWindow + 2]; // one sentinel on either side
    __shared__ int sOrd[kWindow + 2];
    __shared__ int sCaller[kWindow + 2];   // caller index of the entries that sit in a run (view-local)
    const int n = nDevice ? min(*nDevice, nHost) : nHost;
    const int t = threadIdx.x, base = blockIdx.x * 256;
    if (base >= n) return; // whole workgroup
    for (int idx = t; idx < kWindow + 2; idx += 256)
    {
        const int g = base - kTieHalo - 1 + idx;
        const bool live = g >= 0 && g < n && idx != 0 && idx != kWindow + 1;
        sKey[idx] = live ? keys[g] : 0xFFFFFFFFu; // (no key has 32 bits: the ends of the window end every run)
        sOrd[idx] = live ? orderIn[g] : -1;
    }
    __syncthreads();
    // view of a (virtual) entry; one view: no division
    const auto viewBase = [&](int entry) { return views > 1 ? (entry / splatsPerView) * splatsPerView : 0; };
    const auto tied = [&](int a, int b) { return sKey[a] == sKey[b] && (views == 1 || viewBase(sOrd[a]) == viewBase(sOrd[b])); };
    for (int idx = 1 + t; idx <= kWindow; idx += 256)
    {
        // (an entry at the window's edge is only asked about the neighbour the window holds: a run that leaves the window is not
        // resolved from it anyway)
        const bool inRun = sKey[idx] != 0u && sOrd[idx] >= 0 && (tied(idx, idx - 1) || tied(idx, idx + 1));
        if (inRun) sCaller[idx] = storageToSplat[sOrd[idx] - viewBase(sOrd[idx])];
    }
    __syncthreads();
    const int i = base + t, l = t + kTieHalo + 1;
    if (i >= n) return;
    const uint32_t key = sKey[l];
    const int mine = sOrd[l];
    int before = 0, after = 0; // run members left / right of this lane, as far as the window shows
    if (key != 0u)
    {
        while (before < kTieHalo && tied(l, l - before - 1)) ++before;
        while (after < kTieHalo && tied(l, l + after + 1)) ++after;
    }
    if (before + after == 0)
    {
        orderOut[i] = mine;
        return;
    }
    const int firstOfView = viewBase(mine), lastOfView = firstOfView + splatsPerView - 1;
    const int own = sCaller[l];
    if (before < kTieHalo && after < kTieHalo)
    {
        // the whole run lies inside the window: rank among its members
        int rank = 0;
        for (int d = 1; d <= before; ++d) rank += sCaller[l - d] < own ? 1 : 0;
        for (int d = 1; d <= after; ++d) rank += sCaller[l + d] < own ? 1 : 0;
        orderOut[i - before + rank] = mine;
        return;
    }
    // a run that leaves the window: its extent from global memory, as far as kMaxTieRun entries either way
    const auto same = [&](int j) { return keys[j] == key && orderIn[j] >= firstOfView && orderIn[j] <= lastOfView; };
    // (sixteen entries per step, their loads in flight together: the lane that walks the longest run sets the kernel's duration)
    int begin = i - before, end = i + after + 1;
    for (bool more = true; more && begin > 0 && i - begin < kMaxTieRun;)
    {
        uint32_t k16[16];
        int o16[16];
#pragma unroll
        for (int k = 0; k != 16; ++k)
        {
            const int j = max(begin - 1 - k, 0);
            k16[k] = keys[j];
            o16[k] = orderIn[j];
        }
        int run = 0;
#pragma unroll
        for (int k = 0; k != 16; ++k)
            if (run == k && begin - 1 - k >= 0 && k16[k] == key && o16[k] >= firstOfView && o16[k] <= lastOfView) run = k + 1;
        begin -= run;
        more = run == 16;
    }
    for (bool more = true; more && end < n && end - i <= kMaxTieRun;)
    {
        uint32_t k16[16];
        int o16[16];
#pragma unroll
        for (int k = 0; k != 16; ++k)
        {
            const int j = min(end + k, n - 1);
            k16[k] = keys[j];
            o16[k] = orderIn[j];
        }
        int run = 0;
#pragma unroll
        for (int k = 0; k != 16; ++k)
            if (run == k && end + k < n && k16[k] == key && o16[k] >= firstOfView && o16[k] <= lastOfView) run = k + 1;
        end += run;
        more = run == 16;
    }
    // the same verdict in every lane of a run: its two ends were found AND it is at most kMaxTieRun entries long (a lane in the middle
    // of a longer run may see both ends where a lane at its edge cannot)
    const bool whole = (begin == 0 || !same(begin - 1)) && (end == n || !same(end)) && end - begin <= kMaxTieRun;
    if (!whole)
    {
        orderOut[i] = mine;
        return;
    }
    int rank = 0;
    for (int j = begin; j < end; j += 16) // sixteen members at a time: their entries first, then their caller indices, each set in flight together
    {
        int member[16];
#pragma unroll
        for (int k = 0; k != 16; ++k) member[k] = j + k < end ? orderIn[j + k] : mine;
#pragma unroll
        for (int k = 0; k != 16; ++k) member[k] = storageToSplat[member[k] - firstOfView];
#pragma unroll
        for (int k = 0; k != 16; ++k) rank += member[k] < own ? 1 : 0;
    }
    orderOut[begin + rank] = mine;
}

} // namespace

void launchDepthTieFix(hipStream_t stream, int n, const int* countDevice, const uint32_t* sortedKeys, const int32_t* orderIn, int32_t* orderOut,
                       const int32_t* storageToSplat, int splatsPerView)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(depthTieFixKernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, countDevice, sortedKeys, orderIn, orderOut,
                       storageToSplat, splatsPerView, (n + splatsPerView - 1) / splatsPerView);
}

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
                         int n, int beginBit, int endBit)
{
    if (n <= 0 || endBit <= beginBit) return 0;
    ensureRadixScratch(s, n);
    if (keyBytes == 2) return sortT<uint16_t>(stream, (uint16_t*) keys0, (uint16_t*) keys1, values0, values1, n, beginBit, endBit, s);
    return sortT<uint32_t>(stream, (uint32_t*) keys0, (uint32_t*) keys1, values0, values1, n, beginBit, endBit, s);
}

} // namespace gs
