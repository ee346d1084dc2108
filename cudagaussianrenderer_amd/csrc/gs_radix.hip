// gs_radix.hip -- stable LSD radix sort of (16/32-bit key, 32-bit value) pairs for the renderer's two sorts
// (splats by 32-bit depth key; depth-ordered pairs by 13-15 tile bits).
//
// rocPRIM's device radix sort stays behind the drop-in sortTileList (64-bit keys, gs_sort.hip).  For the
// renderer's element counts (1 M splats, ~5 M pairs) it spends most of its time outside the streaming
// passes (merge-sort fallback at 1 M elements, two memsets + look-back state per onesweep pass), so the
// renderer uses this plain three-kernel pass instead; every kernel is barrier-free across workgroups (no
// look-back, no spinning), so there is nothing that can hang:
//   1. histogram  -- one workgroup per 4096-element tile counts its 8-bit digits (wave-aggregated through
//                    ballot matching, so a skewed digit does not serialise on one LDS word) -> table[digit][tile];
//   2. chunk scan -- exclusive scan of the table in 4096-entry chunks + one total per chunk;
//   3. scatter    -- the tile is re-read (keys and values in registers), ranked stably (ballot match + per-wave
//                    running counters in LDS), reordered by digit in LDS, and written out as runs of consecutive
//                    addresses (base = chunk prefix + scanned table entry).
// HBM traffic per pass and element: key read twice, value read once, key + value written once.
#include "gs_internal.h"

namespace gs
{
namespace
{

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / kWave;
constexpr int kItems = 16;
constexpr int kTile = kThreads * kItems; // 4096 elements per workgroup
constexpr int kDigits = 256;
constexpr int kScanChunk = 4096;         // table entries per scan workgroup (1024 lanes x 4)
constexpr int kMaxChunks = 1024;

// Lanes of the wave whose (valid) digit equals this lane's digit.
template <int BITS>
__device__ __forceinline__ unsigned long long matchDigit(int digit, bool valid)
{
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit != BITS; ++bit)
    {
        const bool b = ((digit >> bit) & 1) != 0;
        const unsigned long long bal = __ballot(b && valid);
        m &= b ? bal : ~bal;
    }
    return valid ? m : 0ull;
}

template <typename K, int BITS>
__global__ __launch_bounds__(kThreads) void radixHistogramKernel(const K* __restrict__ keys, int n, int shift, int numTiles,
                                                                  int32_t* __restrict__ table)
{
    __shared__ int hist[kDigits];
    const int t = threadIdx.x, lane = t & (kWave - 1);
    hist[t] = 0;
    __syncthreads();
    const int base = blockIdx.x * kTile;
    const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (kWave - lane));
#pragma unroll 4
    for (int r = 0; r != kItems; ++r)
    {
        const int i = base + r * kThreads + t;
        const bool valid = i < n;
        const int digit = valid ? (int) ((keys[i] >> shift) & ((1 << BITS) - 1)) : 0;
        const unsigned long long m = matchDigit<BITS>(digit, valid);
        if (valid && (m & lower) == 0ull) atomicAdd(&hist[digit], __popcll(m)); // one add per distinct digit per wave
    }
    __syncthreads();
    table[t * numTiles + blockIdx.x] = hist[t];
}

// Exclusive scan of `entries` ints in chunks of kScanChunk; chunk totals to chunkTotals.
__global__ __launch_bounds__(1024) void radixChunkScanKernel(int entries, const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                             int32_t* __restrict__ chunkTotals)
{
    __shared__ int waveTotals[16];
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
    const int i = blockIdx.x * kScanChunk + t * 4;
    int4 v = make_int4(0, 0, 0, 0);
    if (i + 3 < entries) v = *reinterpret_cast<const int4*>(in + i);
    else
    {
        if (i < entries) v.x = in[i];
        if (i + 1 < entries) v.y = in[i + 1];
        if (i + 2 < entries) v.z = in[i + 2];
    }
    const int own = v.x + v.y + v.z + v.w;
    int incl = own;
#pragma unroll
    for (int delta = 1; delta < kWave; delta <<= 1)
    {
        const int o = __shfl_up(incl, delta);
        if (lane >= delta) incl += o;
    }
    if (lane == kWave - 1) waveTotals[wave] = incl;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int wv = 0; wv != 16; ++wv)
    {
        const int wt = waveTotals[wv];
        before += (wv < wave) ? wt : 0;
        all += wt;
    }
    int p = before + incl - own;
    if (i + 3 < entries) *reinterpret_cast<int4*>(out + i) = make_int4(p, p + v.x, p + v.x + v.y, p + v.x + v.y + v.z);
    else
    {
        if (i < entries) out[i] = p;
        if (i + 1 < entries) out[i + 1] = p + v.x;
        if (i + 2 < entries) out[i + 2] = p + v.x + v.y;
    }
    if (t == 0) chunkTotals[blockIdx.x] = all;
}

template <typename K, int BITS>
__global__ __launch_bounds__(kThreads) void radixScatterKernel(const K* __restrict__ keysIn, const int32_t* __restrict__ valuesIn,
                                                                K* __restrict__ keysOut, int32_t* __restrict__ valuesOut, int n,
                                                                int shift, int numTiles, const int32_t* __restrict__ scanned,
                                                                const int32_t* __restrict__ chunkTotals, int numChunks)
{
    __shared__ int cnt[kWaves][kDigits];   // per-wave running digit counters, then per-wave exclusive prefixes
    __shared__ int localBase[kDigits];     // first slot of each digit inside the tile
    __shared__ int globalBase[kDigits];    // first output index of each digit for this tile
    __shared__ int chunkPrefix[kMaxChunks];
    __shared__ int waveSums[kWaves];
    __shared__ uint32_t sKeys[kTile];
    __shared__ int32_t sVals[kTile];

    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
    const int tile = blockIdx.x;
    const int base = tile * kTile;
    const int tileCount = min(kTile, n - base);
    constexpr int kDigitMask = (1 << BITS) - 1;

#pragma unroll
    for (int w = 0; w != kWaves; ++w) cnt[w][t] = 0;
    // exclusive prefix of the chunk totals (numChunks <= 1024): 4 per lane + workgroup scan
    {
        int c[4], s = 0;
#pragma unroll
        for (int j = 0; j != 4; ++j)
        {
            const int idx = t * 4 + j;
            c[j] = idx < numChunks ? chunkTotals[idx] : 0;
            s += c[j];
        }
        int incl = s;
#pragma unroll
        for (int delta = 1; delta < kWave; delta <<= 1)
        {
            const int o = __shfl_up(incl, delta);
            if (lane >= delta) incl += o;
        }
        if (lane == kWave - 1) waveSums[wave] = incl;
        __syncthreads();
        int before = 0;
#pragma unroll
        for (int w = 0; w != kWaves; ++w) before += (w < wave) ? waveSums[w] : 0;
        int p = before + incl - s;
#pragma unroll
        for (int j = 0; j != 4; ++j)
        {
            chunkPrefix[t * 4 + j] = p;
            p += c[j];
        }
    }
    __syncthreads();

    // ---- load the tile: element (wave, row, lane) is tile index wave*1024 + row*64 + lane ------------
    K key[kItems];
    int32_t val[kItems];
    int rank[kItems];
    const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (kWave - lane));
#pragma unroll
    for (int r = 0; r != kItems; ++r)
    {
        const int i = base + wave * (kWave * kItems) + r * kWave + lane;
        const bool valid = i < n;
        key[r] = valid ? keysIn[i] : K(0);
        val[r] = valid ? valuesIn[i] : 0;
    }
    // ---- stable rank inside the wave: running per-digit counters -------------------------------------
#pragma unroll
    for (int r = 0; r != kItems; ++r)
    {
        const int i = base + wave * (kWave * kItems) + r * kWave + lane;
        const bool valid = i < n;
        const int digit = (int) ((key[r] >> shift) & kDigitMask);
        const unsigned long long m = matchDigit<BITS>(digit, valid);
        const int leader = valid ? (__ffsll((long long) m) - 1) : lane;
        int old = 0;
        if (valid && lane == leader)
        {
            old = cnt[wave][digit];
            cnt[wave][digit] = old + __popcll(m);
        }
        old = __shfl(old, leader);
        rank[r] = old + __popcll(m & lower);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();

    // ---- per digit (lane t owns digit t): prefixes over waves, tile-local base, global base ----------
    {
        int sum = 0;
#pragma unroll
        for (int w = 0; w != kWaves; ++w)
        {
            const int c = cnt[w][t];
            cnt[w][t] = sum;
            sum += c;
        }
        int incl = sum;
#pragma unroll
        for (int delta = 1; delta < kWave; delta <<= 1)
        {
            const int o = __shfl_up(incl, delta);
            if (lane >= delta) incl += o;
        }
        if (lane == kWave - 1) waveSums[wave] = incl;
        __syncthreads();
        int before = 0;
#pragma unroll
        for (int w = 0; w != kWaves; ++w) before += (w < wave) ? waveSums[w] : 0;
        localBase[t] = before + incl - sum;
        const int entry = t * numTiles + tile;
        globalBase[t] = (t <= kDigitMask) ? scanned[entry] + chunkPrefix[entry / kScanChunk] : 0;
    }
    __syncthreads();

    // ---- reorder by digit in LDS ----------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r != kItems; ++r)
    {
        const int i = base + wave * (kWave * kItems) + r * kWave + lane;
        if (i < n)
        {
            const int digit = (int) ((key[r] >> shift) & kDigitMask);
            const int slot = localBase[digit] + cnt[wave][digit] + rank[r];
            sKeys[slot] = (uint32_t) key[r];
            sVals[slot] = val[r];
        }
    }
    __syncthreads();

    // ---- write runs ---------------------------------------------------------------------------------
#pragma unroll 4
    for (int j = t; j < tileCount; j += kThreads)
    {
        const uint32_t k = sKeys[j];
        const int digit = (int) ((k >> shift) & kDigitMask);
        const int pos = globalBase[digit] + (j - localBase[digit]);
        keysOut[pos] = (K) k;
        valuesOut[pos] = sVals[j];
    }
}

template <typename K, int BITS>
void runPass(hipStream_t stream, const K* keysIn, const int32_t* valuesIn, K* keysOut, int32_t* valuesOut, int n, int shift,
             RadixScratch& s)
{
    const int numTiles = (n + kTile - 1) / kTile;
    const int entries = kDigits * numTiles;
    const int numChunks = (entries + kScanChunk - 1) / kScanChunk;
    hipLaunchKernelGGL((radixHistogramKernel<K, BITS>), dim3(numTiles), dim3(kThreads), 0, stream, keysIn, n, shift, numTiles, s.table);
    hipLaunchKernelGGL(radixChunkScanKernel, dim3(numChunks), dim3(1024), 0, stream, entries, s.table, s.scanned, s.chunkTotals);
    hipLaunchKernelGGL((radixScatterKernel<K, BITS>), dim3(numTiles), dim3(kThreads), 0, stream, keysIn, valuesIn, keysOut, valuesOut, n,
                       shift, numTiles, s.scanned, s.chunkTotals, numChunks);
}

template <typename K>
void runPassBits(hipStream_t stream, int bits, const K* keysIn, const int32_t* valuesIn, K* keysOut, int32_t* valuesOut, int n,
                 int shift, RadixScratch& s)
{
    switch (bits)
    {
    case 1: runPass<K, 1>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, s); break;
    case 2: runPass<K, 2>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, s); break;
    case 3: runPass<K, 3>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, s); break;
    case 4: runPass<K, 4>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, s); break;
    case 5: runPass<K, 5>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, s); break;
    case 6: runPass<K, 6>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, s); break;
    case 7: runPass<K, 7>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, s); break;
    default: runPass<K, 8>(stream, keysIn, valuesIn, keysOut, valuesOut, n, shift, s); break;
    }
}

template <typename K>
int sortT(hipStream_t stream, K* keys0, K* keys1, int32_t* values0, int32_t* values1, int n, int beginBit, int endBit, RadixScratch& s)
{
    int sel = 0;
    for (int shift = beginBit; shift < endBit; shift += 8)
    {
        const int bits = std::min(8, endBit - shift);
        K* kin = sel ? keys1 : keys0;
        K* kout = sel ? keys0 : keys1;
        int32_t* vin = sel ? values1 : values0;
        int32_t* vout = sel ? values0 : values1;
        runPassBits<K>(stream, bits, kin, vin, kout, vout, n, shift, s);
        sel ^= 1;
    }
    return sel;
}

} // namespace

bool radixSortSupported(int keyBytes, int n)
{
    const long long numTiles = ((long long) n + kTile - 1) / kTile;
    return (keyBytes == 2 || keyBytes == 4) && numTiles * kDigits <= (long long) kMaxChunks * kScanChunk;
}

void ensureRadixScratch(RadixScratch& s, int n)
{
    const int numTiles = (n + kTile - 1) / kTile;
    const int entries = kDigits * std::max(numTiles, 1);
    if (entries > s.capacityEntries)
    {
        if (s.table) GS_CHECK(hipFree(s.table));
        if (s.scanned) GS_CHECK(hipFree(s.scanned));
        GS_CHECK(hipMalloc((void**) &s.table, sizeof(int32_t) * (size_t) entries));
        GS_CHECK(hipMalloc((void**) &s.scanned, sizeof(int32_t) * (size_t) entries));
        s.capacityEntries = entries;
    }
    if (s.chunkTotals == nullptr) GS_CHECK(hipMalloc((void**) &s.chunkTotals, sizeof(int32_t) * kMaxChunks));
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
