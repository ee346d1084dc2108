// gs_tiles.hip -- splat -> (tile|depth key, splat index) pair list.
//
// Replaces buildTileListKernel + buildTileList (GaussianRender.cu:456-802).  Observable result is
// the same set { (key(tile, z_i), i) : tile in AABB_i and ellipseRectOverlap(ellipse_i, tile) }.
//
// CDNA4 design (not the reference's persistent 32-lane-loader scheme):
//   * one 64-lane wave owns 64 consecutive splats; every lane loads its splat with coalesced loads and
//     computes the tile AABB; a wave prefix sum turns the 64 AABB areas into a work-item list that the
//     whole wave walks 64 items at a time (lane = one (splat, tile) overlap test), so big and small
//     splats cost the same per tile and no block-level barrier or shared atomic is needed;
//   * hits are compacted with __ballot + popcount of the lower lanes;
//   * two passes -- COUNT (per-wave hit totals), a single-workgroup exclusive scan, EMIT -- so the list
//     is written in (splat, local tile) order with no global atomics.  That makes the list, and with a
//     stable sort the whole frame, bit-reproducible, lets the caller size the list exactly before
//     emitting (no overflow frame as in GaussianRender.cu:700-703), and is the tie-break the oracle uses.
//
// Floating-point contraction is OFF in this file: the overlap predicate is compared bit-for-bit with
// the CPU oracle, so every operation must round exactly as written.
#include "gs_depth_key.h"
#include "gs_internal.h"
#include "gs_overlap_exact.h"

#pragma clang fp contract(off)

namespace gs
{
namespace
{

constexpr int kWavesPerBlock = 4;

// glm min/max: min(x,y) = (y < x) ? y : x ; max(x,y) = (x < y) ? y : x  (NaN in y is ignored)
__device__ __forceinline__ float gmin(float x, float y) { return (y < x) ? y : x; }
__device__ __forceinline__ float gmax(float x, float y) { return (x < y) ? y : x; }

// v_cvt_i32_f32 semantics, spelled out so host and device agree: truncate, saturate, NaN -> 0.
__device__ __forceinline__ int f2iSat(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return -2147483647 - 1;
    return (int) v;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// quantizeDepth (the low word of getKey, GaussianRender.cu:446-454): gs_depth_key.h

// Per-wave staging of the 64 splats the wave owns.
struct WaveStage
{
    int excl[kWave];      // exclusive prefix of AABB areas
    int x0[kWave], y0[kWave], w[kWave];
    float cx[kWave], cy[kWave], ex[kWave], ey[kWave], cosA[kWave], sinA[kWave];
    uint32_t depth[kWave];
};

template <bool EMIT>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void tileListKernel(int count, Screen scr, const float2* __restrict__ clipXY,
                                                                         const float* __restrict__ clipZ,
                                                                         const float4* __restrict__ screenEllipse,
                                                                         int32_t* __restrict__ waveCounts,
                                                                         const int32_t* __restrict__ waveOffsets,
                                                                         uint64_t* __restrict__ keys, int32_t* __restrict__ values,
                                                                         int capacity)
{
    __shared__ WaveStage stage[kWavesPerBlock];
    const int waveInBlock = threadIdx.x >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int waveGlobal = blockIdx.x * kWavesPerBlock + waveInBlock;
    const int splat = waveGlobal * kWave + lane;
    if (waveGlobal * kWave >= count) return; // whole wave out of range (uniform)
    WaveStage& st = stage[waveInBlock];

    // ---- per-lane: load the splat, AABB in tiles (GaussianRender.cu:526-551) -----------------
    int area = 0, x0 = 0, y0 = 0, w = 0;
    EllipseD e{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint32_t depth = 0;
    if (splat < count)
    {
        const float2 c = clipXY[splat];
        const float4 se = screenEllipse[splat];
        e = EllipseD{c.x, c.y, se.z, se.w, se.x, se.y};
        if (EMIT) depth = quantizeDepth(clipZ[splat]);

        // getAABBRect (GaussianRender.cu:408-436)
        const float rightX = e.cosA * e.ex, rightY = e.sinA * e.ex;
        const float upX = e.sinA * e.ey, upY = -e.cosA * e.ey;
        float minX = 1e12f, minY = 1e12f, maxX = -1e12f, maxY = -1e12f;
#pragma unroll
        for (int j = 0; j != 4; ++j)
        {
            const int bottomBit = j & 1, topBit = j >> 1;
            const float r = (float) (bottomBit ^ topBit) * 2.0f - 1.0f;
            const float u = (float) topBit * 2.0f - 1.0f;
            const float vx = rightX * r + upX * u, vy = rightY * r + upY * u;
            minX = gmin(minX, vx);
            minY = gmin(minY, vy);
            maxX = gmax(maxX, vx);
            maxY = gmax(maxY, vy);
        }
        minX += e.cx;
        minY += e.cy;
        maxX += e.cx;
        maxY += e.cy;
        const float tx0 = (minX + 1.0f) * 0.5f * scr.tilesPerScreenX;
        const float ty0 = (minY + 1.0f) * 0.5f * scr.tilesPerScreenY;
        const float tx1 = (maxX + 1.0f) * 0.5f * scr.tilesPerScreenX;
        const float ty1 = (maxY + 1.0f) * 0.5f * scr.tilesPerScreenY;
        x0 = clampi(f2iSat(floorf(tx0)), 0, scr.tilesX);
        y0 = clampi(f2iSat(floorf(ty0)), 0, scr.tilesY);
        const int x1 = clampi(f2iSat(ceilf(tx1)), 0, scr.tilesX);
        int y1 = clampi(f2iSat(ceilf(ty1)), 0, scr.tilesY);
        // multi-GPU: only the tile rows of this band (tiles are independent, so this only removes pairs)
        y0 = max(y0, scr.rowBegin);
        y1 = min(y1, scr.rowEnd);
        w = x1 - x0;
        const int h = y1 - y0;
        area = (w > 0 && h > 0) ? w * h : 0;
    }

    // ---- wave inclusive scan of the areas ---------------------------------------------------
    int incl = area;
#pragma unroll
    for (int delta = 1; delta < kWave; delta <<= 1)
    {
        const int n = __shfl_up(incl, delta);
        if (lane >= delta) incl += n;
    }
    const int total = __builtin_amdgcn_readfirstlane(__shfl(incl, kWave - 1)); // wave-uniform -> scalar loop

    st.excl[lane] = incl - area;
    st.x0[lane] = x0;
    st.y0[lane] = y0;
    st.w[lane] = max(w, 1);
    st.cx[lane] = e.cx;
    st.cy[lane] = e.cy;
    st.ex[lane] = e.ex;
    st.ey[lane] = e.ey;
    st.cosA[lane] = e.cosA;
    st.sinA[lane] = e.sinA;
    if (EMIT) st.depth[lane] = depth;
    // Same-wave LDS traffic is executed in order; this only pins the compiler's schedule.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    int running = 0;
    int outBase = 0;
    if (EMIT) outBase = __builtin_amdgcn_readfirstlane(waveOffsets[waveGlobal]);
    const unsigned long long lowerLanes = (lane == 0) ? 0ull : (~0ull >> (kWave - lane));

    // ---- walk the work items, 64 per step (GaussianRender.cu:645-679) ------------------------
    for (int base = 0; base < total; base += kWave)
    {
        const int item = base + lane;
        const bool valid = item < total;
        // owner = last lane whose exclusive prefix is <= item (prefixes are non-decreasing)
        int k = 0;
#pragma unroll
        for (int step = kWave / 2; step != 0; step >>= 1)
        {
            const int t = k + step;
            if (st.excl[t] <= item) k = t;
        }
        const int local = item - st.excl[k];
        const int wk = st.w[k];
        const int ly = local / wk;
        const int lx = local - ly * wk;
        const int gx = st.x0[k] + lx;
        const int gy = st.y0[k] + ly;

        // tile rectangle in clip space (GaussianRender.cu:656-664)
        const float minX = tileClipMin(gx, scr.tileClipSizeX);
        const float minY = tileClipMin(gy, scr.tileClipSizeY);
        const float maxX = minX + scr.tileClipSizeX;
        const float maxY = minY + scr.tileClipSizeY;
        const EllipseD ek{st.cx[k], st.cy[k], st.ex[k], st.ey[k], st.cosA[k], st.sinA[k]};
        const bool hit = valid && ellipseRectOverlapExact(ek, minX, minY, maxX, maxY); // gs_overlap_exact.h

        const unsigned long long mask = __ballot(hit);
        if (EMIT)
        {
            const int pos = outBase + running + __popcll(mask & lowerLanes);
            if (hit && pos < capacity)
            {
                const int tileIndex = gy * scr.tilesX + gx;
                keys[pos] = ((uint64_t) (uint32_t) tileIndex << 32) | (uint64_t) st.depth[k];
                values[pos] = waveGlobal * kWave + k;
            }
        }
        running += __popcll(mask);
    }

    if (!EMIT && lane == 0) waveCounts[waveGlobal] = running;
}

// Exclusive scan of n ints by ONE workgroup of 1024 lanes (n = ceil(N/64): 15,625 at 1 M splats).
// Each lane owns a contiguous chunk; chunk sums are scanned through LDS; the chunk is re-read (L2) and
// written back as prefixes.  Also writes the grand total as int64.
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void scanWaveCountsKernel(int n, const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                                     int64_t* __restrict__ total)
{
    __shared__ long long partial[kScanThreads];
    const int t = threadIdx.x;
    const int per = (n + kScanThreads - 1) / kScanThreads;
    const int begin = min(t * per, n), end = min(begin + per, n);
    long long sum = 0;
    for (int i = begin; i < end; ++i) sum += in[i];
    partial[t] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials.
    for (int delta = 1; delta < kScanThreads; delta <<= 1)
    {
        long long v = (t >= delta) ? partial[t - delta] : 0;
        __syncthreads();
        partial[t] += v;
        __syncthreads();
    }
    long long prefix = partial[t] - sum;
    for (int i = begin; i < end; ++i)
    {
        // Offsets beyond INT32_MAX only occur past any capacity the int32 API can express; the emit pass
        // drops those pairs (pos >= capacity), so clamping keeps the compare valid.
        out[i] = (int32_t) min(prefix, (long long) 0x7FFFFFFF);
        prefix += in[i];
    }
    if (t == kScanThreads - 1) *total = partial[t];
}

} // namespace

void ensureTileListScratch(TileListScratch& s, int count)
{
    const int waves = (count + kWave - 1) / kWave;
    if (s.total == nullptr)
    {
        GS_CHECK(hipMalloc((void**) &s.total, sizeof(int64_t)));
        GS_CHECK(hipHostMalloc((void**) &s.totalHost, 2 * sizeof(int64_t), hipHostMallocMapped)); // [1]: the pairs a layered frame listed in all
        s.totalHost[0] = s.totalHost[1] = 0;
        if (hipHostGetDevicePointer((void**) &s.totalHostMapped, s.totalHost, 0) != hipSuccess) s.totalHostMapped = nullptr;
    }
    if (waves > s.capacityWaves)
    {
        if (s.waveCounts) GS_CHECK(hipFree(s.waveCounts));
        if (s.waveOffsets) GS_CHECK(hipFree(s.waveOffsets));
        GS_CHECK(hipMalloc((void**) &s.waveCounts, sizeof(int32_t) * (size_t) waves));
        GS_CHECK(hipMalloc((void**) &s.waveOffsets, sizeof(int32_t) * (size_t) waves));
        s.capacityWaves = waves;
    }
}

void freeTileListScratch(TileListScratch& s)
{
    if (s.waveCounts) (void) hipFree(s.waveCounts);
    if (s.waveOffsets) (void) hipFree(s.waveOffsets);
    if (s.hitMasks) (void) hipFree(s.hitMasks);
    if (s.boxRecords) (void) hipFree(s.boxRecords);
    if (s.digitTable) (void) hipFree(s.digitTable);
    if (s.digitScanned) (void) hipFree(s.digitScanned);
    if (s.digitTotals) (void) hipFree(s.digitTotals);
    if (s.total) (void) hipFree(s.total);
    if (s.totalHost) (void) hipHostFree(s.totalHost);
    s = TileListScratch{};
}

void launchTileListCount(hipStream_t stream, int count, const Screen& scr, const float2* clipXY, const float4* screenEllipse,
                         TileListScratch& s)
{
    ensureTileListScratch(s, count);
    const int waves = (count + kWave - 1) / kWave;
    if (waves == 0)
    {
        GS_CHECK(hipMemsetAsync(s.total, 0, sizeof(int64_t), stream));
    }
    else
    {
        const dim3 grid((waves + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
        hipLaunchKernelGGL(tileListKernel<false>, grid, block, 0, stream, count, scr, clipXY, (const float*) nullptr, screenEllipse,
                           s.waveCounts, (const int32_t*) nullptr, (uint64_t*) nullptr, (int32_t*) nullptr, 0);
        hipLaunchKernelGGL(scanWaveCountsKernel, dim3(1), dim3(kScanThreads), 0, stream, waves, s.waveCounts, s.waveOffsets, s.total);
    }
    GS_CHECK(hipMemcpyAsync(s.totalHost, s.total, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
}

void launchTileListEmit(hipStream_t stream, int count, const Screen& scr, const float2* clipXY, const float* clipZ,
                        const float4* screenEllipse, const TileListScratch& s, uint64_t* keys, int32_t* values, int capacity)
{
    const int waves = (count + kWave - 1) / kWave;
    if (waves == 0 || capacity <= 0) return;
    const dim3 grid((waves + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
    hipLaunchKernelGGL(tileListKernel<true>, grid, block, 0, stream, count, scr, clipXY, clipZ, screenEllipse, (int32_t*) nullptr,
                       (const int32_t*) s.waveOffsets, keys, values, capacity);
}

} // namespace gs
