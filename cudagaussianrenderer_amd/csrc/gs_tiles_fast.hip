// gs_tiles_fast.hip -- tile-list builder of the one-call renderer (gs_renderer_render).
//
// Same observable set as buildTileListKernel (GaussianRender.cu:456-770) and as gs_tiles.hip, but built
// for the renderer's two-level sort: splats arrive ALREADY SORTED BY DEPTH (`order` is the permutation
// produced by the 32-bit depth sort), pairs are emitted in that order, and the key is just the tile index
// (16 or 32 bit).  A stable sort on the tile index then yields exactly the list the reference's 64-bit
// (tile|depth) sort yields, at a fraction of the radix-sort traffic ("Two-level sort").
//
// Differences from the bit-exact builder in gs_tiles.hip:
//   * a workgroup (4 waves) owns 256 consecutive sorted splats and walks ALL their work units together, 256 units per
//     round: the nearest splats come first in the depth order and have by far the largest AABBs, so a wave that owned 64
//     of them alone (an earlier version) set the kernel's duration;
//   * work unit = (splat, 4 consecutive AABB tiles): one owner search and one LDS parameter fetch per
//     four overlap tests; per-unit hit masks are compacted with ballots, unit-major, so the emission
//     order is (sorted splat, local tile) with no atomics;
//   * the count pass leaves each lane's four hit bits of its first 16 rounds in memory; the emit pass replays them instead
//     of repeating the overlap tests (later rounds are tested again);
//   * the ellipse/rectangle predicate is evaluated in the ellipse's unit-circle space with the affine map
//     hoisted per splat (reciprocals instead of the reference's 14 divisions per test).  It is the same
//     predicate -- centre-in-rect OR rect-centre-in-ellipse OR an edge within the unit circle
//     (GaussianRender.cu:375-405) -- evaluated in a different floating-point order, so a pair whose
//     overlap margin is at rounding level may differ from the oracle's; tests check exactly that.
#include "gs_aabb.h"
#include "gs_internal.h"

namespace gs
{
namespace
{

constexpr int kWavesPerBlock = 4;
constexpr int kGroup = kWave * kWavesPerBlock; // splats (and lanes) per workgroup
constexpr int kTilesPerUnit = 4;
constexpr int kMaskRounds = 16;                // rounds of a workgroup whose hit masks are kept for the emit pass
// The emit pass sums the group totals in front of each group itself up to this many groups (1 M splats: 3 906); longer lists get
// the scan kernel.  A band-culled list (device-side count) keeps only a fraction of the groups the grid was sized for, so the
// limit on the host-side group count is wider there.
constexpr int kSelfSumGroups = 4096;
constexpr int kSelfSumGroupsCulled = 16384;

struct GroupStage
{
    int excl[kGroup];   // exclusive prefix of units
    int area[kGroup];   // AABB tiles
    int origin[kGroup]; // x0 | y0 << 16
    int w[kGroup];
    float rcpW[kGroup];
    float4 centreM[kGroup]; // cx, cy, m00, m01
    float4 mInv[kGroup];    // m10, m11, 1/|u|^2, 1/|v|^2
    int splat[kGroup];
};

// Squared distance of the origin to the segment q + t e, t in [0,1]: the edge comes within the unit circle when it is
// below 1 (GaussianRender.cu:361-372).
__device__ __forceinline__ float edgeDistanceSquared(float qx, float qy, float ex, float ey, float qe, float invEE)
{
    const float t = __builtin_amdgcn_fmed3f(-qe * invEE, 0.0f, 1.0f); // saturate in one instruction (v_med3_f32)
    const float rx = fmaf(t, ex, qx), ry = fmaf(t, ey, qy);
    return fmaf(rx, rx, ry * ry);
}

template <bool EMIT, typename TileKey>
__global__ __launch_bounds__(kGroup) void tileListFastKernel(int countHost, const int* __restrict__ countDevice, Screen scr, const uint32_t* __restrict__ order,
                                                             const float2* __restrict__ clipXY, const float4* __restrict__ screenEllipse,
                                                             int32_t* __restrict__ groupCounts, const int32_t* __restrict__ groupOffsets,
                                                             TileKey* __restrict__ tileKeys, int32_t* __restrict__ values, int capacity,
                                                             uint8_t* __restrict__ hitMasks, const float4* __restrict__ tileRecord,
                                                             int64_t* __restrict__ total, int64_t* __restrict__ totalHostMapped, int viewSplats)
{
    __shared__ GroupStage st;
    __shared__ int waveTotals[kWavesPerBlock];
    __shared__ long long scanTotals[kWavesPerBlock];
    __shared__ int roundHits[2][kWavesPerBlock];
    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & (kWave - 1);
    const int group = blockIdx.x;
    const int count = countDevice ? min(*countDevice, countHost) : countHost; // device-side count: culled splat list
    if (EMIT && count <= 0 && group == 0 && t == 0)
    {
        // nothing reached the band: no workgroup gets as far as the pair total below
        *total = 0;
        if (totalHostMapped != nullptr) *totalHostMapped = 0;
    }
    if (group * kGroup >= count) return; // whole workgroup
    const int sorted = group * kGroup + t;

    int area = 0, x0 = 0, y0 = 0, w = 0, splat = 0;
    float cx = 0.f, cy = 0.f, m00 = 0.f, m01 = 0.f, m10 = 0.f, m11 = 0.f, invUU = 0.f, invVV = 0.f;
    if (sorted < count)
    {
        splat = (int) order[sorted];
        float2 c;
        float4 se; // cos, sin, extent.x, extent.y
        if (tileRecord != nullptr)
        {
            const float4 ra = tileRecord[2 * splat], rb = tileRecord[2 * splat + 1]; // one 32-byte sector
            c = make_float2(ra.x, ra.y);
            se = make_float4(ra.z, ra.w, rb.x, rb.y);
        }
        else
        {
            c = clipXY[splat];
            se = screenEllipse[splat];
        }
        cx = c.x;
        cy = c.y;
        // AABB in tiles, exactly as gs_tiles.hip / GaussianRender.cu:408-436,537-551 (shared with the band cull).
        const TileAabb box = splatTileAabb(cx, cy, se.x, se.y, se.z, se.w, scr);
        x0 = box.x0;
        y0 = box.y0;
        w = box.x1 - box.x0;
        const int h = box.y1 - box.y0;
        area = (w > 0 && h > 0) ? w * h : 0;
        // Affine map clip space -> the ellipse's unit-circle space (GaussianRender.cu:351-358), hoisted:
        //   l = M (p - c),  M = diag(1/ex, 1/ey) * [[cos, sin], [-sin, cos]]
        const float invEx = 1.0f / se.z, invEy = 1.0f / se.w;
        m00 = se.x * invEx;
        m01 = se.y * invEx;
        m10 = -se.y * invEy;
        m11 = se.x * invEy;
        // Images of the tile edges: u = M (sx, 0), v = M (0, sy).
        const float ux = m00 * scr.tileClipSizeX, uy = m10 * scr.tileClipSizeX;
        const float vx = m01 * scr.tileClipSizeY, vy = m11 * scr.tileClipSizeY;
        invUU = 1.0f / fmaf(ux, ux, uy * uy);
        invVV = 1.0f / fmaf(vx, vx, vy * vy);
    }
    const int units = (area + kTilesPerUnit - 1) / kTilesPerUnit;

    // exclusive prefix of the units over the workgroup's 256 splats
    int incl = units;
#pragma unroll
    for (int delta = 1; delta < kWave; delta <<= 1)
    {
        const int n = __shfl_up(incl, delta);
        if (lane >= delta) incl += n;
    }
    if (lane == kWave - 1) waveTotals[wave] = incl;
    __syncthreads();
    int unitsBefore = 0, totalUnits = 0;
#pragma unroll
    for (int wv = 0; wv != kWavesPerBlock; ++wv)
    {
        const int v = waveTotals[wv];
        unitsBefore += (wv < wave) ? v : 0;
        totalUnits += v;
    }
    const int lastGroup = (count + kGroup - 1) / kGroup - 1;
    if (totalUnits == 0 && !(EMIT && group == lastGroup)) // nothing on screen (e.g. the culled splats, which sort first): uniform exit
    {
        if (!EMIT && t == 0) groupCounts[group] = 0;
        return;
    }
    // EMIT: where this workgroup's pairs start = the pairs of all groups before it.  Up to kSelfSumGroups groups every workgroup
    // sums the count pass's group totals in front of it itself (a few thousand L2-resident words, 16 per lane) instead of reading the
    // result of a one-workgroup scan kernel: one kernel and one dependent launch less per frame; the last group also leaves the pair
    // total.  Beyond that (5 M splats: 19.5 K groups, 190 M word loads) the self-summing would be quadratic work: the host launches
    // the scan kernel in between and the offsets (and the total) come from there.
    int outBase = 0;
    if (EMIT && groupOffsets != nullptr)
    {
        outBase = groupOffsets[group];
        if (totalUnits == 0) return;
    }
    else if (EMIT)
    {
        long long mine = 0;
        for (int j = t; j < group; j += kGroup) mine += groupCounts[j];
#pragma unroll
        for (int delta = kWave / 2; delta != 0; delta >>= 1) mine += __shfl_xor(mine, delta);
        __syncthreads(); // waveTotals: everybody is past the prologue's reads
        if (lane == 0) scanTotals[wave] = mine;
        __syncthreads();
        const long long before = scanTotals[0] + scanTotals[1] + scanTotals[2] + scanTotals[3];
        outBase = (int) min(before, (long long) 0x7FFFFFFF);
        if (group == lastGroup && t == 0)
        {
            const long long all = before + groupCounts[group];
            *total = all;
            // pinned, device-mapped host word: the host reads it while the rest of the frame runs (no copy kernel)
            if (totalHostMapped != nullptr) *totalHostMapped = all;
        }
        if (totalUnits == 0) return;
    }

    st.excl[t] = unitsBefore + incl - units;
    st.area[t] = area;
    st.origin[t] = x0 | (y0 << 16);
    st.w[t] = max(w, 1);
    st.rcpW[t] = 1.0f / (float) max(w, 1);
    st.centreM[t] = make_float4(cx, cy, m00, m01);
    st.mInv[t] = make_float4(m10, m11, invUU, invVV);
    st.splat[t] = splat;
    __syncthreads();

    int running = 0; // COUNT: hits of this wave so far; EMIT: hits of the whole workgroup in the rounds before this one
    const unsigned long long lowerLanes = (lane == 0) ? 0ull : (~0ull >> (kWave - lane));
    const float sx = scr.tileClipSizeX, sy = scr.tileClipSizeY;

    for (int base = 0, round = 0; base < totalUnits; base += kGroup, ++round)
    {
        const int unit = base + t;
        const bool valid = unit < totalUnits;
        int k = 0;
#pragma unroll
        for (int step = kGroup / 2; step != 0; step >>= 1)
        {
            const int probe = k + step;
            if (st.excl[probe] <= unit) k = probe;
        }
        const int local0 = (unit - st.excl[k]) * kTilesPerUnit;
        const int areaK = valid ? st.area[k] : 0;
        const int wk = st.w[k];
        const int origin = st.origin[k];
        // local0 / wk through the reciprocal: the quotient's fractional part is at least 0.5/wk away from an
        // integer, far above the float error for local0 < 2^16.
        const int ly0 = (int) (((float) local0 + 0.5f) * st.rcpW[k]);
        const int lx0 = local0 - ly0 * wk;

        // The count pass leaves the four hit bits of every lane for its first kMaskRounds rounds; the emit pass replays
        // them instead of repeating the overlap tests (rounds past that are simply tested again). Uniform branches.
        const bool recorded = round < kMaskRounds;
        uint8_t* maskSlot = hitMasks + ((size_t) group * kMaskRounds + (size_t) (recorded ? round : 0)) * kGroup + t;
        unsigned hitMask = 0;
        if (EMIT && recorded) hitMask = *maskSlot;
        else
        {
            const float4 cm = st.centreM[k];
            const float4 mi = st.mInv[k];
            const float ux = cm.z * sx, uy = mi.x * sx; // u = M (sx, 0)
            const float vx = cm.w * sy, vy = mi.y * sy; // v = M (0, sy)
            const float uv = fmaf(ux, vx, uy * vy);
            int lx = lx0, ly = ly0;
#pragma unroll
            for (int e = 0; e != kTilesPerUnit; ++e)
            {
                bool hit = false;
                if (local0 + e < areaK)
                {
                    const int gx = (origin & 0xFFFF) + lx, gy = (origin >> 16) + ly;
                    const float minX = (float) gx * sx - 1.0f, minY = (float) gy * sy - 1.0f; // GaussianRender.cu:656-664
                    // centre inside the rectangle (strict, GaussianRender.cu:382-386)
                    hit = cm.x > minX && cm.x < minX + sx && cm.y > minY && cm.y < minY + sy;
                    // corner q0 = M (min - c); the other corners are q0 + u, q0 + v, q0 + u + v
                    const float dx = minX - cm.x, dy = minY - cm.y;
                    const float q0x = fmaf(cm.z, dx, cm.w * dy), q0y = fmaf(mi.x, dx, mi.y * dy);
                    const float qcx = fmaf(0.5f, ux + vx, q0x), qcy = fmaf(0.5f, uy + vy, q0y);
                    // rectangle centre inside the ellipse (:389), or an edge within the unit circle (:392-402): the smallest of
                    // the five squared distances decides (a NaN distance never wins a min, as it never passed the comparison)
                    const float centre2 = fmaf(qcx, qcx, qcy * qcy);
                    const float q0u = fmaf(q0x, ux, q0y * uy), q0v = fmaf(q0x, vx, q0y * vy);
                    const float bottom2 = edgeDistanceSquared(q0x, q0y, ux, uy, q0u, mi.z);
                    const float right2 = edgeDistanceSquared(q0x + ux, q0y + uy, vx, vy, q0v + uv, mi.w);
                    const float top2 = edgeDistanceSquared(q0x + vx, q0y + vy, ux, uy, q0u + uv, mi.z);
                    const float left2 = edgeDistanceSquared(q0x, q0y, vx, vy, q0v, mi.w);
                    hit |= fminf(fminf(fminf(bottom2, right2), fminf(top2, left2)), centre2) < 1.0f;
                }
                hitMask |= hit ? (1u << e) : 0u;
                // next local tile
                ++lx;
                if (lx == wk)
                {
                    lx = 0;
                    ++ly;
                }
            }
            if (!EMIT && recorded) *maskSlot = (uint8_t) hitMask;
        }
        unsigned long long ballots[kTilesPerUnit];
        int waveHits = 0;
#pragma unroll
        for (int e = 0; e != kTilesPerUnit; ++e)
        {
            ballots[e] = __ballot((hitMask & (1u << e)) != 0u);
            waveHits += __popcll(ballots[e]);
        }

        if (!EMIT) running += waveHits;
        else
        {
            // this round's hits of the waves before this one (order = unit order: round, wave, lane, tile)
            if (lane == 0) roundHits[round & 1][wave] = waveHits;
            __syncthreads(); // one barrier per round: the other half of roundHits belongs to the next round
            int before = 0, all = 0;
#pragma unroll
            for (int wv = 0; wv != kWavesPerBlock; ++wv)
            {
                const int v = roundHits[round & 1][wv];
                before += (wv < wave) ? v : 0;
                all += v;
            }
            // (unsigned: a demand beyond 2^31 pairs wraps past every capacity instead of into negative positions)
            unsigned pos = (unsigned) outBase + (unsigned) running + (unsigned) before;
#pragma unroll
            for (int e = 0; e != kTilesPerUnit; ++e) pos += (unsigned) __popcll(ballots[e] & lowerLanes);
            // re-walk the unit's tiles to write the hits, unit-major
            int wx = lx0, wy = ly0;
            const int splatK = st.splat[k];
            // several views per chain (gs_renderer_set_views): virtual splat v*N + i belongs to view v, whose tiles are the virtual
            // tiles v*totalTiles + t -- the tile sort then separates the views like it separates tiles
            int viewTileBase = 0;
            if (viewSplats > 0)
#pragma unroll
                for (int v = 1; v != kMaxViews; ++v) viewTileBase += (splatK >= v * viewSplats) ? scr.totalTiles : 0;
#pragma unroll
            for (int e = 0; e != kTilesPerUnit; ++e)
            {
                if (hitMask & (1u << e))
                {
                    if (pos < (unsigned) capacity)
                    {
                        const int gx = (origin & 0xFFFF) + wx, gy = (origin >> 16) + wy;
                        tileKeys[pos] = (TileKey) (gy * scr.tilesX + gx + viewTileBase);
                        values[pos] = splatK;
                    }
                    ++pos;
                }
                ++wx;
                if (wx == wk)
                {
                    wx = 0;
                    ++wy;
                }
            }
            running += all;
        }
    }

    if (!EMIT)
    {
        __syncthreads(); // waveTotals: everybody is past the prologue's reads
        if (lane == 0) waveTotals[wave] = running;
        __syncthreads();
        if (t == 0) groupCounts[group] = waveTotals[0] + waveTotals[1] + waveTotals[2] + waveTotals[3];
    }
}

// Exclusive scan of n ints by ONE workgroup (n = ceil(N/64)): 1024 lanes x 16 consecutive ints per pass (four 16-byte
// loads in flight per lane), one wave scan of the lane sums by shuffles, 16 wave totals through LDS, carry across passes
// (one pass up to 16 Ki entries = 1 M splats).
constexpr int kScanThreads = 1024;
constexpr int kScanItems = 16;
__global__ __launch_bounds__(kScanThreads) void scanFastKernel(int nHost, const int* __restrict__ countDevice, int groupSize, const int32_t* __restrict__ in,
                                                               int32_t* __restrict__ out, int64_t* __restrict__ total, int64_t* __restrict__ totalHostMapped)
{
    __shared__ long long waveTotals[2][kScanThreads / kWave];
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
    // nHost = groups (of groupSize splats) the grid of the count pass was sized for; with a device-side splat count only
    // the first ceil(count/groupSize) of them wrote their totals.
    const int n = countDevice ? min((*countDevice + groupSize - 1) / groupSize, nHost) : nHost;
    long long carry = 0;
    int flip = 0;
    for (int base = 0; base < n; base += kScanThreads * kScanItems, flip ^= 1)
    {
        const int i = base + t * kScanItems;
        int v[kScanItems];
        if (i + kScanItems <= n)
        {
#pragma unroll
            for (int q = 0; q != kScanItems / 4; ++q)
            {
                const int4 w = *reinterpret_cast<const int4*>(in + i + 4 * q);
                v[4 * q] = w.x;
                v[4 * q + 1] = w.y;
                v[4 * q + 2] = w.z;
                v[4 * q + 3] = w.w;
            }
        }
        else
        {
#pragma unroll
            for (int k = 0; k != kScanItems; ++k) v[k] = (i + k < n) ? in[i + k] : 0;
        }
        long long own = 0;
#pragma unroll
        for (int k = 0; k != kScanItems; ++k) own += v[k];
        long long incl = own;
#pragma unroll
        for (int delta = 1; delta < kWave; delta <<= 1)
        {
            const long long o = __shfl_up(incl, delta);
            if (lane >= delta) incl += o;
        }
        if (lane == kWave - 1) waveTotals[flip][wave] = incl;
        __syncthreads(); // the other half of waveTotals is free for the next pass: one barrier per pass
        long long before = carry, all = 0;
#pragma unroll
        for (int wv = 0; wv != kScanThreads / kWave; ++wv)
        {
            const long long ws = waveTotals[flip][wv];
            before += (wv < wave) ? ws : 0;
            all += ws;
        }
        long long p = before + incl - own;
        const long long cap = 0x7FFFFFFF;
        if (i + kScanItems <= n)
        {
#pragma unroll
            for (int q = 0; q != kScanItems / 4; ++q)
            {
                int4 w;
                w.x = (int32_t) min(p, cap);
                p += v[4 * q];
                w.y = (int32_t) min(p, cap);
                p += v[4 * q + 1];
                w.z = (int32_t) min(p, cap);
                p += v[4 * q + 2];
                w.w = (int32_t) min(p, cap);
                p += v[4 * q + 3];
                *reinterpret_cast<int4*>(out + i + 4 * q) = w;
            }
        }
        else
        {
#pragma unroll
            for (int k = 0; k != kScanItems; ++k)
            {
                if (i + k < n) out[i + k] = (int32_t) min(p, cap);
                p += v[k];
            }
        }
        carry += all;
    }
    if (t == 0)
    {
        *total = carry;
        // pinned, device-mapped host word: the host reads it after the event that follows this kernel (no copy kernel)
        if (totalHostMapped != nullptr) *totalHostMapped = carry;
    }
}

template <typename TileKey>
void launchEmit(hipStream_t stream, dim3 grid, dim3 block, int count, const Screen& scr, const uint32_t* order, const float2* clipXY,
                const float4* screenEllipse, const TileListScratch& s, const int32_t* groupOffsets, void* tileKeys, int32_t* values, int capacity,
                int viewSplats)
{
    hipLaunchKernelGGL((tileListFastKernel<true, TileKey>), grid, block, 0, stream, count, s.countDevice, scr, order, clipXY, screenEllipse,
                       s.waveCounts, groupOffsets, (TileKey*) tileKeys, values, capacity, s.hitMasks, s.tileRecord, s.total,
                       s.totalHostMapped, viewSplats);
}

} // namespace

void launchScanInt32(hipStream_t stream, int n, const int32_t* in, int32_t* out, int64_t* total)
{
    hipLaunchKernelGGL(scanFastKernel, dim3(1), dim3(kScanThreads), 0, stream, n, (const int*) nullptr, kWave, in, out, total, (int64_t*) nullptr);
}

void ensureTileListFastScratch(TileListScratch& s, int count)
{
    ensureTileListScratch(s, count);
    const int groups = (count + kGroup - 1) / kGroup;
    if (groups > s.capacityMaskGroups)
    {
        if (s.hitMasks) GS_CHECK(hipFree(s.hitMasks));
        GS_CHECK(hipMalloc((void**) &s.hitMasks, (size_t) groups * kMaskRounds * kGroup));
        s.capacityMaskGroups = groups;
    }
}

void launchTileListFastCount(hipStream_t stream, int count, const Screen& scr, const uint32_t* order, const float2* clipXY,
                             const float4* screenEllipse, TileListScratch& s)
{
    ensureTileListFastScratch(s, count);
    const int groups = (count + kGroup - 1) / kGroup;
    if (groups == 0)
    {
        GS_CHECK(hipMemsetAsync(s.total, 0, sizeof(int64_t), stream));
    }
    else
    {
        // per-group totals and their scan live in the first ceil(N/256) entries of the per-wave arrays
        // per-group totals live in the first ceil(N/256) entries of the per-wave array; the emit pass sums them itself and
        // leaves the pair total (no scan kernel in between)
        hipLaunchKernelGGL((tileListFastKernel<false, uint16_t>), dim3(groups), dim3(kGroup), 0, stream, count, s.countDevice, scr, order, clipXY,
                           screenEllipse, s.waveCounts, (const int32_t*) nullptr, (uint16_t*) nullptr, (int32_t*) nullptr, 0, s.hitMasks, s.tileRecord,
                           (int64_t*) nullptr, (int64_t*) nullptr, 0);
    }
    if (groups == 0) GS_CHECK(hipMemcpyAsync(s.totalHost, s.total, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
}

void launchTileListFastEmit(hipStream_t stream, int count, const Screen& scr, const uint32_t* order, const float2* clipXY,
                            const float4* screenEllipse, const TileListScratch& s, void* tileKeys, int tileKeyBytes, int32_t* values,
                            int capacity, int viewSplats)
{
    const int groups = (count + kGroup - 1) / kGroup;
    if (groups == 0) return;
    // (capacity 0 still runs: the pass is also what produces the pair total; nothing is written past `capacity`)
    const dim3 grid(groups), block(kGroup);
    const int32_t* groupOffsets = nullptr;
    if (groups > (s.countDevice != nullptr ? kSelfSumGroupsCulled : kSelfSumGroups))
    {
        // long lists: exclusive scan of the group totals (and the pair total) by one workgroup, linear work
        hipLaunchKernelGGL(scanFastKernel, dim3(1), dim3(kScanThreads), 0, stream, groups, s.countDevice, kGroup, s.waveCounts, s.waveOffsets, s.total,
                           s.totalHostMapped);
        groupOffsets = s.waveOffsets;
    }
    if (tileKeyBytes == 2) launchEmit<uint16_t>(stream, grid, block, count, scr, order, clipXY, screenEllipse, s, groupOffsets, tileKeys, values, capacity, viewSplats);
    else launchEmit<uint32_t>(stream, grid, block, count, scr, order, clipXY, screenEllipse, s, groupOffsets, tileKeys, values, capacity, viewSplats);
    if (s.totalHostMapped == nullptr) GS_CHECK(hipMemcpyAsync(s.totalHost, s.total, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
}

} // namespace gs
