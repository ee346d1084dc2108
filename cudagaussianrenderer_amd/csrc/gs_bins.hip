// gs_bins.hip -- the renderer's "binned" tile lists: depth-ordered splats go STRAIGHT to their final slot of the
// per-tile lists, with no pair sort and no range search.
//
// Observable result = the reference's (GaussianRender.cu:456-906: buildTileList + 64-bit sort + evaluateTileRange):
// for every tile the indices of the splats that overlap it, ordered by (quantised depth, splat index), stored
// back to back in tile order, plus the [begin, end) range of every tile.  How it gets there:
//
//   the splats arrive sorted by depth (`order`, from the 32-bit depth sort) and are cut into ROWS of 64*batches
//   consecutive sorted splats; one wave owns one row.
//   1. count  -- the wave walks its row exactly like gs_tiles_fast.hip (same AABB, same overlap predicate, work unit =
//                4 AABB tiles) and counts its hits per tile in LDS (16-bit counters, two per word) -> counts[row][tile].
//   2. scan   -- per tile column: exclusive scan over the rows -> offsets[row][tile]; column total -> tileTotals[tile].
//   3. starts -- exclusive scan of the tile totals (one workgroup) -> tileStart[tile], the range table, the pair total.
//   4. emit   -- every wave seeds one LDS cursor per tile with tileStart + offsets[row][tile], walks its row again and
//                writes each hit's splat index at the cursor (LDS fetch-add).
//
// Order inside a tile: rows are in depth order and so are the splats inside a row, so all that is needed is that two
// hits of one wave on the same tile take their cursor values in splat order.  Lanes that work on the SAME splat never
// share a tile; lanes of DIFFERENT splats issue their fetch-adds in separate instructions, earlier splat first, and LDS
// instructions of one wave execute in program order -- nothing depends on how the LDS serialises lanes inside one
// instruction.  (Tried and dropped: one cursor table per 8-wave workgroup with unordered fetch-adds and a per-cell
// order fix-up afterwards -- count and emit get faster (56 + 88 us at C3) but the scene's hot tiles put ~10^5 cells above
// eight entries and the fix-up passes cost more than the ordering they replace.)
//
// HBM traffic per frame (1080p, P pairs, R non-empty rows, T tiles): counts 2 B x R x T written + read twice,
// offsets 4 B x R x T written + read, values 4 B x P written: ~75 MB at C3 against ~330 MB for emit + two radix passes
// over the pairs + the range search of the sort-based path.
#include "gs_aabb.h"
#include "gs_internal.h"

namespace gs
{
namespace
{

constexpr int kTilesPerUnit = 4;
constexpr int kColumnPad = 256;    // row stride of the tables in tiles (multiple of 4 lanes x 64)
constexpr int kScanSegments = 16;  // waves per workgroup of the column scan

struct WaveStage
{
    int excl[kWave];   // exclusive prefix of units
    int area[kWave];   // AABB tiles
    int origin[kWave]; // x0 | y0 << 16
    int w[kWave];
    float rcpW[kWave];
    float4 centreM[kWave]; // cx, cy, m00, m01
    float4 mInv[kWave];    // m10, m11, 1/|u|^2, 1/|v|^2
    int splat[kWave];
};
static_assert(sizeof(WaveStage) % 16 == 0, "cursor table must stay 16-byte aligned");

struct BinArgs
{
    int countHost;
    const int* countDevice;
    Screen scr;
    int batches;   // 64-splat batches per row
    int bandTiles; // tiles of the band (columns in use)
    int stride;    // row stride of counts/offsets in tiles
    const uint32_t* order;
    const float2* clipXY;
    const float4* screenEllipse;
    const float4* tileRecord; // optional: (x, y, cos, sin | extent x, y, 0, 0), one sector per splat
    uint16_t* counts;     // [rows][stride]
    uint32_t* offsets;    // [rows][stride]
    int32_t* rowFlags;    // [rows] 1 = the row has hits
    int32_t* tileTotals;  // [stride]
    uint32_t* tileStart;  // [stride] exclusive scan of tileTotals
    int32_t* values;      // final per-tile lists, back to back
    int capacity;
    int32_t* tileRange;   // [totalTiles][2], global tile index
    int64_t* total;
    int64_t* totalHostMapped;
};

// Squared distance of the origin to the segment q + t e, t in [0,1]: the edge comes within the unit circle when it is
// below 1 (GaussianRender.cu:361-372).
__device__ __forceinline__ float edgeDistanceSquared(float qx, float qy, float ex, float ey, float qe, float invEE)
{
    const float t = __builtin_amdgcn_fmed3f(-qe * invEE, 0.0f, 1.0f); // saturate in one instruction (v_med3_f32)
    const float rx = fmaf(t, ex, qx), ry = fmaf(t, ey, qy);
    return fmaf(rx, rx, ry * ry);
}

// One row of sorted splats -> per-tile hit counts (EMIT = false) or list entries (EMIT = true).
// Same walk and the same predicate arithmetic as tileListFastKernel (gs_tiles_fast.hip): both pipelines agree on
// every hit bit for bit.
template <bool EMIT>
__global__ __launch_bounds__(kWave) void binRowsKernel(BinArgs a)
{
    extern __shared__ __align__(16) unsigned char ldsRaw[];
    WaveStage& st = *reinterpret_cast<WaveStage*>(ldsRaw);
    unsigned* spare = reinterpret_cast<unsigned*>(ldsRaw + sizeof(WaveStage)); // one private word per lane
    unsigned* cursor = spare + kWave;

    const Screen& scr = a.scr;
    const int lane = threadIdx.x;
    const int row = blockIdx.x;
    const int count = a.countDevice ? min(*a.countDevice, a.countHost) : a.countHost;
    const int rowSplats = a.batches * kWave;
    const int rows = (count + rowSplats - 1) / rowSplats;
    const int first = row * rowSplats;

    if (!EMIT)
    {
        if (row >= rows) return;
        const int words = a.stride / 2;
        for (int i = lane * 4; i < words; i += kWave * 4) *reinterpret_cast<uint4*>(cursor + i) = make_uint4(0u, 0u, 0u, 0u);
    }
    else
    {
        if (row >= rows || a.rowFlags[row] == 0) return;
        // cursor = tile start + this row's offset inside the tile's list (both from the scan kernels)
        const uint4* startRow = reinterpret_cast<const uint4*>(a.tileStart);
        const uint4* offsetRow = reinterpret_cast<const uint4*>(a.offsets + (size_t) row * a.stride);
        const int quads = a.stride / 4; // multiple of 64
#pragma unroll 4
        for (int q = lane; q < quads; q += kWave)
        {
            const uint4 s4 = startRow[q];
            const uint4 o4 = offsetRow[q];
            *reinterpret_cast<uint4*>(cursor + 4 * q) = make_uint4(s4.x + o4.x, s4.y + o4.y, s4.z + o4.z, s4.w + o4.w);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const float sx = scr.tileClipSizeX, sy = scr.tileClipSizeY;
    int anyHits = 0;

    // Splat data travels two steps ahead of its use: the index of batch b+2 and the clip data of batch b+1 are
    // requested before batch b is walked, so one wave per SIMD is enough to keep the loads off the critical path.
    const int lastSorted = count - 1;
    auto loadIndex = [&](int batch) -> int {
        const int sorted = first + batch * kWave + lane;
        return (batch < a.batches && sorted <= lastSorted) ? (int) a.order[sorted] : -1;
    };
    int splatNext = loadIndex(0);
    int splatNext2 = loadIndex(1);
    float2 cNext = make_float2(0.f, 0.f);
    float4 seNext = make_float4(1.f, 0.f, 0.f, 0.f);
    auto loadSplat = [&](int splat, float2& c, float4& se) {
        if (a.tileRecord != nullptr)
        {
            const float4 ra = a.tileRecord[2 * splat], rb = a.tileRecord[2 * splat + 1];
            c = make_float2(ra.x, ra.y);
            se = make_float4(ra.z, ra.w, rb.x, rb.y);
        }
        else
        {
            c = a.clipXY[splat];
            se = a.screenEllipse[splat];
        }
    };
    if (splatNext >= 0) loadSplat(splatNext, cNext, seNext);

    for (int batch = 0; batch < a.batches; ++batch)
    {
        const int batchFirst = first + batch * kWave;
        if (batchFirst >= count) break;
        const int splatNow = splatNext;
        const float2 c = cNext;
        const float4 se = seNext; // cos, sin, extent.x, extent.y
        splatNext = splatNext2;
        splatNext2 = loadIndex(batch + 2);
        if (splatNext >= 0) loadSplat(splatNext, cNext, seNext);

        int area = 0, x0 = 0, y0 = 0, w = 0, splat = 0;
        float cx = 0.f, cy = 0.f, m00 = 0.f, m01 = 0.f, m10 = 0.f, m11 = 0.f, invUU = 0.f, invVV = 0.f;
        if (splatNow >= 0)
        {
            splat = splatNow;
            cx = c.x;
            cy = c.y;
            const TileAabb box = splatTileAabb(cx, cy, se.x, se.y, se.z, se.w, scr);
            x0 = box.x0;
            y0 = box.y0;
            w = box.x1 - box.x0;
            const int h = box.y1 - box.y0;
            area = (w > 0 && h > 0) ? w * h : 0;
            const float invEx = 1.0f / se.z, invEy = 1.0f / se.w;
            m00 = se.x * invEx;
            m01 = se.y * invEx;
            m10 = -se.y * invEy;
            m11 = se.x * invEy;
            const float ux = m00 * scr.tileClipSizeX, uy = m10 * scr.tileClipSizeX;
            const float vx = m01 * scr.tileClipSizeY, vy = m11 * scr.tileClipSizeY;
            invUU = 1.0f / fmaf(ux, ux, uy * uy);
            invVV = 1.0f / fmaf(vx, vx, vy * vy);
        }
        const int units = (area + kTilesPerUnit - 1) / kTilesPerUnit;
        int incl = units;
#pragma unroll
        for (int delta = 1; delta < kWave; delta <<= 1)
        {
            const int n = __shfl_up(incl, delta);
            if (lane >= delta) incl += n;
        }
        const int totalUnits = __builtin_amdgcn_readfirstlane(__shfl(incl, kWave - 1));
        if (totalUnits == 0) continue;

        // the previous batch's readers are done with the stage (same wave, program order) -- refill it
        st.excl[lane] = incl - units;
        st.area[lane] = area;
        st.origin[lane] = x0 | (y0 << 16);
        st.w[lane] = max(w, 1);
        st.rcpW[lane] = 1.0f / (float) max(w, 1);
        st.centreM[lane] = make_float4(cx, cy, m00, m01);
        st.mInv[lane] = make_float4(m10, m11, invUU, invVV);
        st.splat[lane] = splat;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        for (int base = 0; base < totalUnits; base += kWave)
        {
            const int unit = base + lane;
            const bool valid = unit < totalUnits;
            int k = 0;
#pragma unroll
            for (int step = kWave / 2; step != 0; step >>= 1)
            {
                const int t = k + step;
                if (st.excl[t] <= unit) k = t;
            }
            const int local0 = (unit - st.excl[k]) * kTilesPerUnit;
            const int areaK = valid ? st.area[k] : 0;
            const int wk = st.w[k];
            const int origin = st.origin[k];
            const float4 cm = st.centreM[k];
            const float4 mi = st.mInv[k];
            int ly = (int) (((float) local0 + 0.5f) * st.rcpW[k]);
            int lx = local0 - ly * wk;

            const float ux = cm.z * sx, uy = mi.x * sx; // u = M (sx, 0)
            const float vx = cm.w * sy, vy = mi.y * sy; // v = M (0, sy)
            const float uv = fmaf(ux, vx, uy * vy);

            unsigned hitMask = 0;
            int tileOf[kTilesPerUnit];
#pragma unroll
            for (int t = 0; t != kTilesPerUnit; ++t)
            {
                bool hit = false;
                const int gx = (origin & 0xFFFF) + lx, gy = (origin >> 16) + ly;
                tileOf[t] = (gy - scr.rowBegin) * scr.tilesX + gx; // band-local tile
                if (local0 + t < areaK)
                {
                    const float minX = (float) gx * sx - 1.0f, minY = (float) gy * sy - 1.0f; // GaussianRender.cu:656-664
                    hit = cm.x > minX && cm.x < minX + sx && cm.y > minY && cm.y < minY + sy;
                    const float dx = minX - cm.x, dy = minY - cm.y;
                    const float q0x = fmaf(cm.z, dx, cm.w * dy), q0y = fmaf(mi.x, dx, mi.y * dy);
                    const float qcx = fmaf(0.5f, ux + vx, q0x), qcy = fmaf(0.5f, uy + vy, q0y);
                    const float centre2 = fmaf(qcx, qcx, qcy * qcy);
                    const float q0u = fmaf(q0x, ux, q0y * uy), q0v = fmaf(q0x, vx, q0y * vy);
                    const float bottom2 = edgeDistanceSquared(q0x, q0y, ux, uy, q0u, mi.z);
                    const float right2 = edgeDistanceSquared(q0x + ux, q0y + uy, vx, vy, q0v + uv, mi.w);
                    const float top2 = edgeDistanceSquared(q0x + vx, q0y + vy, ux, uy, q0u + uv, mi.z);
                    const float left2 = edgeDistanceSquared(q0x, q0y, vx, vy, q0v, mi.w);
                    hit |= fminf(fminf(fminf(bottom2, right2), fminf(top2, left2)), centre2) < 1.0f;
                }
                hitMask |= hit ? (1u << t) : 0u;
                ++lx;
                if (lx == wk)
                {
                    lx = 0;
                    ++ly;
                }
            }
            const unsigned long long hitLanes = __ballot(hitMask != 0u);
            if (hitLanes == 0ull) continue;
            anyHits = 1;

            if (!EMIT)
            {
                // order does not matter for counting: one LDS add per hit, 16-bit counters packed two per word
#pragma unroll
                for (int t = 0; t != kTilesPerUnit; ++t)
                    if (hitMask & (1u << t)) atomicAdd(&cursor[tileOf[t] >> 1], (tileOf[t] & 1) ? 0x10000u : 1u);
            }
            else
            {
                // Lanes are ordered by splat (k never decreases with the lane).  One set of fetch-adds per splat, earlier
                // splat first: two splats that hit the same tile in this round take their slots in depth order.  Lanes
                // outside the current splat add 0 to a private word, so every set is four full-wave LDS instructions with
                // no branch; four sets are in flight before their results are awaited.
                const int kHit = (hitMask != 0u) ? k : -1;
                unsigned long long pending = hitLanes;
                unsigned pos[kTilesPerUnit] = {0u, 0u, 0u, 0u};
                constexpr int kSets = 4;
                while (pending != 0ull)
                {
                    unsigned got[kSets][kTilesPerUnit];
                    bool mine[kSets];
#pragma unroll
                    for (int j = 0; j != kSets; ++j)
                    {
                        const int leader = (pending != 0ull) ? __ffsll((long long) pending) - 1 : 0;
                        const int kk = (pending != 0ull) ? __builtin_amdgcn_readlane(kHit, leader) : -2;
                        mine[j] = kHit == kk;
                        pending &= ~__ballot(mine[j]);
#pragma unroll
                        for (int t = 0; t != kTilesPerUnit; ++t)
                        {
                            const bool take = mine[j] && (hitMask & (1u << t)) != 0u;
                            got[j][t] = atomicAdd(take ? &cursor[tileOf[t]] : &spare[lane], take ? 1u : 0u);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // keep the sets in program order
                    }
#pragma unroll
                    for (int j = 0; j != kSets; ++j)
#pragma unroll
                        for (int t = 0; t != kTilesPerUnit; ++t) pos[t] = mine[j] ? got[j][t] : pos[t];
                }
                const int splatK = st.splat[k];
#pragma unroll
                for (int t = 0; t != kTilesPerUnit; ++t)
                    if ((hitMask & (1u << t)) && pos[t] < (unsigned) a.capacity) a.values[pos[t]] = splatK;
            }
        }
    }

    if (!EMIT)
    {
        anyHits = __builtin_amdgcn_readfirstlane(anyHits);
        if (lane == 0) a.rowFlags[row] = anyHits;
        if (anyHits)
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int words = a.stride / 2;
            unsigned* out = reinterpret_cast<unsigned*>(a.counts + (size_t) row * a.stride);
            for (int i = lane * 4; i < words; i += kWave * 4) *reinterpret_cast<uint4*>(out + i) = *reinterpret_cast<const uint4*>(cursor + i);
        }
    }
}

// Column scan of counts[row][tile] over the rows.  One workgroup = 64 tile columns x 16 row segments (one wave per
// segment, a wave reads 128 contiguous bytes of a row); empty rows are skipped through a ballot of their flags.
__global__ __launch_bounds__(kWave* kScanSegments) void binScanKernel(BinArgs a)
{
    __shared__ int segSum[kScanSegments][kWave];
    const int lane = threadIdx.x & (kWave - 1), seg = threadIdx.x >> 6;
    const int col = blockIdx.x * kWave + lane;
    const int count = a.countDevice ? min(*a.countDevice, a.countHost) : a.countHost;
    const int rowSplats = a.batches * kWave;
    const int rows = (count + rowSplats - 1) / rowSplats;
    const int segLen = (rows + kScanSegments - 1) / kScanSegments;
    const int r0 = min(seg * segLen, rows), r1 = min(r0 + segLen, rows);

    // Rows are visited eight at a time so that eight independent loads are in flight per round.
    constexpr int kGang = 8;
    int sum = 0;
    for (int base = r0; base < r1; base += kWave)
    {
        const int r = base + lane;
        unsigned long long live = __ballot(r < r1 && a.rowFlags[r] != 0);
        while (live != 0ull)
        {
            const int firstBit = __ffsll((long long) live) - 1;
            int v[kGang];
#pragma unroll
            for (int j = 0; j != kGang; ++j)
            {
                const bool have = live != 0ull;
                const int bit = have ? __ffsll((long long) live) - 1 : firstBit;
                live &= live - 1; // 0 stays 0
                const int c = a.counts[(size_t) (base + bit) * a.stride + col];
                v[j] = have ? c : 0;
            }
#pragma unroll
            for (int j = 0; j != kGang; ++j) sum += v[j];
        }
    }
    segSum[seg][lane] = sum;
    __syncthreads();
    int running = 0, total = 0;
#pragma unroll
    for (int s = 0; s != kScanSegments; ++s)
    {
        const int v = segSum[s][lane];
        running += (s < seg) ? v : 0;
        total += v;
    }
    for (int base = r0; base < r1; base += kWave)
    {
        const int r = base + lane;
        unsigned long long live = __ballot(r < r1 && a.rowFlags[r] != 0);
        while (live != 0ull)
        {
            const int firstBit = __ffsll((long long) live) - 1;
            int v[kGang], bits[kGang];
#pragma unroll
            for (int j = 0; j != kGang; ++j)
            {
                const bool have = live != 0ull;
                bits[j] = have ? __ffsll((long long) live) - 1 : -1;
                live &= live - 1;
                const int c = a.counts[(size_t) (base + (have ? bits[j] : firstBit)) * a.stride + col];
                v[j] = have ? c : 0;
            }
#pragma unroll
            for (int j = 0; j != kGang; ++j)
            {
                if (bits[j] >= 0) a.offsets[(size_t) (base + bits[j]) * a.stride + col] = (unsigned) running;
                running += v[j];
            }
        }
    }
    if (seg == 0) a.tileTotals[col] = total; // padding columns were never counted: zero
}

// Tile starts = exclusive scan of the tile totals (one workgroup); also the range table of the band and the pair total.
constexpr int kStartThreads = 1024;
__global__ __launch_bounds__(kStartThreads) void binTileStartKernel(BinArgs a)
{
    __shared__ int waveTotals[2][kStartThreads / kWave];
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
    const Screen& scr = a.scr;
    int carry = 0, flip = 0;
    for (int base = 0; base < a.stride; base += kStartThreads * 4, flip ^= 1)
    {
        const int tile = base + t * 4;
        int4 tot = make_int4(0, 0, 0, 0);
        if (tile < a.stride) tot = *reinterpret_cast<const int4*>(a.tileTotals + tile);
        const int own = tot.x + tot.y + tot.z + tot.w;
        int incl = own;
#pragma unroll
        for (int delta = 1; delta < kWave; delta <<= 1)
        {
            const int o = __shfl_up(incl, delta);
            if (lane >= delta) incl += o;
        }
        if (lane == kWave - 1) waveTotals[flip][wave] = incl;
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w != kStartThreads / kWave; ++w)
        {
            const int ws = waveTotals[flip][w];
            before += (w < wave) ? ws : 0;
            all += ws;
        }
        const int s0 = carry + before + incl - own, s1 = s0 + tot.x, s2 = s1 + tot.y, s3 = s2 + tot.z, s4 = s3 + tot.w;
        if (tile < a.stride)
        {
            *reinterpret_cast<uint4*>(a.tileStart + tile) = make_uint4((unsigned) s0, (unsigned) s1, (unsigned) s2, (unsigned) s3);
            const int cap = a.capacity;
            const int starts[5] = {s0, s1, s2, s3, s4};
#pragma unroll
            for (int j = 0; j != 4; ++j)
            {
                if (tile + j < a.bandTiles)
                {
                    const int b = min(starts[j], cap), e = min(starts[j + 1], cap);
                    // an empty tile keeps the reference's cleared value (-1, -1) (Demo.cpp:403)
                    reinterpret_cast<int2*>(a.tileRange)[scr.rowBegin * scr.tilesX + tile + j] = (e > b) ? make_int2(b, e) : make_int2(-1, -1);
                }
            }
        }
        carry += all;
    }
    if (t == 0)
    {
        *a.total = carry;
        if (a.totalHostMapped != nullptr) *a.totalHostMapped = carry;
    }
}

} // namespace

void ensureBinScratch(BinScratch& s, int splats, int bandTiles, int batches)
{
    const int rowSplats = batches * kWave;
    const int rows = std::max(1, (splats + rowSplats - 1) / rowSplats);
    const int stride = (std::max(bandTiles, 1) + kColumnPad - 1) / kColumnPad * kColumnPad;
    if (rows > s.rowsCapacity || stride > s.strideCapacity)
    {
        const float4* record = s.tileRecord; // not owned: survives the reallocation
        freeBinScratch(s);
        s.tileRecord = record;
        s.rowsCapacity = rows;
        s.strideCapacity = stride;
        GS_CHECK(hipMalloc((void**) &s.counts, sizeof(uint16_t) * (size_t) rows * stride));
        GS_CHECK(hipMalloc((void**) &s.offsets, sizeof(uint32_t) * (size_t) rows * stride));
        GS_CHECK(hipMalloc((void**) &s.rowFlags, sizeof(int32_t) * (size_t) rows));
        GS_CHECK(hipMalloc((void**) &s.tileTotals, sizeof(int32_t) * (size_t) stride));
        GS_CHECK(hipMalloc((void**) &s.tileStart, sizeof(uint32_t) * (size_t) stride));
    }
}

void freeBinScratch(BinScratch& s)
{
    if (s.counts) (void) hipFree(s.counts);
    if (s.offsets) (void) hipFree(s.offsets);
    if (s.rowFlags) (void) hipFree(s.rowFlags);
    if (s.tileTotals) (void) hipFree(s.tileTotals);
    if (s.tileStart) (void) hipFree(s.tileStart);
    s = BinScratch{};
}

int binBatchesFor(int bandTiles)
{
    // Rows as long as the LDS cursor table is worth: the tables cost 6 bytes per (row, tile).
    return bandTiles > 4096 ? 8 : bandTiles > 1024 ? 4 : 2;
}

bool binsSupported(const Screen& scr)
{
    const int bandTiles = (scr.rowEnd - scr.rowBegin) * scr.tilesX;
    return bandTiles > 0 && bandTiles <= kBinMaxTiles;
}

namespace
{
BinArgs makeArgs(int count, const int* countDevice, const Screen& scr, const uint32_t* order, const float2* clipXY, const float4* screenEllipse,
                 const BinScratch& s, int batches, int32_t* values, int capacity, int32_t* tileRange, int64_t* total,
                 int64_t* totalHostMapped)
{
    BinArgs a{};
    a.countHost = count;
    a.countDevice = countDevice;
    a.scr = scr;
    a.batches = batches;
    a.bandTiles = (scr.rowEnd - scr.rowBegin) * scr.tilesX;
    a.stride = (a.bandTiles + kColumnPad - 1) / kColumnPad * kColumnPad;
    a.order = order;
    a.clipXY = clipXY;
    a.screenEllipse = screenEllipse;
    a.tileRecord = s.tileRecord;
    a.counts = s.counts;
    a.offsets = s.offsets;
    a.rowFlags = s.rowFlags;
    a.tileTotals = s.tileTotals;
    a.tileStart = s.tileStart;
    a.values = values;
    a.capacity = capacity;
    a.tileRange = tileRange;
    a.total = total;
    a.totalHostMapped = totalHostMapped;
    return a;
}
} // namespace

void launchBinCountScan(hipStream_t stream, int count, const int* countDevice, const Screen& scr, const uint32_t* order,
                        const float2* clipXY, const float4* screenEllipse, const BinScratch& s, int batches)
{
    if (count <= 0) return;
    const BinArgs a = makeArgs(count, countDevice, scr, order, clipXY, screenEllipse, s, batches, nullptr, 0, nullptr, nullptr, nullptr);
    const int rows = (count + batches * kWave - 1) / (batches * kWave);
    const size_t lds = sizeof(WaveStage) + sizeof(unsigned) * kWave + sizeof(uint16_t) * (size_t) a.stride;
    hipLaunchKernelGGL(binRowsKernel<false>, dim3(rows), dim3(kWave), lds, stream, a);
    hipLaunchKernelGGL(binScanKernel, dim3(a.stride / kWave), dim3(kWave * kScanSegments), 0, stream, a);
}

void launchBinEmit(hipStream_t stream, int count, const int* countDevice, const Screen& scr, const uint32_t* order, const float2* clipXY,
                   const float4* screenEllipse, const BinScratch& s, int batches, int32_t* values, int capacity, int32_t* tileRange,
                   int64_t* total, int64_t* totalHostMapped)
{
    if (count <= 0) return;
    const BinArgs a = makeArgs(count, countDevice, scr, order, clipXY, screenEllipse, s, batches, values, capacity, tileRange, total,
                               totalHostMapped);
    const int rows = (count + batches * kWave - 1) / (batches * kWave);
    const size_t lds = sizeof(WaveStage) + sizeof(unsigned) * kWave + sizeof(uint32_t) * (size_t) a.stride;
    hipLaunchKernelGGL(binTileStartKernel, dim3(1), dim3(kStartThreads), 0, stream, a); // needs the capacity: lives with the emit
    hipLaunchKernelGGL(binRowsKernel<true>, dim3(rows), dim3(kWave), lds, stream, a);
}

} // namespace gs
