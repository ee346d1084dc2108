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
//     (GaussianRender.cu:375-405) -- evaluated in a different floating-point order, and (round 4) on the two edges nearest to the
//     ellipse's centre only, which is where the smallest of the reference's five squared distances lies (see the hot loop), so
//     its decision value (that smallest squared distance, compared with 1) differs from the reference's by rounding.  A GUARD
//     BAND makes the decisions the reference's all the same: a tile whose value lies within kGuard (1 + A) of 1 is
//     decided again by the reference's own arithmetic (gs_overlap_exact.h, shared with gs_tiles.hip).  A is what the
//     rounding errors of BOTH formulations scale with: the absolute terms of the two dot products that map the tile's
//     corner into unit-circle space, before they cancel, plus the L1 size of the tile's edge vectors there.  The hot loop
//     compares a unit's four values with ONE bound per splat: a tile whose value is near 1 has its closest point within
//     the ellipse's larger extent of the centre, so its corner offset is at most (extent + tile) per axis -- A* bounds A for
//     every such tile.  tools/guard_band_check.cpp runs both formulations on the host over 10^8 tiles of harsh ellipses
//     (anisotropy up to 1000, extents 1e-4 .. 2 clip units, five resolutions): near the decision the two values are never
//     further apart than 2.5e-6 (1 + A), in every range of A; kGuard = 1e-5 is four times that, no differing decision lies
//     outside the band.  On the bench scene ~1 000 of 9 M tiles per frame are decided exactly and 1-4 decisions change
//     (tools/guard_stats.py).  The "centre inside the rectangle" term is decided per SPLAT with the reference's arithmetic
//     (which tile holds the centre) and is a single integer compare per tile.  The pair list is therefore the staged
//     builder's, bit for bit (test_two_level_pipeline_equals_reference_pipeline asserts equality).
#include "gs_aabb.h"
#include "gs_internal.h"
#include "gs_overlap_exact.h"
#include "gs_wave_match.h"

namespace gs
{
namespace
{

constexpr int kWavesPerBlock = 4;
constexpr int kGroup = kWave * kWavesPerBlock; // splats (and lanes) per workgroup
constexpr int kTilesPerUnit = 4;
constexpr int kMaskRounds = 16;                // rounds of a workgroup whose hit masks are kept for the emit pass
constexpr int kDoneBoxRows = 8; // depth layers: tile boxes of up to 8 rows x 32 columns are checked against the stopped tiles as a whole
// The emit pass sums the group totals in front of each group itself up to this many groups (1 M splats: 3 906); longer lists get
// the scan kernel.  A band-culled list (device-side count) keeps only a fraction of the groups the grid was sized for, so the
// limit on the host-side group count is wider there.
constexpr int kSelfSumGroups = 4096;
constexpr int kSelfSumGroupsCulled = 16384;
// Emit pass that carries the tile sort's first pass (SORT_BITS > 0): pairs are staged in LDS in emission order and flushed -- ranked
// stably by their low tile digit, reordered, written as runs -- when another round (up to kGroup * kTilesPerUnit pairs) might not fit.
#ifndef GS_FUSED_STAGE_ROWS
#define GS_FUSED_STAGE_ROWS 12
#endif
#ifndef GS_FUSED_XCD_CHUNK
#define GS_FUSED_XCD_CHUNK 16 // consecutive groups per XCD in the fused emit pass (0: round-robin, as dispatched); 8..64 measure alike, profiles/r04_list_fusion.txt
#endif
#ifndef GS_FUSED_COUNT_CHUNKED
#define GS_FUSED_COUNT_CHUNKED 1 // the count pass of a fused frame deals its groups the same way: a line of its digit table (16 groups) is written through one L2
#endif
constexpr int kStageRows = GS_FUSED_STAGE_ROWS;         // rows of 64 pairs per wave and flush (register arrays of this length)
constexpr int kStageCap = kStageRows * kGroup;         // pairs the staging buffer holds
constexpr int kStageFlushAt = kStageCap - kGroup * kTilesPerUnit;
static_assert(kStageFlushAt > 0, "a round's pairs must fit the staging buffer");

// OPERANDS = false (the emit pass that carries the tile sort's first pass, whose LDS goes to the staging buffer): without the operands of
// the overlap tests -- the rare rounds past the recorded ones fetch their unit's splat record again and derive them on the spot.
template <bool OPERANDS, bool CARRY = false>
struct GroupStage
{
    unsigned depth[CARRY ? kGroup : 1]; // CARRY (the staged buildTileList): getKey's quantised depth of the group's splats, the low word of their pairs' 64-bit keys
    int excl[kGroup];   // exclusive prefix of units
    int area[kGroup];   // AABB tiles
    int origin[kGroup]; // x0 | y0 << 16
    int w[kGroup];
    int centreLocal[kGroup]; // local index (row-major in the box) of the tile that strictly contains the centre, or -1 (OPERANDS = false: or -2, see the prologue)
    int splat[kGroup];
    float4 centreM[OPERANDS ? kGroup : 1]; // cx, cy, m00, m01
    float4 mInv[OPERANDS ? kGroup : 1];    // m10, m11, 1/|u|^2, 1/|v|^2
    float guard[OPERANDS ? kGroup : 1];    // kGuard (1 + A*): the splat-wide filter of the guard band
    float4 ellipse[OPERANDS ? kGroup : 1]; // cos, sin, extent x, y as the clip stage stored them: operands of the exact predicate inside the band
};

// Guard band of the cheap overlap decision (see the head of the file and tools/guard_band_check.cpp).
constexpr float kGuard = 1.0e-5f;
#if defined(GS_AUDIT_OVERLAP) && !defined(GS_AB_GUARD_COUNT)
#define GS_AB_GUARD_COUNT 1
#endif
#ifdef GS_AB_GUARD_COUNT
// diagnostics builds: [0] units tested, [1] units that entered the band's slow side, [2] tiles decided exactly, [3] decisions changed.
// -DGS_AUDIT_OVERLAP (libgsplat_hip_audit.so, `make audit`; tests/test_gpu_parity.py::test_overlap_audit_build_finds_no_differing_decision)
// additionally sends EVERY tile of the count pass through the reference's predicate and compares: [4] tiles audited, [5] tiles whose
// final decision differs from the reference's predicate (must stay 0: the guard band's claim, checked at run time instead of by
// the host model alone), [6] tiles of the reference's box that the ellipse's own bounding box left out, [7] of those, tiles the
// reference's predicate would have listed (must stay 0: the tight box's claim).
__device__ unsigned long long g_guardStats[8];
// depth layers (later layers' count passes): [0] splats with a box on screen, [1] of those, dropped because every tile under the box is
// stopped, [2] boxes too large for that test, [3] units of those, [4] units of the boxes that were tested and kept, [5] units those
// boxes would have had before being shrunk to their open tiles
__device__ unsigned long long g_layerStats[8];
#endif

// Squared distance of the origin to the segment q + t e, t in [0,1]: the edge comes within the unit circle when it is
// below 1 (GaussianRender.cu:361-372).
__device__ __forceinline__ float edgeDistanceSquared(float qx, float qy, float ex, float ey, float qe, float invEE)
{
    const float t = __builtin_amdgcn_fmed3f(-qe * invEE, 0.0f, 1.0f); // saturate in one instruction (v_med3_f32)
    const float rx = fmaf(t, ex, qx), ry = fmaf(t, ey, qy);
    return fmaf(rx, rx, ry * ry);
}

// The rare side of the guard band: the tiles of one work unit whose cheap decision value is within kGuard (1 + A) of 1 (A from the
// tile's own corner) are decided by the reference's predicate.  A lane gets here for about one unit in three thousand; the loop is
// rolled and the predicate is the register-lean form, so that this path does not set the kernel's register count.
__device__ __forceinline__ unsigned decideInsideGuardBand(unsigned hitMask, float4 dev4, float4 se, int local0, int areaK, int wk, float rcpW,
                                                          int origin, float4 cm, float4 mi, float sx, float sy)
{
    const float ux = cm.z * sx, uy = mi.x * sx, vx = cm.w * sy, vy = mi.y * sy;
    const float uvAbs = 1.0f + (fabsf(ux) + fabsf(uy)) + (fabsf(vx) + fabsf(vy));
    const EllipseD ell{cm.x, cm.y, se.z, se.w, se.x, se.y};
    static_assert(kTilesPerUnit == 4, "dev4 carries one value per tile of the unit");
#pragma unroll 1
    for (int e = 0; e != kTilesPerUnit; ++e)
    {
        const int local = local0 + e;
        if (local >= areaK) break;
        const int ly = (int) (((float) local + 0.5f) * rcpW), lx = local - ly * wk; // as the caller derives lx0, ly0
        const int gx = (origin & 0xFFFF) + lx, gy = (origin >> 16) + ly;
        const float minX = tileClipMin(gx, sx), minY = tileClipMin(gy, sy);
        const float dx = fabsf(minX - cm.x), dy = fabsf(minY - cm.y);
        const float a1 = uvAbs + ((fabsf(cm.z) + fabsf(mi.x)) * dx + (fabsf(cm.w) + fabsf(mi.y)) * dy);
        const float dev = e == 0 ? dev4.x : e == 1 ? dev4.y : e == 2 ? dev4.z : dev4.w;
        if (dev > kGuard * a1) continue; // certain (a NaN fails the comparison and is decided exactly)
        const bool exact = ellipseRectOverlapExactRolled(ell, minX, minY, minX + sx, minY + sy);
#ifdef GS_AB_GUARD_COUNT
        atomicAdd(&g_guardStats[2], 1ull);
        if (exact != ((hitMask >> e) & 1u)) atomicAdd(&g_guardStats[3], 1ull);
#endif
        hitMask = (hitMask & ~(1u << e)) | (exact ? (1u << e) : 0u);
    }
    return hitMask;
}

#ifndef GS_VISIBLE_EDGES
#define GS_VISIBLE_EDGES 1 // A/B builds: 0 = all four edges and the centre term of every tile (round 3's hot loop)
#endif
// Which tile strictly contains the centre along one axis, by the reference's arithmetic (GaussianRender.cu:382-386,656-664:
// min = g * size - 1 and max = min + size, each rounded): the candidates around floor((c + 1) / size) are tested exactly.  Returns the
// tile, -1 when none does (a centre on a boundary), -2 when two neighbours both do (their rounded intervals overlap by an ulp).
__device__ __forceinline__ int centreTileOnAxis(float c, float size, float tilesPerScreen)
{
    const float guess = floorf((c + 1.0f) * 0.5f * tilesPerScreen);
    if (!(guess > -4.0f && guess < 1.0e6f)) return -1; // off screen by far (or NaN): no tile of any box holds the centre
    const int g0 = (int) guess;
    int found = -1, count = 0;
#pragma unroll
    for (int g = g0 - 1; g <= g0 + 1; ++g)
    {
        const float lo = tileClipMin(g, size);
        if (c > lo && c < lo + size)
        {
            found = g;
            ++count;
        }
    }
    return count == 1 ? found : (count == 0 ? -1 : -2);
}

// Inverse of compactDepthKey (gs_depth_key.h): the 32-bit quantised depth of getKey() back from its lossless 27-bit re-coding.
__device__ __forceinline__ unsigned expandDepthKey(unsigned code)
{
    const unsigned cls = code >> 23;
    if (cls == 0u) return code;
    if (cls >= 10u) return 0xFFFFFFFFu;
    const unsigned top = cls + 22u;
    return (1u << top) | ((code & 0x7FFFFFu) << (top - 23u));
}

// CARRY (EMIT only; the staged buildTileList, gs_staged.hip): every pair also gets the quantised depth of its splat (decoded from the
// depth-sorted codes, `depthSorted`) written to carryOut at the pair's position -- the sort's remaining passes carry it along and the
// last one writes tile << 32 | depth as the reference's 64-bit key, without a gather.  The staging buffer then holds the splat's
// LOCAL index; splat index and depth come from the group's LDS tables when the pair is written.
template <bool EMIT, typename TileKey, int SORT_BITS, bool CARRY = false>
__global__ __launch_bounds__(kGroup) void tileListFastKernel(int countHost, const int* __restrict__ countDevice, Screen scr, const uint32_t* __restrict__ order,
                                                             const float2* __restrict__ clipXY, const float4* __restrict__ screenEllipse,
                                                             int32_t* __restrict__ groupCounts, const int32_t* __restrict__ groupOffsets,
                                                             TileKey* __restrict__ tileKeys, int32_t* __restrict__ values, int capacity,
                                                             uint8_t* __restrict__ hitMasks, const float4* __restrict__ tileRecord,
                                                             int64_t* __restrict__ total, int64_t* __restrict__ totalHostMapped, int viewSplats,
                                                             int4* __restrict__ boxRecords, int32_t* __restrict__ digitTable,
                                                             const int32_t* __restrict__ digitTotals, int digitStride,
                                                             const uint32_t* __restrict__ depthSorted, uint32_t* __restrict__ carryOut,
                                                             const uint32_t* __restrict__ openBits, int openWordsPerRow, int64_t* __restrict__ frameTotals,
                                                             int layerFlags)
{
    static_assert(!CARRY || EMIT, "only the emit pass carries depths");
    // Depth layers (gs_renderer_set_depth_layers): openBits != nullptr -- a later layer lists no pair of a tile that an earlier layer's
    // raster already stopped (its bit is clear; rows of openWordsPerRow words per tile row, one table per view); frameTotals != nullptr -- the emit pass keeps the frame's largest layer total ([0], what the list
    // capacity must hold) and the sum ([1]) and hands the largest to the host when this is the last layer (layerFlags: 1 first, 2 last).
    // SORT_BITS > 0: the first pass of the tile sort happens here.  COUNT also counts the pairs' low tile digits (digitTable[digit][group],
    // raw); EMIT gets the table scanned over the groups + the digit totals and writes every pair where a stable scatter on that digit
    // would have put it.
    constexpr bool FUSED = SORT_BITS > 0;
    constexpr int kBins = FUSED ? (1 << SORT_BITS) : 1;
    static_assert(kBins <= kGroup, "one digit per lane");
    constexpr bool LEAN = EMIT && FUSED; // no overlap operands in LDS
    __shared__ GroupStage<!LEAN, CARRY> st;
    __shared__ int digitHist[!EMIT && FUSED ? kBins : 1];
    __shared__ int sortCnt[EMIT && FUSED ? kWavesPerBlock : 1][kBins];
    __shared__ int sortLocalBase[EMIT && FUSED ? kBins : 1];
    __shared__ int sortRunBase[EMIT && FUSED ? kBins : 1];
    __shared__ int sortWaveSums[kWavesPerBlock];
    __shared__ TileKey stageKeys[EMIT && FUSED ? kStageCap : 1];
    __shared__ int32_t stageValues[EMIT && FUSED ? kStageCap : 1];
    __shared__ int waveTotals[kWavesPerBlock];
    __shared__ long long scanTotals[kWavesPerBlock];
    __shared__ int roundHits[2][kWavesPerBlock];
    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & (kWave - 1);
    // Workgroups are dealt to the 8 XCDs round-robin.  The emit pass that carries the sort's first pass writes, per digit, a run
    // right behind the run of the group before it: runs of GS_FUSED_XCD_CHUNK consecutive groups meet in ONE L2 when the groups are
    // dealt to the XCDs in chunks (any bijection works: everything is indexed by `group`; the grid is padded to whole chunks).
    constexpr int kXcdChunk = (SORT_BITS > 0 && (EMIT || GS_FUSED_COUNT_CHUNKED)) ? GS_FUSED_XCD_CHUNK : 0;
    const int slotInXcd = (int) blockIdx.x >> 3;
    const int group = kXcdChunk > 0 ? ((slotInXcd / kXcdChunk) * 8 + ((int) blockIdx.x & 7)) * kXcdChunk + slotInXcd % kXcdChunk : (int) blockIdx.x;
    const int count = countDevice ? min(*countDevice, countHost) : countHost; // device-side count: culled splat list
    if (EMIT && groupOffsets == nullptr && count <= 0 && group == 0 && t == 0)
    {
        // nothing reached the band: no workgroup gets as far as the pair total below.  (With the scan kernel in front -- groupOffsets --
        // the total is the scan's to write, once per frame: the host may re-arm the mapped word as soon as it has seen that value.)
        *total = 0;
        if (frameTotals != nullptr)
        {
            // (depth layers: a layer none of whose splats still has an open tile)
            if (layerFlags & 1) frameTotals[0] = frameTotals[1] = 0;
            if ((layerFlags & 2) && totalHostMapped != nullptr)
            {
                totalHostMapped[1] = frameTotals[1];
                __threadfence_system();
                totalHostMapped[0] = frameTotals[0];
            }
        }
        else if (totalHostMapped != nullptr) *totalHostMapped = 0;
    }
    if (group * kGroup >= count) return; // whole workgroup
    const int sorted = group * kGroup + t;
    if (!EMIT && FUSED && t < kBins) digitHist[t] = 0; // (the prefix barrier below comes before the first add)

    // What a lane knows about its splat: the tile box (always) and the operands of the overlap tests (when tests will be run).
    int area = 0, x0 = 0, y0 = 0, w = 0, splat = 0, centreLocal = -1;
    float cx = 0.f, cy = 0.f, m00 = 0.f, m01 = 0.f, m10 = 0.f, m11 = 0.f, invUU = 0.f, invVV = 0.f, guard = 0.f;
    float4 se = make_float4(0.f, 0.f, 0.f, 0.f); // cos, sin, extent.x, extent.y
    const auto loadSplat = [&]() {
        float2 c;
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
    };
    const auto overlapOperands = [&]() {
        // Affine map clip space -> the ellipse's unit-circle space (GaussianRender.cu:351-358), hoisted:
        //   l = M (p - c),  M = diag(1/ex, 1/ey) * [[cos, sin], [-sin, cos]]
        // (hardware reciprocals, 1 ulp: these are the cheap formulation's own operands -- what they cost in accuracy is inside the
        // guard band's margin, tools/guard_band_check.cpp perturbs them the same way)
        const float invEx = __builtin_amdgcn_rcpf(se.z), invEy = __builtin_amdgcn_rcpf(se.w);
        m00 = se.x * invEx;
        m01 = se.y * invEx;
        m10 = -se.y * invEy;
        m11 = se.x * invEy;
        // Images of the tile edges: u = M (sx, 0), v = M (0, sy).
        const float ux = m00 * scr.tileClipSizeX, uy = m10 * scr.tileClipSizeX;
        const float vx = m01 * scr.tileClipSizeY, vy = m11 * scr.tileClipSizeY;
        invUU = __builtin_amdgcn_rcpf(fmaf(ux, ux, uy * uy));
        invVV = __builtin_amdgcn_rcpf(fmaf(vx, vx, vy * vy));
        // Splat-wide filter of the guard band (see the head of the file): the corner of a tile whose decision value can be near 1 lies
        // within (larger extent + one tile) of the centre on either axis.
        // (1.05: a value within the widest band still used, 0.05, of 1 belongs to a point within sqrt(1.1) extents)
        const float reach = 1.05f * fmaxf(se.z, se.w);
        float aStar = (fabsf(m00) + fabsf(m10)) * (reach + scr.tileClipSizeX) + (fabsf(m01) + fabsf(m11)) * (reach + scr.tileClipSizeY) +
                      ((fabsf(ux) + fabsf(uy)) + (fabsf(vx) + fabsf(vy)));
        aStar = fmaf(0.0f, invUU + invVV, aStar); // a non-finite reciprocal (degenerate extent) poisons the bound: every tile is then decided exactly
        guard = kGuard * (1.0f + aStar);
        if (!(guard <= 0.05f)) guard = __builtin_inff(); // absurd conditioning, NaN or inf: no unit of this splat passes the "certain" test below
#ifdef GS_AB_GUARD_NEVER
        guard = -1.0f; // A/B builds: the band is never entered (what the hot loop's share of the guard costs)
#endif
    };
    if (sorted < count)
    {
        splat = (int) order[sorted];
        if (!EMIT)
        {
            loadSplat();
            // AABB in tiles, exactly as gs_tiles.hip / GaussianRender.cu:408-436,537-551 (shared with the band cull).
            TileAabb box = splatTileAabb(cx, cy, se.x, se.y, se.z, se.w, scr);
#ifdef GS_AUDIT_OVERLAP
            const TileAabb referenceBox = box;
#endif
#if GS_TIGHT_BOX
            tightenToEllipse(box, cx, cy, se, scr);
#endif
#ifdef GS_AUDIT_OVERLAP
            if (referenceBox.x1 > referenceBox.x0 && referenceBox.y1 > referenceBox.y0)
            {
                // every tile the tight box dropped from the reference's box must be a tile the reference's predicate rejects
                const EllipseD ell{cx, cy, se.z, se.w, se.x, se.y};
                const bool kept = box.x1 > box.x0 && box.y1 > box.y0;
#pragma unroll 1
                for (int gy = referenceBox.y0; gy < referenceBox.y1; ++gy)
#pragma unroll 1
                    for (int gx = referenceBox.x0; gx < referenceBox.x1; ++gx)
                    {
                        if (kept && gx >= box.x0 && gx < box.x1 && gy >= box.y0 && gy < box.y1) continue;
                        const float minX = tileClipMin(gx, scr.tileClipSizeX), minY = tileClipMin(gy, scr.tileClipSizeY);
                        atomicAdd(&g_guardStats[6], 1ull);
                        if (ellipseRectOverlapExactRolled(ell, minX, minY, minX + scr.tileClipSizeX, minY + scr.tileClipSizeY))
                            atomicAdd(&g_guardStats[7], 1ull);
                    }
            }
#endif
            x0 = box.x0;
            y0 = box.y0;
            w = box.x1 - box.x0;
            const int h = box.y1 - box.y0;
            area = (w > 0 && h > 0) ? w * h : 0;
            // Depth layers: a splat whose whole box lies in tiles that earlier layers have stopped makes no work units at all (the per-tile
            // mask further down would take its pairs out one by one, but a wave only saves instructions when whole units disappear).
            // Boxes of up to 8 x 32 tiles are looked at (one or two words of open bits per row); larger ones keep their units.
            if (openBits != nullptr && area > 0 && w <= 32 && h <= kDoneBoxRows)
            {
                int rowBase = 0; // (several views per chain: one table per view)
                if (viewSplats > 0)
#pragma unroll
                    for (int v = 1; v != kMaxViews; ++v) rowBase += (splat >= v * viewSplats) ? scr.tilesY : 0;
                const int wi = box.x0 >> 5, sh = box.x0 & 31;
                const unsigned mask = w == 32 ? 0xFFFFFFFFu : ((1u << w) - 1u);
                const bool straddles = sh + w > 32;
                unsigned open = 0;
#pragma unroll
                for (int r = 0; r != kDoneBoxRows; ++r) // (independent loads: no early exit)
                {
                    if (r < h)
                    {
                        const uint32_t* row = openBits + (size_t) (rowBase + box.y0 + r) * openWordsPerRow + wi;
                        unsigned bits = row[0] >> sh;
                        if (straddles) bits |= row[1] << (32 - sh);
                        open |= bits & mask;
                    }
                }
                if (open == 0) area = 0;
#ifdef GS_AB_GUARD_COUNT
                atomicAdd(&g_layerStats[0], 1ull);
                if (open == 0) atomicAdd(&g_layerStats[1], 1ull);
                else atomicAdd(&g_layerStats[4], (unsigned long long) ((area + kTilesPerUnit - 1) / kTilesPerUnit));
#endif
            }
#ifdef GS_AB_GUARD_COUNT
            else if (openBits != nullptr && area > 0)
            {
                atomicAdd(&g_layerStats[0], 1ull);
                atomicAdd(&g_layerStats[2], 1ull);
                atomicAdd(&g_layerStats[3], (unsigned long long) ((area + kTilesPerUnit - 1) / kTilesPerUnit));
            }
#endif
            overlapOperands();
            // The tile that holds the centre (GaussianRender.cu:382-386), found once per splat with the reference's rounding.
            const int cgx = centreTileOnAxis(cx, scr.tileClipSizeX, scr.tilesPerScreenX), cgy = centreTileOnAxis(cy, scr.tileClipSizeY, scr.tilesPerScreenY);
            if (cgx == -2 || cgy == -2) centreLocal = -2; // two candidate tiles an ulp apart: every tile of this splat goes to the exact predicate
            else if (cgx >= box.x0 && cgx < box.x1 && cgy >= box.y0 && cgy < box.y1) centreLocal = (cgy - box.y0) * w + (cgx - box.x0);
            // the box, for the emit pass: it replays the recorded hit masks and needs neither the splat's record nor any of the above
            boxRecords[sorted] = make_int4(area, x0 | (y0 << 16), w, centreLocal);
        }
        else
        {
            const int4 b = boxRecords[sorted];
            area = b.x;
            x0 = b.y & 0xFFFF;
            y0 = b.y >> 16;
            w = b.z;
            centreLocal = b.w;
        }
        if (!LEAN && centreLocal == -2) // (LEAN keeps the flag: the guard is not staged, the late rounds look at the flag itself)
        {
            guard = __builtin_inff();
            centreLocal = -1;
        }
    }
    const int units = (area + kTilesPerUnit - 1) / kTilesPerUnit;

    // exclusive prefix of the units over the workgroup's 256 splats
    const int incl = waveInclusiveSum(units);
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
        if (!EMIT && FUSED && t < kBins) digitTable[(size_t) t * digitStride + group] = 0;
        return;
    }
    // EMIT: where this workgroup's pairs start = the pairs of all groups before it.  Up to kSelfSumGroups groups every workgroup
    // sums the count pass's group totals in front of it itself (a few thousand L2-resident words, 16 per lane) instead of reading the
    // result of a one-workgroup scan kernel: one kernel and one dependent launch less per frame; the last group also leaves the pair
    // total.  Beyond that (5 M splats: 19.5 K groups, 190 M word loads) the self-summing would be quadratic work: the host launches
    // the scan kernel in between and the offsets (and the total) come from there.
    int outBase = 0;
    int digitCursor = 0; // FUSED EMIT, lane d < kBins: where this group's next pair of digit d goes
    if (EMIT && FUSED)
    {
        // first output position of digit d for this group = the pairs of all smaller digits (exclusive scan of the digit totals over
        // the lanes) + the pairs of digit d in the groups before this one (the scanned table); the last group leaves the pair total.
        const unsigned myTotal = t < kBins ? (unsigned) digitTotals[t] : 0u;
        const int myEntry = t < kBins ? digitTable[(size_t) t * digitStride + group] : 0;
        const unsigned long long incl64 = waveInclusiveSumWide(myTotal);
        __syncthreads(); // waveTotals: everybody is past the prologue's reads
        if (lane == kWave - 1) scanTotals[wave] = (long long) incl64;
        __syncthreads();
        long long before = 0, all = 0;
#pragma unroll
        for (int wv = 0; wv != kWavesPerBlock; ++wv)
        {
            const long long v = scanTotals[wv];
            before += (wv < wave) ? v : 0;
            all += v;
        }
        digitCursor = (int) (unsigned) ((unsigned long long) before + incl64 - myTotal + (unsigned) myEntry);
        if (group == lastGroup && t == 0)
        {
            *total = all;
            if (frameTotals != nullptr)
            {
                const long long largest = (layerFlags & 1) ? all : max(all, (long long) frameTotals[0]);
                frameTotals[0] = largest;
                frameTotals[1] = (layerFlags & 1) ? all : frameTotals[1] + all;
                if ((layerFlags & 2) && totalHostMapped != nullptr)
                {
                    totalHostMapped[1] = frameTotals[1]; // (the sum first: the host waits for word 0)
                    __threadfence_system();
                    totalHostMapped[0] = largest;
                }
            }
            else if (totalHostMapped != nullptr) *totalHostMapped = all;
        }
        if (totalUnits == 0) return;
    }
    else if (EMIT && groupOffsets != nullptr)
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

    if (EMIT && !LEAN && totalUnits > kMaskRounds * kGroup && sorted < count)
    {
        // rounds past the recorded ones are tested again: only then does the emit pass need the splats themselves
        // (uniform branch; a group of 256 splats with more than 4 096 work units is a handful of near, large splats)
        const bool poisoned = guard > 1.0f;
        loadSplat();
        overlapOperands();
        if (poisoned) guard = __builtin_inff();
    }
    if constexpr (CARRY) st.depth[t] = sorted < count ? expandDepthKey(depthSorted[sorted]) : 0u;
    st.excl[t] = unitsBefore + incl - units;
    st.area[t] = area;
    st.origin[t] = x0 | (y0 << 16);
    st.w[t] = max(w, 1);
    st.centreLocal[t] = centreLocal;
    st.splat[t] = splat;
    if constexpr (!LEAN)
    {
        st.centreM[t] = make_float4(cx, cy, m00, m01);
        st.mInv[t] = make_float4(m10, m11, invUU, invVV);
        st.guard[t] = guard;
        st.ellipse[t] = se;
    }
    __syncthreads();

    int running = 0; // COUNT: hits of this wave so far; EMIT: hits of the whole workgroup in the rounds before this one

    // FUSED EMIT: the first pass of the tile sort on the `staged` pairs in the staging buffer (emission order), exactly what
    // radixScatterKernel does with a tile of the unsorted list (gs_radix.hip): stable rank inside each wave's contiguous share
    // (ballot match + running per-digit counters), prefixes over the waves and over the digits, reorder in LDS, runs of consecutive
    // addresses out.  Pairs of one tile keep their (depth) order; the group's digit cursors move on by what was written.
    const auto flushStaged = [&](int staged) {
        (void) staged;
        if constexpr (EMIT && FUSED)
        {
            constexpr int kDigitMask = kBins - 1;
            const int rows = (staged + kGroup - 1) / kGroup; // rows of 64 per wave, <= kStageRows
            const int waveBase = wave * rows * kWave;
            if (t < kBins)
#pragma unroll
                for (int wv = 0; wv != kWavesPerBlock; ++wv) sortCnt[wv][t] = 0;
            __syncthreads(); // the staged pairs and the zeroed counters are visible
            TileKey key[kStageRows];
            int32_t val[kStageRows];
            int rank[kStageRows];
#pragma unroll
            for (int r = 0; r != kStageRows; ++r)
            {
                rank[r] = -1;
                if (r < rows) // uniform
                {
                    const int idx = waveBase + r * kWave + lane;
                    const bool valid = idx < staged;
                    key[r] = valid ? stageKeys[idx] : TileKey(0);
                    val[r] = valid ? stageValues[idx] : 0;
                    const int digit = (int) key[r] & kDigitMask;
                    const unsigned long long m = matchDigit<SORT_BITS>(digit, valid);
                    // every lane of a digit group reads the running count (one broadcast read), then the group's first lane -- the one
                    // with no lane of the group below it -- bumps it: LDS instructions of a wave execute in order, so the next row sees
                    // the bumped value
                    const int below = lanesBelow(m);
                    const int old = valid ? sortCnt[wave][digit] : 0;
                    if (valid && below == 0) sortCnt[wave][digit] = old + __popcll(m);
                    if (valid) rank[r] = old + below;
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __syncthreads();
            // per digit (lane d): exclusive prefixes over the waves, the digit's first slot in the buffer, its first output position
            int sum = 0;
            if (t < kBins)
            {
#pragma unroll
                for (int wv = 0; wv != kWavesPerBlock; ++wv)
                {
                    const int c = sortCnt[wv][t];
                    sortCnt[wv][t] = sum;
                    sum += c;
                }
            }
            const int incl = waveInclusiveSum(sum);
            if (lane == kWave - 1) sortWaveSums[wave] = incl;
            __syncthreads();
            int beforeWaves = 0;
#pragma unroll
            for (int wv = 0; wv != kWavesPerBlock; ++wv) beforeWaves += (wv < wave) ? sortWaveSums[wv] : 0;
            if (t < kBins)
            {
                const int localBase = beforeWaves + incl - sum;
                sortLocalBase[t] = localBase;
                sortRunBase[t] = digitCursor - localBase; // output position of buffer slot j of this digit = sortRunBase + j
                digitCursor += sum;
            }
            __syncthreads();
            // reorder by digit, in place (every pair of the buffer is in a register by now)
#pragma unroll
            for (int r = 0; r != kStageRows; ++r)
            {
                if (r < rows && rank[r] >= 0)
                {
                    const int digit = (int) key[r] & kDigitMask;
                    const int slot = sortLocalBase[digit] + sortCnt[wave][digit] + rank[r];
                    stageKeys[slot] = key[r];
                    stageValues[slot] = val[r];
                }
            }
            __syncthreads();
            // runs out (unsigned compare: a demand beyond the capacity -- or beyond 2^31 -- is dropped, the frame is rendered again)
            for (int j = t; j < staged; j += kGroup)
            {
                const TileKey k = stageKeys[j];
                const unsigned pos = (unsigned) (sortRunBase[(int) k & kDigitMask] + j);
                if (pos < (unsigned) capacity)
                {
                    tileKeys[pos] = k;
                    if constexpr (CARRY)
                    {
                        const int local = stageValues[j]; // the splat's index in this group
                        values[pos] = st.splat[local];
                        carryOut[pos] = st.depth[local];
                    }
                    else values[pos] = stageValues[j];
                }
            }
        }
    };
    const float sx = scr.tileClipSizeX, sy = scr.tileClipSizeY;

    unsigned nextMask = 0;
    if (EMIT) nextMask = hitMasks[(size_t) group * kMaskRounds * kGroup + t]; // round 0 (totalUnits > 0 here)
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
        // local0 / wk through the (approximate, 1 ulp) reciprocal: the quotient's fractional part is at least 0.5/wk away from an
        // integer, far above the float error for local0 < 2^16.
        const float rcpW = __builtin_amdgcn_rcpf((float) wk);
        int centreK = st.centreLocal[k];
        const bool centreAmbiguous = LEAN && centreK == -2; // (two candidate centre tiles an ulp apart: every tile of the splat is decided exactly)
        if (centreAmbiguous) centreK = -1;
        const int ly0 = (int) (((float) local0 + 0.5f) * rcpW);
        const int lx0 = local0 - ly0 * wk;

        // The count pass leaves the four hit bits of every lane for its first kMaskRounds rounds; the emit pass replays
        // them instead of repeating the overlap tests (rounds past that are simply tested again). Uniform branches.
        const bool recorded = round < kMaskRounds;
        uint8_t* maskSlot = hitMasks + ((size_t) group * kMaskRounds + (size_t) (recorded ? round : 0)) * kGroup + t;
        unsigned hitMask = 0;
        if (EMIT && recorded)
        {
            // (the next round's masks are fetched one round early: their address depends on nothing the round computes)
            hitMask = nextMask;
            if (round + 1 < kMaskRounds && base + kGroup < totalUnits) nextMask = maskSlot[kGroup];
        }
        else
        {
            float4 cm, mi, seK;
            float guardK;
            if constexpr (LEAN)
            {
                // the unit's splat once more (a round past the recorded ones: a group of near, large splats)
                splat = st.splat[k];
                loadSplat();
                overlapOperands();
                cm = make_float4(cx, cy, m00, m01);
                mi = make_float4(m10, m11, invUU, invVV);
                seK = se;
                guardK = centreAmbiguous ? __builtin_inff() : guard;
            }
            else
            {
                cm = st.centreM[k];
                mi = st.mInv[k];
                seK = st.ellipse[k];
                guardK = st.guard[k];
            }
            const float ux = cm.z * sx, uy = mi.x * sx; // u = M (sx, 0)
            const float vx = cm.w * sy, vy = mi.y * sy; // v = M (0, sy)
            const float uv = fmaf(ux, vx, uy * vy);
            const float negHalfSx = -0.5f * sx, negHalfSy = -0.5f * sy;
            int lx = lx0, ly = ly0;
            int histTileBase = 0; // several views per chain: the virtual tile of view v is v * totalTiles + tile, as the emit pass writes it
            if (!EMIT && FUSED && viewSplats > 0)
            {
                const int splatK = st.splat[k];
#pragma unroll
                for (int v = 1; v != kMaxViews; ++v) histTileBase += (splatK >= v * viewSplats) ? scr.totalTiles : 0;
            }
            // depth layers: tiles that an earlier layer's raster has stopped take no more pairs
            unsigned doneBits = 0;
            if (openBits != nullptr && valid)
            {
                int rowBase = 0; // (several views per chain: one table per view)
                if (viewSplats > 0)
                {
                    const int splatD = st.splat[k];
#pragma unroll
                    for (int v = 1; v != kMaxViews; ++v) rowBase += (splatD >= v * viewSplats) ? scr.tilesY : 0;
                }
                int dx_ = lx0, dy_ = ly0;
#pragma unroll
                for (int e = 0; e != kTilesPerUnit; ++e)
                {
                    if (local0 + e < areaK)
                    {
                        const int gx = (origin & 0xFFFF) + dx_, gy = (origin >> 16) + dy_;
                        const unsigned word = openBits[(size_t) (rowBase + gy) * openWordsPerRow + (gx >> 5)];
                        doneBits |= ((word >> (gx & 31)) & 1u) ? 0u : (1u << e);
                    }
                    ++dx_;
                    if (dx_ == wk)
                    {
                        dx_ = 0;
                        ++dy_;
                    }
                }
            }
            float dev[kTilesPerUnit]; // |decision value - 1| of the unit's tiles (tiles past the box: certain)
#pragma unroll
            for (int e = 0; e != kTilesPerUnit; ++e)
            {
                bool hit = false;
                dev[e] = __builtin_inff();
                if (local0 + e < areaK)
                {
                    const int gx = (origin & 0xFFFF) + lx, gy = (origin >> 16) + ly;
                    // GaussianRender.cu:656-664, product and difference rounded as the reference rounds them: a fused g * s - 1 is up to
                    // an ulp of 1 off, which the map into unit-circle space magnifies by 1 / extent -- more than the guard band allows for
                    const float minX = tileClipMin(gx, sx), minY = tileClipMin(gy, sy);
                    // centre inside the rectangle (strict, GaussianRender.cu:382-386): decided per splat, exactly (prologue)
                    hit = local0 + e == centreK;
                    // corner q0 = M (min - c); the other corners are q0 + u, q0 + v, q0 + u + v
                    const float dx = minX - cm.x, dy = minY - cm.y;
                    const float q0x = fmaf(cm.z, dx, cm.w * dy), q0y = fmaf(mi.x, dx, mi.y * dy);
                    const float q0u = fmaf(q0x, ux, q0y * uy), q0v = fmaf(q0x, vx, q0y * vy);
#if GS_VISIBLE_EDGES
                    // Rectangle centre inside the ellipse (:389), or an edge within the unit circle (:392-402): the smallest of five
                    // squared distances decides -- and for a rectangle that does not hold the ellipse's centre that smallest value lies
                    // on an edge the centre can SEE (the quadratic form grows along every ray from the centre): the left edge when the
                    // centre is left of the tile, the right one when it is right of it; likewise bottom / top; the rectangle's own
                    // centre is never closer than where the segment towards it enters the rectangle.  Inside the tile's column (row)
                    // neither vertical (horizontal) edge is visible and either may be taken; the NEARER one (one compare against the
                    // tile's midline, which outside the column is the visible edge) also covers a centre within rounding of a tile
                    // boundary: the tile it grazes gets a distance of ~0 whichever side the arithmetic puts it on.  Two edge distances
                    // instead of four and the centre term, the same bits as those (fma(1, u, q) = q + u); a dropped term that would
                    // have won by rounding moves the value by at most 3e-7 (1 + A), inside the guard band's margin
                    // (tools/guard_band_check.cpp, 9 x 10^8 tiles: no differing decision outside the band).
                    const float fx = dx < negHalfSx ? 1.0f : 0.0f, fy = dy < negHalfSy ? 1.0f : 0.0f;
                    const float xEdge2 = edgeDistanceSquared(fmaf(fx, ux, q0x), fmaf(fx, uy, q0y), vx, vy, fmaf(fx, uv, q0v), mi.w);
                    const float yEdge2 = edgeDistanceSquared(fmaf(fy, vx, q0x), fmaf(fy, vy, q0y), ux, uy, fmaf(fy, uv, q0u), mi.z);
                    const float decision = fminf(xEdge2, yEdge2);
                    hit |= decision < 1.0f;
                    dev[e] = fabsf(decision - 1.0f);
#else
                    const float qcx = fmaf(0.5f, ux + vx, q0x), qcy = fmaf(0.5f, uy + vy, q0y);
                    // rectangle centre inside the ellipse (:389), or an edge within the unit circle (:392-402): the smallest of
                    // the five squared distances decides
                    const float centre2 = fmaf(qcx, qcx, qcy * qcy);
                    const float bottom2 = edgeDistanceSquared(q0x, q0y, ux, uy, q0u, mi.z);
                    const float right2 = edgeDistanceSquared(q0x + ux, q0y + uy, vx, vy, q0v + uv, mi.w);
                    const float top2 = edgeDistanceSquared(q0x + vx, q0y + vy, ux, uy, q0u + uv, mi.z);
                    const float left2 = edgeDistanceSquared(q0x, q0y, vx, vy, q0v, mi.w);
                    const float decision = fminf(fminf(fminf(bottom2, right2), fminf(top2, left2)), centre2);
                    hit |= decision < 1.0f;
                    dev[e] = fabsf(decision - 1.0f);
#endif
                    hit = hit && !(doneBits & (1u << e));
                    // digit histogram of this group's pairs (the emit pass will rank them by it), by the cheap decision: a decision the
                    // guard band turns over is put right below
                    if (!EMIT && FUSED && hit) atomicAdd(&digitHist[(gy * scr.tilesX + gx + histTileBase) & (kBins - 1)], 1);
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
            // Guard band: a decision value this close to 1 may fall on the other side in the reference's arithmetic.  First the
            // splat-wide bound (one compare per unit); the few units that fail it look at each tile with the tile's own S and
            // decide the ones inside the band with the reference's predicate, operation for operation (gs_overlap_exact.h).
            // Every comparison is written so that a NaN (degenerate extents) lands on the exact side.
#ifdef GS_AB_GUARD_COUNT
            if (!EMIT && valid) atomicAdd(&g_guardStats[0], 1ull);
            if (!EMIT && !(fminf(fminf(dev[0], dev[1]), fminf(dev[2], dev[3])) > guardK)) atomicAdd(&g_guardStats[1], 1ull);
#endif
            if (!(fminf(fminf(dev[0], dev[1]), fminf(dev[2], dev[3])) > guardK))
            {
                const unsigned cheapMask = hitMask;
                hitMask = decideInsideGuardBand(hitMask, make_float4(dev[0], dev[1], dev[2], dev[3]), seK, local0, areaK, wk, rcpW, origin, cm,
                                                mi, sx, sy) &
                          ~doneBits;
                if (!EMIT && FUSED && hitMask != cheapMask)
                {
                    // (a handful of tiles per frame) the histogram follows the exact decision
                    int wx = lx0, wy = ly0;
#pragma unroll 1
                    for (int e = 0; e != kTilesPerUnit; ++e)
                    {
                        if ((hitMask ^ cheapMask) & (1u << e))
                        {
                            const int gx = (origin & 0xFFFF) + wx, gy = (origin >> 16) + wy;
                            atomicAdd(&digitHist[(gy * scr.tilesX + gx + histTileBase) & (kBins - 1)], (hitMask & (1u << e)) ? 1 : -1);
                        }
                        ++wx;
                        if (wx == wk)
                        {
                            wx = 0;
                            ++wy;
                        }
                    }
                }
            }
#ifdef GS_AUDIT_OVERLAP
            if (!EMIT)
            {
                const EllipseD ell{cm.x, cm.y, seK.z, seK.w, seK.x, seK.y};
                int ax = lx0, ay = ly0;
#pragma unroll 1
                for (int e = 0; e != kTilesPerUnit; ++e)
                {
                    if (local0 + e < areaK)
                    {
                        const int gx = (origin & 0xFFFF) + ax, gy = (origin >> 16) + ay;
                        const float minX = tileClipMin(gx, sx), minY = tileClipMin(gy, sy);
                        const bool exact = ellipseRectOverlapExactRolled(ell, minX, minY, minX + sx, minY + sy);
                        atomicAdd(&g_guardStats[4], 1ull);
                        if (exact != ((hitMask >> e) & 1u)) atomicAdd(&g_guardStats[5], 1ull);
                    }
                    ++ax;
                    if (ax == wk)
                    {
                        ax = 0;
                        ++ay;
                    }
                }
            }
#endif
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

        if (!EMIT)
        {
            running += waveHits;
        }
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
            // (FUSED: position in the staging buffer -- `running` counts the pairs staged since the last flush)
            unsigned pos = (FUSED ? 0u : (unsigned) outBase) + (unsigned) running + (unsigned) before;
#pragma unroll
            for (int e = 0; e != kTilesPerUnit; ++e) pos = (unsigned) lanesBelow(ballots[e], (int) pos);
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
                    const int gx = (origin & 0xFFFF) + wx, gy = (origin >> 16) + wy;
                    if (FUSED)
                    {
                        stageKeys[pos] = (TileKey) (gy * scr.tilesX + gx + viewTileBase);
                        stageValues[pos] = CARRY ? k : splatK;
                    }
                    else if (pos < (unsigned) capacity)
                    {
                        tileKeys[pos] = (TileKey) (gy * scr.tilesX + gx + viewTileBase);
                        values[pos] = splatK;
                        if constexpr (CARRY) carryOut[pos] = st.depth[k];
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
            if (FUSED && (running > kStageFlushAt || base + kGroup >= totalUnits))
            {
                flushStaged(running); // (uniform; its last reads of the staging buffer are behind the next round's barrier)
                running = 0;
            }
        }
    }

    if (!EMIT)
    {
        __syncthreads(); // waveTotals: everybody is past the prologue's reads
        if (FUSED && t < kBins) digitTable[(size_t) t * digitStride + group] = digitHist[t];
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

template <typename TileKey, int SORT_BITS>
void launchEmit(hipStream_t stream, dim3 grid, dim3 block, int count, const Screen& scr, const uint32_t* order, const float2* clipXY,
                const float4* screenEllipse, const TileListScratch& s, const int32_t* groupOffsets, void* tileKeys, int32_t* values, int capacity,
                int viewSplats)
{
    if (s.carryOut != nullptr) // (the staged buildTileList: see CARRY)
        hipLaunchKernelGGL((tileListFastKernel<true, TileKey, SORT_BITS, true>), grid, block, 0, stream, count, s.countDevice, scr, order, clipXY, screenEllipse,
                           s.waveCounts, groupOffsets, (TileKey*) tileKeys, values, capacity, s.hitMasks, s.tileRecord, s.total,
                           s.totalHostMapped, viewSplats, (int4*) s.boxRecords, s.digitScanned, s.digitTotals, s.digitStride, s.depthSorted, s.carryOut,
                           s.openBits, s.openWordsPerRow, s.frameTotals, s.layerFlags);
    else
        hipLaunchKernelGGL((tileListFastKernel<true, TileKey, SORT_BITS, false>), grid, block, 0, stream, count, s.countDevice, scr, order, clipXY, screenEllipse,
                           s.waveCounts, groupOffsets, (TileKey*) tileKeys, values, capacity, s.hitMasks, s.tileRecord, s.total,
                           s.totalHostMapped, viewSplats, (int4*) s.boxRecords, s.digitScanned, s.digitTotals, s.digitStride, (const uint32_t*) nullptr,
                           (uint32_t*) nullptr, s.openBits, s.openWordsPerRow, s.frameTotals, s.layerFlags);
}

template <int SORT_BITS>
void launchCount(hipStream_t stream, int groups, int count, const Screen& scr, const uint32_t* order, const float2* clipXY, const float4* screenEllipse,
                 TileListScratch& s, int viewSplats)
{
    if (SORT_BITS > 0 && GS_FUSED_COUNT_CHUNKED && GS_FUSED_XCD_CHUNK > 0)
        groups = (groups + 8 * GS_FUSED_XCD_CHUNK - 1) / (8 * GS_FUSED_XCD_CHUNK) * (8 * GS_FUSED_XCD_CHUNK); // whole chunks on every XCD
    hipLaunchKernelGGL((tileListFastKernel<false, uint16_t, SORT_BITS>), dim3(groups), dim3(kGroup), 0, stream, count, s.countDevice, scr, order, clipXY,
                       screenEllipse, s.waveCounts, (const int32_t*) nullptr, (uint16_t*) nullptr, (int32_t*) nullptr, 0, s.hitMasks, s.tileRecord,
                       (int64_t*) nullptr, (int64_t*) nullptr, viewSplats, (int4*) s.boxRecords, s.digitTable, (const int32_t*) nullptr, s.digitStride,
                       (const uint32_t*) nullptr, (uint32_t*) nullptr, s.openBits, s.openWordsPerRow, (int64_t*) nullptr, 0);
}

} // namespace

#ifdef GS_AB_GUARD_COUNT
extern "C" void gs_debug_guard_stats(unsigned long long out[4], int reset)
{
    GS_CHECK(hipDeviceSynchronize());
    GS_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_guardStats), sizeof(unsigned long long) * 4));
    if (reset)
    {
        const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        GS_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_guardStats), zero, sizeof(zero)));
    }
}
extern "C" void gs_debug_layer_stats(unsigned long long out[8], int reset)
{
    GS_CHECK(hipDeviceSynchronize());
    GS_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_layerStats), sizeof(unsigned long long) * 8));
    if (reset)
    {
        const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        GS_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_layerStats), zero, sizeof(zero)));
    }
}
// all eight counters (the audit build's four after the guard band's four, see g_guardStats)
extern "C" void gs_debug_overlap_audit(unsigned long long out[8], int reset)
{
    GS_CHECK(hipDeviceSynchronize());
    GS_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_guardStats), sizeof(unsigned long long) * 8));
    if (reset)
    {
        const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        GS_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_guardStats), zero, sizeof(zero)));
    }
}
#endif

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
        if (s.boxRecords) GS_CHECK(hipFree(s.boxRecords));
        GS_CHECK(hipMalloc((void**) &s.hitMasks, (size_t) groups * kMaskRounds * kGroup));
        GS_CHECK(hipMalloc(&s.boxRecords, (size_t) groups * kGroup * 16)); // (area, origin, width, centre tile) per sorted splat
        // (emit pass with the tile sort's first pass: up to 256 digit rows of `groups` counters, raw and scanned)
        if (s.digitTable) GS_CHECK(hipFree(s.digitTable));
        if (s.digitScanned) GS_CHECK(hipFree(s.digitScanned));
        s.digitStride = (groups + 3) & ~3;
        GS_CHECK(hipMalloc((void**) &s.digitTable, sizeof(int32_t) * 256 * (size_t) s.digitStride));
        GS_CHECK(hipMalloc((void**) &s.digitScanned, sizeof(int32_t) * 256 * (size_t) s.digitStride));
        if (s.digitTotals == nullptr) GS_CHECK(hipMalloc((void**) &s.digitTotals, sizeof(int32_t) * 256));
        s.capacityMaskGroups = groups;
    }
}

// ---- depth layers: the later layers walk only the splats that still have an open tile under their box ----------------------------------
// A later layer's count / emit passes spend their time per GROUP of 256 depth-sorted splats (prologue, prefix sums, at least one round,
// the digit table), and at BASELINE.json configs[4] 87-93 % of a later layer's splats lie wholly in stopped tiles -- spread evenly over
// the groups, so no group disappears (profiles/r05_depth_layers.txt).  Three small kernels in front of the count pass make the groups
// dense again: (1) every splat, in STORAGE order, has its box (8 bytes, left by the clip kernel) tested against the open bits and leaves one bit; (2) the layer's
// slice of the depth order counts its live splats per block of 1 024 and (3) writes them out, in order, with the count for the passes'
// device-side count.  The test is the count pass's own: the same box (splatTileBox, left by the clip kernel) and the same limit on its
// size (at most 32 tiles wide and kDoneBoxRows tall; larger boxes are kept unseen by both).
constexpr int kCompactItems = 4;
constexpr int kCompactBlock = kGroup * kCompactItems;
constexpr int kCompactSelfSumBlocks = 4096; // (4 M splats in a layer's slice)

constexpr int kAliveTableWords = 8192; // the open-bit table of all views in LDS when it fits (3840x2160: 1 215 words per view)
template <bool TABLE_IN_LDS>
__global__ __launch_bounds__(kGroup) void layerAliveKernel(int count, Screen scr, const ushort4* __restrict__ tileBox, int viewSplats,
                                                           const uint32_t* __restrict__ openBitsGlobal, int openWordsPerRow, int tableWords,
                                                           unsigned long long* __restrict__ aliveWords)
{
    // Sixteen scattered one-word loads per splat keep the texture path busy for ~16 cycles each (28 us at configs[4], the 40 MB of boxes
    // would take 8): the table is a few KB, so every workgroup copies it into LDS once and walks many splats.
    __shared__ uint32_t table[TABLE_IN_LDS ? kAliveTableWords : 1];
    if (TABLE_IN_LDS)
    {
        for (int i = threadIdx.x; i < tableWords; i += kGroup) table[i] = openBitsGlobal[i];
        __syncthreads();
    }
    const uint32_t* openBits = TABLE_IN_LDS ? table : openBitsGlobal;
    // (four splats per lane and round: their box loads are in flight together -- one load per round left every wave waiting for memory,
    // 23 us for 5 M splats whatever the test cost)
    constexpr int kPerLane = 4;
    for (int base = (int) blockIdx.x * kGroup * kPerLane; base < count; base += (int) gridDim.x * kGroup * kPerLane)
    {
        ushort4 boxes[kPerLane];
#pragma unroll
        for (int k = 0; k != kPerLane; ++k)
        {
            const int splat = base + k * kGroup + (int) threadIdx.x;
            boxes[k] = splat < count ? tileBox[splat] : make_ushort4(0, 0, 0, 0); // (x0, y0, x1, y1) as the clip kernel left it
        }
#pragma unroll
        for (int k = 0; k != kPerLane; ++k)
        {
            const int splat = base + k * kGroup + (int) threadIdx.x;
            const ushort4 box = boxes[k];
            bool alive = false;
            const int w = (int) box.z - (int) box.x, h = (int) box.w - (int) box.y;
            if (w > 0 && h > 0)
            {
                int rowBase = 0;
                if (viewSplats > 0)
#pragma unroll
                    for (int v = 1; v != kMaxViews; ++v) rowBase += (splat >= v * viewSplats) ? scr.tilesY : 0;
                const uint32_t* row = openBits + (size_t) (rowBase + box.y) * openWordsPerRow;
                const int w0 = box.x >> 5, w1 = (box.z - 1) >> 5;
                const unsigned firstMask = 0xFFFFFFFFu << (box.x & 31), lastMask = 0xFFFFFFFFu >> (31 - ((box.z - 1) & 31));
                unsigned open = 0;
                // (the count pass's own limit, gs_tiles_fast.hip "whole box": up to 32 tiles wide -- one or two words per row -- and
                // kDoneBoxRows tall.  Round 5 looked at boxes of up to 64 tiles here, which the count pass keeps unseen: the results
                // agreed only because the per-tile mask removed those pairs again)
                if (w <= 32 && h <= kDoneBoxRows)
                {
#pragma unroll
                    for (int r = 0; r != kDoneBoxRows; ++r) // (independent loads)
                        if (r < h)
                        {
                            const unsigned a = row[(size_t) r * openWordsPerRow + w0], b = row[(size_t) r * openWordsPerRow + w1];
                            open |= w0 == w1 ? (a & firstMask & lastMask) : ((a & firstMask) | (b & lastMask));
                        }
                }
                else open = 1u; // a larger box is kept unseen: 3 % of the splats, nearly all of them over some open tile, and one of them in
                                // a wave would keep the other 63 lanes waiting for its rows
                alive = open != 0;
            }
            const unsigned long long ballot = __ballot(alive);
            if ((threadIdx.x & (kWave - 1)) == 0 && splat < count) aliveWords[splat >> 6] = ballot;
        }
    }
}

// SCATTER = false: blockCounts[b] = live splats among order[b * 1024 ...), and every lane's four live bits in liveNibbles (the gather from the
// bit table is the expensive part: once is enough); true: the live ones written to out[] behind those of the blocks in front (every
// block sums the counts in front of it itself: a few thousand words), the last block leaves the total.
// Where a layer's slice of the depth order begins and ends when the list's length is only known on the device (a band's culled list):
// the frame driver's own rule (gs_capture.hip), cut by cut.
struct LayerSlice
{
    const int* listCountDevice; // null: the host cut the slice itself (order and count are the slice's)
    float cut[kMaxDepthLayers];
    int layer, layers;
};
__device__ __forceinline__ void layerSliceBounds(const LayerSlice& sl, int listCount, int& begin, int& end)
{
    begin = 0;
    end = listCount;
    for (int l = 0; l <= sl.layer; ++l)
    {
        end = l + 1 == sl.layers ? listCount : min(listCount, max(begin + 1, (int) ((double) sl.cut[l] * (double) listCount)));
        if (l != sl.layer) begin = end;
    }
    if (begin > listCount) begin = listCount;
}

template <bool SCATTER>
__global__ __launch_bounds__(kGroup) void layerCompactKernel(int countHost, const uint32_t* __restrict__ order, const unsigned long long* __restrict__ aliveWords,
                                                             int32_t* __restrict__ blockCounts, uint8_t* __restrict__ liveNibbles,
                                                             uint32_t* __restrict__ out, int32_t* __restrict__ countOut,
                                                             const int32_t* __restrict__ blockOffsets, LayerSlice slice)
{
    __shared__ int waveSums[kWavesPerBlock];
    __shared__ int baseSums[kWavesPerBlock];
    const int t = threadIdx.x, wave = t >> 6, lane = t & (kWave - 1);
    int count = countHost;
    if (slice.listCountDevice != nullptr)
    {
        // (a band's culled list: `order` is the whole list, countHost its capacity; the grid is sized for that)
        int begin, end;
        layerSliceBounds(slice, min(*slice.listCountDevice, countHost), begin, end);
        order += begin;
        count = end - begin;
    }
    const int activeBlocks = (count + kCompactBlock - 1) / kCompactBlock;
    if ((int) blockIdx.x >= activeBlocks)
    {
        if (!SCATTER && t == 0) blockCounts[blockIdx.x] = 0;
        if (SCATTER && activeBlocks == 0 && blockIdx.x == 0 && t == 0) *countOut = 0; // an empty slice: nobody else leaves the total
        return;
    }
    const int first = (int) blockIdx.x * kCompactBlock + t * kCompactItems;
    uint32_t ids[kCompactItems];
    unsigned live = 0;
#pragma unroll
    for (int k = 0; k != kCompactItems; ++k) ids[k] = first + k < count ? order[first + k] : 0u;
    if (SCATTER) live = liveNibbles[(size_t) blockIdx.x * kGroup + t];
    else
    {
#pragma unroll
        for (int k = 0; k != kCompactItems; ++k)
            if (first + k < count) live |= (aliveWords == nullptr ? 1u : (unsigned) ((aliveWords[ids[k] >> 6] >> (ids[k] & 63u)) & 1ull)) << k; // (null: a first layer)
        liveNibbles[(size_t) blockIdx.x * kGroup + t] = (uint8_t) live;
    }
    const int mine = __popc(live);
    const int incl = waveInclusiveSum(mine);
    if (lane == kWave - 1) waveSums[wave] = incl;
    if (SCATTER)
    {
        // the live splats of the blocks in front: summed here (a few thousand words), or -- slices of more than kCompactSelfSumBlocks
        // blocks, where that would be quadratic work -- read from a scan the host put in between
        int before = 0;
        if (blockOffsets != nullptr) before = t == 0 ? blockOffsets[blockIdx.x] : 0;
        else
            for (int b = t; b < (int) blockIdx.x; b += kGroup) before += blockCounts[b];
        const int beforeIncl = waveInclusiveSum(before);
        if (lane == kWave - 1) baseSums[wave] = beforeIncl;
    }
    __syncthreads();
    int inFront = 0, blockTotal = 0, base = 0;
#pragma unroll
    for (int wv = 0; wv != kWavesPerBlock; ++wv)
    {
        const int v = waveSums[wv];
        inFront += (wv < wave) ? v : 0;
        blockTotal += v;
        if (SCATTER) base += baseSums[wv];
    }
    if (!SCATTER)
    {
        if (t == 0) blockCounts[blockIdx.x] = blockTotal;
        return;
    }
    int at = base + inFront + incl - mine;
#pragma unroll
    for (int k = 0; k != kCompactItems; ++k)
        if (live & (1u << k)) out[at++] = ids[k];
    if (t == 0 && (int) blockIdx.x == activeBlocks - 1) *countOut = base + blockTotal;
}

// order[0 .. count): a layer's slice of the depth order; liveOrder[0 .. *liveCount): its splats that still have an open tile under their box.
void launchLayerCompaction(hipStream_t stream, const Screen& scr, int splats, int viewSplats, const ushort4* tileBox, const uint32_t* openBits, int openWordsPerRow, unsigned long long* aliveWords,
                           const uint32_t* order, int count, int32_t* blockCounts, uint8_t* liveNibbles, uint32_t* liveOrder, int32_t* liveCount,
                           int32_t* blockOffsets, int64_t* scanTotal, const int* listCountDevice, const float* cuts, int layer, int layers)
{
    const int views = viewSplats > 0 ? (splats + viewSplats - 1) / viewSplats : 1;
    const int tableWords = views * scr.tilesY * openWordsPerRow;
    if (count <= 0)
    {
        GS_CHECK(hipMemsetAsync(liveCount, 0, sizeof(int32_t), stream));
        return;
    }
    LayerSlice slice{};
    slice.listCountDevice = listCountDevice;
    for (int l = 0; l != kMaxDepthLayers; ++l) slice.cut[l] = cuts != nullptr ? cuts[l] : 1.0f;
    slice.layer = layer;
    slice.layers = layers;
    const bool everySplatLive = openBits == nullptr; // (a first layer: only cut out of a list whose length the host does not know)
    if (!everySplatLive)
    {
        if (tableWords <= kAliveTableWords)
            hipLaunchKernelGGL((layerAliveKernel<true>), dim3(std::min((splats + 4 * kGroup - 1) / (4 * kGroup), 2048)), dim3(kGroup), 0, stream, splats, scr, tileBox,
                               viewSplats, openBits, openWordsPerRow, tableWords, aliveWords);
        else
            hipLaunchKernelGGL((layerAliveKernel<false>), dim3(std::min((splats + 4 * kGroup - 1) / (4 * kGroup), 4096)), dim3(kGroup), 0, stream, splats, scr, tileBox,
                               viewSplats, openBits, openWordsPerRow, tableWords, aliveWords);
    }
    const unsigned long long* alive = everySplatLive ? nullptr : aliveWords;
    const int blocks = (count + kCompactBlock - 1) / kCompactBlock;
    hipLaunchKernelGGL((layerCompactKernel<false>), dim3(blocks), dim3(kGroup), 0, stream, count, order, alive, blockCounts, liveNibbles, liveOrder, liveCount,
                       (const int32_t*) nullptr, slice);
    const bool scanned = blocks > kCompactSelfSumBlocks;
    if (scanned) launchScanInt32(stream, blocks, blockCounts, blockOffsets, scanTotal);
    hipLaunchKernelGGL((layerCompactKernel<true>), dim3(blocks), dim3(kGroup), 0, stream, count, order, alive, blockCounts, liveNibbles, liveOrder, liveCount,
                       scanned ? (const int32_t*) blockOffsets : (const int32_t*) nullptr, slice);
}

int tileListFusedSortBits(int tileBitCount, int tileKeyBytes)
{
    if (tileKeyBytes != 2 || tileBitCount <= 9 || tileBitCount > 18) return 0; // one-pass sorts keep their pass (it fills the range table)
#ifndef GS_FUSED_FIRST_DIGIT_BIAS
#define GS_FUSED_FIRST_DIGIT_BIAS 0 // A/B builds: a wider (+1) or narrower (-1) first digit than the even split
#endif
    const int bits = (tileBitCount + 1) / 2 + GS_FUSED_FIRST_DIGIT_BIAS;        // the first digit of gs_radix.hip's two-pass split
    return (bits >= 6 && bits <= 8) ? bits : 0;
}

void launchTileListFastCount(hipStream_t stream, int count, const Screen& scr, const uint32_t* order, const float2* clipXY,
                             const float4* screenEllipse, TileListScratch& s, int sortBits, int viewSplats)
{
    ensureTileListFastScratch(s, count);
    const int groups = (count + kGroup - 1) / kGroup;
    if (groups == 0)
    {
        GS_CHECK(hipMemsetAsync(s.total, 0, sizeof(int64_t), stream));
    }
    else
    {
        // per-group totals live in the first ceil(N/256) entries of the per-wave array; the emit pass sums them itself and
        // leaves the pair total (no scan kernel in between)
        switch (sortBits)
        {
        case 6: launchCount<6>(stream, groups, count, scr, order, clipXY, screenEllipse, s, viewSplats); break;
        case 7: launchCount<7>(stream, groups, count, scr, order, clipXY, screenEllipse, s, viewSplats); break;
        case 8: launchCount<8>(stream, groups, count, scr, order, clipXY, screenEllipse, s, viewSplats); break;
        default: launchCount<0>(stream, groups, count, scr, order, clipXY, screenEllipse, s, 0); break;
        }
    }
    if (groups == 0) GS_CHECK(hipMemcpyAsync(s.totalHost, s.total, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
}

void launchTileListFastEmit(hipStream_t stream, int count, const Screen& scr, const uint32_t* order, const float2* clipXY,
                            const float4* screenEllipse, const TileListScratch& s, void* tileKeys, int tileKeyBytes, int32_t* values,
                            int capacity, int viewSplats, int sortBits)
{
    const int groups = (count + kGroup - 1) / kGroup;
    if (groups == 0) return;
    // (capacity 0 still runs: the pass is also what produces the pair total; nothing is written past `capacity`)
    const dim3 grid(groups), block(kGroup);
    if (sortBits > 0 && tileKeyBytes == 2)
    {
        // the count pass left table[digit][group]: scanned over the groups (and summed per digit) it is where every group's pairs of
        // every digit go; the emit pass then IS the first scatter of the tile sort
        launchRadixColumnScan(stream, 1 << sortBits, count, s.countDevice, kGroup, s.digitStride, s.digitTable, s.digitScanned, s.digitTotals);
#if GS_FUSED_XCD_CHUNK > 0
        const dim3 grid((groups + 8 * GS_FUSED_XCD_CHUNK - 1) / (8 * GS_FUSED_XCD_CHUNK) * (8 * GS_FUSED_XCD_CHUNK)); // whole chunks on every XCD
#endif
        switch (sortBits)
        {
        case 6: launchEmit<uint16_t, 6>(stream, grid, block, count, scr, order, clipXY, screenEllipse, s, nullptr, tileKeys, values, capacity, viewSplats); break;
        case 7: launchEmit<uint16_t, 7>(stream, grid, block, count, scr, order, clipXY, screenEllipse, s, nullptr, tileKeys, values, capacity, viewSplats); break;
        default: launchEmit<uint16_t, 8>(stream, grid, block, count, scr, order, clipXY, screenEllipse, s, nullptr, tileKeys, values, capacity, viewSplats); break;
        }
        if (s.totalHostMapped == nullptr) GS_CHECK(hipMemcpyAsync(s.totalHost, s.total, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
        return;
    }
    const int32_t* groupOffsets = nullptr;
    if (groups > (s.countDevice != nullptr ? kSelfSumGroupsCulled : kSelfSumGroups))
    {
        // long lists: exclusive scan of the group totals (and the pair total) by one workgroup, linear work
        hipLaunchKernelGGL(scanFastKernel, dim3(1), dim3(kScanThreads), 0, stream, groups, s.countDevice, kGroup, s.waveCounts, s.waveOffsets, s.total,
                           s.totalHostMapped);
        groupOffsets = s.waveOffsets;
    }
    if (tileKeyBytes == 2) launchEmit<uint16_t, 0>(stream, grid, block, count, scr, order, clipXY, screenEllipse, s, groupOffsets, tileKeys, values, capacity, viewSplats);
    else launchEmit<uint32_t, 0>(stream, grid, block, count, scr, order, clipXY, screenEllipse, s, groupOffsets, tileKeys, values, capacity, viewSplats);
    if (s.totalHostMapped == nullptr) GS_CHECK(hipMemcpyAsync(s.totalHost, s.total, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
}

} // namespace gs
