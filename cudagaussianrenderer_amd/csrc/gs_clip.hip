// gs_clip.hip -- per-splat world -> clip data: centre, depth, 3-sigma screen ellipse, conic.
//
// Replaces evaluateSplatClipDataKernel (GaussianRender.cu:204-329).  Pure streaming kernel:
// 32 B read + 44 B written per splat, one splat per lane with 16-byte coalesced loads/stores.
// The arithmetic follows the reference's order (covariance = (R S)(R S)^T, para-perspective
// Jacobian, J W Sigma^T (J W)^T, trace bump, eigen decomposition, conic = Sigma^-1) on plain floats;
// glm's 3x3 products are written out, entry by entry in glm's left-to-right order (type_mat3x3.inl:486-520), so the
// symmetric halves are not recomputed and the terms glm multiplies by a structural zero are left out (they only ever add +-0).
//
// Floating-point contraction is OFF in this file: every operation rounds as written, division and square root are
// IEEE-correct (no -ffast-math), so centre, depth, conic and extents are the same IEEE values the CPU oracle computes with
// glm's operand order (tests/test_gpu_parity.py: _check_clip compares them for equality).  The kernel is HBM-bound, the
// unfused multiplies and adds cost nothing measurable (profiles/r04_ab_experiments.txt: 23.7 -> 24.5 us of event time, inside the box-to-box noise).  It also makes the arithmetic
// independent of the context it is inlined into: the band kernel, the full-frame kernel and the all-views kernel give
// the same bits.
//
// One deliberate simplification: the reference stores (cos(atan2(e.y,e.x)), sin(atan2(e.y,e.x)))
// of the NORMALISED principal eigenvector e; that is e itself, which is what is stored here.
#include "gs_aabb.h"
#include "gs_depth_key.h"
#include "gs_internal.h"
#include "gs_radix_scan.h"
#include "gs_tile_order.h"

#pragma clang fp contract(off)

namespace gs
{
namespace
{

struct Sym3
{
    float xx, xy, xz, yy, yz, zz;
};

// What the kernel writes per (virtual) splat; null pointers are outputs nobody asked for.
struct ClipOutputs
{
    float2* clipXY;
    float* clipZ;
    float4* screenEllipse;
    float4* conic;
    uint32_t* depthKey;
    uint32_t* splatOrder;
    float4* tileRecord;
    float4* rasterRecord;
    ushort4* tileBox; // depth layers: the box the list builder will walk for the splat (gs_aabb.h splatTileBox), for the layers' live test
};

// World-space covariance of a splat, Sigma = (R S)(R S)^T (GaussianRender.cu:215-224): the camera-independent half of the projection.
__device__ __forceinline__ Sym3 splatCovariance(const float4 sr)
{
    // decodeVec4 (GaussianRender.cu:55-58,220-221): four truncated bytes -> [-1,1], x in the top byte;
    // the quaternion is NOT renormalised.
    const uint32_t bits = __float_as_uint(sr.w);
    const float qx = (float) ((bits >> 24) & 0xFFu) / 255.0f * 2.0f - 1.0f;
    const float qy = (float) ((bits >> 16) & 0xFFu) / 255.0f * 2.0f - 1.0f;
    const float qz = (float) ((bits >> 8) & 0xFFu) / 255.0f * 2.0f - 1.0f;
    const float qw = (float) (bits & 0xFFu) / 255.0f * 2.0f - 1.0f;

    // R = mat3_cast(q) (glm/gtc/quaternion.inl:47-72); r{col}{row}
    const float qxx = qx * qx, qyy = qy * qy, qzz = qz * qz;
    const float qxz = qx * qz, qxy = qx * qy, qyz = qy * qz;
    const float qwx = qw * qx, qwy = qw * qy, qwz = qw * qz;
    const float r00 = 1.0f - 2.0f * (qyy + qzz), r01 = 2.0f * (qxy + qwz), r02 = 2.0f * (qxz - qwy);
    const float r10 = 2.0f * (qxy - qwz), r11 = 1.0f - 2.0f * (qxx + qzz), r12 = 2.0f * (qyz + qwx);
    const float r20 = 2.0f * (qxz + qwy), r21 = 2.0f * (qyz - qwx), r22 = 1.0f - 2.0f * (qxx + qyy);

    // RS = R * diag(s): column c scaled by s_c.  Sigma = RS * RS^T (GaussianRender.cu:223-224), symmetric.
    const float m00 = r00 * sr.x, m01 = r01 * sr.x, m02 = r02 * sr.x;
    const float m10 = r10 * sr.y, m11 = r11 * sr.y, m12 = r12 * sr.y;
    const float m20 = r20 * sr.z, m21 = r21 * sr.z, m22 = r22 * sr.z;
    Sym3 S;
    S.xx = m00 * m00 + m10 * m10 + m20 * m20;
    S.xy = m00 * m01 + m10 * m11 + m20 * m21;
    S.xz = m00 * m02 + m10 * m12 + m20 * m22;
    S.yy = m01 * m01 + m11 * m11 + m21 * m21;
    S.yz = m01 * m02 + m11 * m12 + m21 * m22;
    S.zz = m02 * m02 + m12 * m12 + m22 * m22;
    return S;
}

// The camera-dependent half: projection of the covariance, eigen decomposition, conic, frustum cull, (band test,) stores.
// `out`: the (virtual) storage slot the records are stored under; `zOut`: where the clip depth goes (= out, except on the band-cull
// path of a renderer with Morton storage, which walks the splats in the CALLER's order and compacts them in it).
// Returns whether the splat reaches the band (always true without a band test).  No contraction in this file: the two halves give the
// same bits whether a kernel runs them back to back for one camera or keeps the covariance for several.
template <bool BAND>
__device__ __forceinline__ bool projectCovariance(const float4 pd, const Sym3 S, const ClipCamera& cam, const Screen& scr, int out, int zOut,
                                                  const ClipOutputs& o)
{
    const float* v = cam.view;
    const ViewTerms t = viewTerms(pd, cam);
    const float scaleX = t.scaleX, scaleY = t.scaleY, shearX = t.shearX, shearY = t.shearY;

    // Rows 0 and 1 of VP = mat3(J) * mat3(view) (GaussianRender.cu:262); row 2 never reaches the 2x2 block we need.
    //   J row0 = (scaleX, 0, shearX), row1 = (0, scaleY, shearY)
    const float a0 = scaleX * v[0] + shearX * v[2], a1 = scaleX * v[4] + shearX * v[6], a2 = scaleX * v[8] + shearX * v[10];
    const float b0 = scaleY * v[1] + shearY * v[2], b1 = scaleY * v[5] + shearY * v[6], b2 = scaleY * v[9] + shearY * v[10];

    // Clip covariance 2x2 block = rows(a,b) * Sigma * rows(a,b)^T (GaussianRender.cu:264).
    const float sa0 = S.xx * a0 + S.xy * a1 + S.xz * a2;
    const float sa1 = S.xy * a0 + S.yy * a1 + S.yz * a2;
    const float sa2 = S.xz * a0 + S.yz * a1 + S.zz * a2;
    const float sb0 = S.xx * b0 + S.xy * b1 + S.xz * b2;
    const float sb1 = S.xy * b0 + S.yy * b1 + S.yz * b2;
    const float sb2 = S.xz * b0 + S.yz * b1 + S.zz * b2;
    float covA = a0 * sa0 + a1 * sa1 + a2 * sa2; // [0][0]
    const float covB = b0 * sa0 + b1 * sa1 + b2 * sa2; // [1][0]
    float covC = b0 * sb0 + b1 * sb1 + b2 * sb2; // [1][1]

    float clipX = t.clipX, clipY = t.clipY, clipDepth = t.clipDepth;

    // Trace bump (GaussianRender.cu:271-276).
    covA += scr.traceBump;
    covC += scr.traceBump;

    // Eigen decomposition (GaussianRender.cu:279-302).
    const float det = covA * covC - covB * covB;
    const float mid = 0.5f * (covA + covC);
    constexpr float epsilon = 1e-12f;
    const float radius = sqrtf(fmaxf(epsilon, mid * mid - det));
    const float lambda0 = mid + radius;
    const float l1 = mid - radius;
    const float lambda1 = (0.0f < l1) ? l1 : 0.0f; // glm::max(0, x): NaN -> 0
    const float ex = covB, ey = lambda0 - covA;
    const float invLen = 1.0f / sqrtf(ex * ex + ey * ey);
    const float cosA = ex * invLen, sinA = ey * invLen;
    float extentX = sqrtf(lambda0) * scr.extentFactor; // GaussianRender.cu:298-302
    float extentY = sqrtf(lambda1) * scr.extentFactor;

    // Conic (GaussianRender.cu:305-307).
    const float invDet = 1.0f / fmaxf(epsilon, det);

    // Centre-in-frustum cull (GaussianRender.cu:316-321), see viewTerms; lambda1 >= 0 by construction.
    const bool inside = t.inside && !(lambda1 < 0.0f);
    const float isVisible = inside ? 1.0f : 0.0f;
    // mix(-128, clip, v) = -128*(1-v) + clip*v ; extent *= v  (kept as arithmetic so inf/NaN propagate as in the reference)
    clipX = mixInvisible(clipX, isVisible);
    clipY = mixInvisible(clipY, isVisible);
    clipDepth = mixInvisible(clipDepth, isVisible);
    extentX *= isVisible;
    extentY *= isVisible;

    if (BAND)
    {
        // Band cull for the multi-GPU split: does this splat's tile AABB reach the tile rows this GPU owns?  Same AABB
        // routine as the tile-list builder, so nothing that would emit a pair is dropped.
        const TileAabb box = splatTileAabb(clipX, clipY, cosA, sinA, extentX, extentY, scr);
        // Nothing downstream ever looks at a splat that does not reach the band (the compacted list holds the kept ones only):
        // its per-splat outputs are not written.
        if (!(box.x1 > box.x0 && box.y1 > box.y0)) return false;
    }
    if (o.depthKey != nullptr)
    {
        // Quantised depth of getKey() (GaussianRender.cu:446-454) as a sort key of its own (27-bit code, see above), plus
        // the identity permutation: the one-call renderer sorts SPLATS by depth before expanding them to tiles.  (A renderer with
        // Morton storage hands the sort its own input order -- caller index -> slot -- instead: no identity list, gs_radix.hip.)
        o.depthKey[out] = compactDepthKey(clipDepth);
        if (o.splatOrder != nullptr) o.splatOrder[out] = (uint32_t) out;
    }
    if (o.clipXY != nullptr) o.clipXY[out] = make_float2(clipX, clipY);
    if (o.clipZ != nullptr) o.clipZ[zOut] = clipDepth;
    if (o.screenEllipse != nullptr) o.screenEllipse[out] = make_float4(cosA, sinA, extentX, extentY);
    if (o.tileRecord != nullptr)
    {
        // the same six numbers once more as ONE 32-byte record: the tile-list passes fetch them through the depth order
        // (random access), where two separate arrays cost two memory sectors per splat
        o.tileRecord[2 * (size_t) out] = make_float4(clipX, clipY, cosA, sinA);
        o.tileRecord[2 * (size_t) out + 1] = make_float4(extentX, extentY, 0.0f, 0.0f);
    }
    if (o.tileBox != nullptr)
    {
        // (from the six numbers of the tile record, by the builder's own routine: the same box, bit for bit)
        const TileAabb b = splatTileBox(clipX, clipY, cosA, sinA, extentX, extentY, scr);
        o.tileBox[out] = make_ushort4((unsigned short) b.x0, (unsigned short) b.y0, (unsigned short) b.x1, (unsigned short) b.y1);
    }
    if (o.conic != nullptr) o.conic[out] = make_float4(covC * invDet, -covB * invDet, covA * invDet, 0.0f);
    if (o.rasterRecord != nullptr)
    {
        // what the raster kernel gathers per list entry besides the colour, as ONE 32-byte record (one memory sector instead of one
        // for the centre and one for the conic)
        o.rasterRecord[2 * (size_t) out] = make_float4(clipX, clipY, covC * invDet, -covB * invDet);
        o.rasterRecord[2 * (size_t) out + 1] = make_float4(covA * invDet, clipDepth, 0.0f, 0.0f); // (the depth rides along for gs_renderer_buffers)
    }
    return true;
}

template <bool BAND>
__device__ __forceinline__ bool projectSplat(const float4 pd, const float4 sr, const ClipCamera& cam, const Screen& scr, int out, int zOut,
                                             const ClipOutputs& o)
{
    return projectCovariance<BAND>(pd, splatCovariance(sr), cam, scr, out, zOut, o);
}

__device__ __forceinline__ ClipCamera cam_or(const FrameParams* __restrict__ params, int view, const ClipCamera& camArg)
{
    return params ? params[view].camera : camArg; // device-resident when the frame is a captured graph
}

// Upper bound on the largest eigenvalue of a splat's world-space covariance Sigma = (R S)(R S)' (GaussianRender.cu:215-224):
// lmax = smax(R S)^2 <= (|R| max|s|)^2, and for the reference's un-normalised quaternion q = t u (|u| = 1) mat3_cast gives
// R = (1 - t^2) I + t^2 R_u, so |R| <= |1 - t^2| + t^2 (1 for a unit quaternion).  Padded for the rounding of the kernel's own Sigma.
__device__ __forceinline__ float covarianceEigenBound(const float4 sr)
{
    const uint32_t bits = __float_as_uint(sr.w);
    const float qx = (float) ((bits >> 24) & 0xFFu) / 255.0f * 2.0f - 1.0f;
    const float qy = (float) ((bits >> 16) & 0xFFu) / 255.0f * 2.0f - 1.0f;
    const float qz = (float) ((bits >> 8) & 0xFFu) / 255.0f * 2.0f - 1.0f;
    const float qw = (float) (bits & 0xFFu) / 255.0f * 2.0f - 1.0f;
    const float t2 = (qx * qx + qy * qy) + (qz * qz + qw * qw);
    const float rn = fabsf(1.0f - t2) + t2;
    const float smax = fmaxf(fabsf(sr.x), fmaxf(fabsf(sr.y), fabsf(sr.z))); // (a NaN scale makes a NaN ellipse, which reaches no tile: ignored)
    const float m = smax * rn;
    return m * m * 1.0002f;
}

// What the bound needs of a camera besides its matrix: |W|^2 (largest Gershgorin row sum of W W'; 1 for a rigid camera) and the
// view-space z range of the visible clip depths [-1, 1] (GaussianRender.cu:316-318), padded.  Per view and frame: the block test
// computes them and leaves them for the band kernel's per-splat test.
__device__ __forceinline__ BandViewTerms makeBandViewTerms(const ClipCamera& cam)
{
    const float* v = cam.view;
    BandViewTerms t;
    float wn = 0.0f;
#pragma unroll
    for (int r = 0; r != 3; ++r)
    {
        float row = 0.0f;
#pragma unroll
        for (int s = 0; s != 3; ++s) row += fabsf(v[0 + r] * v[0 + s] + v[4 + r] * v[4 + s] + v[8 + r] * v[8 + s]);
        wn = fmaxf(wn, row);
    }
    t.wn = wn * 1.001f;
    const float zA = (-1.0f - cam.depthBias) / cam.depthScale, zB = (1.0f - cam.depthBias) / cam.depthScale;
    const float zLo = fminf(zA, zB), zHi = fmaxf(zA, zB);
    t.zvLo = zLo - 1.0e-4f * fabsf(zLo);
    t.zvHi = zHi + 1.0e-4f * fabsf(zHi);
    t.sane = (cam.fovCotX > 0.0f && cam.fovCotY > 0.0f && t.zvHi < 0.0f) ? 1.0f : 0.0f; // (a camera the bound was not written for: everything is kept)
    return t;
}

// true = NO splat with its centre in the box [cmin, cmax] and its covariance's largest eigenvalue <= lam can reach the band's tile rows.
// cmin == cmax: the test of ONE splat.  Every rejection is a comparison that a NaN fails.
__device__ __forceinline__ bool bandBoundRejects(const float cmin[3], const float cmax[3], float lam, const BandViewTerms& vt, const ClipCamera& cam,
                                                 const Screen& scr)
{
    if (!(vt.sane > 0.5f)) return false;
    const float* v = cam.view;
    // view-space box of the centre box (interval arithmetic per output row, padded by the rounding of its four terms)
    float lo[3], hi[3];
#pragma unroll
    for (int r = 0; r != 3; ++r)
    {
        float l = v[12 + r], h = v[12 + r], mag = fabsf(v[12 + r]);
#pragma unroll
        for (int c = 0; c != 3; ++c)
        {
            const float p0 = v[c * 4 + r] * cmin[c], p1 = v[c * 4 + r] * cmax[c];
            l += fminf(p0, p1);
            h += fmaxf(p0, p1);
            mag += fmaxf(fabsf(p0), fabsf(p1));
        }
        lo[r] = l - 1.0e-5f * mag;
        hi[r] = h + 1.0e-5f * mag;
    }
    const float zLo = fmaxf(lo[2], vt.zvLo), zHi = fminf(hi[2], vt.zvHi);
    if (zLo > zHi) return true; // no centre of the box inside the visible depth range: nothing of it is drawn
    const float dn = -zHi, df = -zLo; // distances in front of the camera, nearest and farthest (dn > 0: vt.sane)
    if (!(dn > 0.0f)) return false;
    const float rn = 1.0f / dn, rf = 1.0f / df;
    // centre extremes in clip space, cut to the frustum (a visible centre has |clip x|, |clip y| <= 1)
    const float cyMaxRaw = cam.fovCotY * hi[1] * (hi[1] >= 0.0f ? rn : rf), cyMinRaw = cam.fovCotY * lo[1] * (lo[1] <= 0.0f ? rn : rf);
    const float cxMaxRaw = cam.fovCotX * hi[0] * (hi[0] >= 0.0f ? rn : rf), cxMinRaw = cam.fovCotX * lo[0] * (lo[0] <= 0.0f ? rn : rf);
    if (cyMinRaw > 1.002f || cyMaxRaw < -1.002f || cxMinRaw > 1.002f || cxMaxRaw < -1.002f) return true; // every centre outside the frustum
    const float cyMax = fminf(cyMaxRaw, 1.002f), cyMin = fmaxf(cyMinRaw, -1.002f);
    const float cxMax = fminf(cxMaxRaw, 1.002f), cxMin = fmaxf(cxMinRaw, -1.002f);
    // |J row 0|^2 + |J row 1|^2 = (cotX^2 + cx^2 + cotY^2 + cy^2) / d^2 with the centre's own clip coordinates in the shear terms
    const float cx2 = fmaxf(cxMin * cxMin, cxMax * cxMax), cy2 = fmaxf(cyMin * cyMin, cyMax * cyMax);
    const float jacobian = ((cam.fovCotX * cam.fovCotX + cx2) + (cam.fovCotY * cam.fovCotY + cy2)) * 1.004f * (rn * rn);
    // (l0 + l1 = trace + 2 bump -- unless the reference's floor under the eigenvalues' radius is in force (GaussianRender.cu:284:
    // sqrt(max(1e-12, mid^2 - det)) >= 1e-6) AND exceeds mid, which clamps l1 to 0: then l0 + l1 = mid + 1e-6.  That is every sub-pixel
    // splat once the bump (1/pi)(2/W)(2/H) falls below 1e-6, i.e. from about 1600 x 900 on: its extent is 3 sqrt(bump + 1e-6), not
    // 3 sqrt(2 bump).  tools/r06_fuzz.py found it: one splat of 8 139, 0.0006 clip units from a band's edge at 1944 x 1649.)
    const float half = 1.01f * scr.extentFactor * __builtin_sqrtf(lam * vt.wn * jacobian + 2.0f * scr.traceBump + 2.0e-6f) + 1.0e-5f;
    // rows, as splatTileAabb takes them: kept iff (maxY + 1) / 2 * tilesPerScreenY > rowBegin and (minY + 1) / 2 * tilesPerScreenY < rowEnd
    const float top = ((cyMax + half) + 1.0f) * 0.5f * scr.tilesPerScreenY, bottom = ((cyMin - half) + 1.0f) * 0.5f * scr.tilesPerScreenY;
    return top < (float) scr.rowBegin - 1.0e-3f || bottom > (float) scr.rowEnd + 1.0e-3f;
}

constexpr int kBandBlock = 256;                // splats per workgroup of the band-cull kernel; the compaction's scan runs over these blocks
constexpr int kBandSlots = kBandBlock / kWave; // 64-splat slots (one kept mask each) per workgroup

// Every splat projected and stored, one per lane (the single-GPU frame, the staged API): workgroup `block` of `blocks`, grid row `view`
// of `views`.
__device__ __forceinline__ void clipDataBlock(int block, int blocks, int view, int views, int count, const Screen& scr, const ClipCamera& camArg,
                                              const FrameParams* __restrict__ params, const float4* __restrict__ position,
                                              const float4* __restrict__ scaleAndRotation, const ClipOutputs& o, int32_t* __restrict__ rangeTable,
                                              int rangeEntries)
{
    const int index = block * 256 + (int) threadIdx.x; // splat of the scene
    // The renderer has this kernel reset the tile range table to -1 (Demo.cpp:403) on the side: nothing reads the table before the
    // range search (the last scatter of the tile sort), so the frame needs no separate clear.
    if (rangeTable != nullptr)
        for (int i = view * (blocks * 256) + index; i < rangeEntries; i += blocks * views * 256) rangeTable[i] = -1;
    if (index >= count) return;
    const ClipCamera cam = params ? params[view].camera : camArg; // device-resident when the frame is a captured graph
    // results are stored under the "virtual splat" view * count + index
    (void) projectSplat<false>(position[index], scaleAndRotation[index], cam, scr, view * count + index, view * count + index, o);
}

__global__ __launch_bounds__(256) void splatClipDataKernel(int count, Screen scr, ClipCamera camArg, const FrameParams* __restrict__ params,
                                                            const float4* __restrict__ position,
                                                            const float4* __restrict__ scaleAndRotation, ClipOutputs o,
                                                            int32_t* __restrict__ rangeTable, int rangeEntries, TileOrderArgs tileOrder)
{
    // gs_renderer_set_tile_start_order: the frame's tile start order is made by the first eight workgroups of this launch (one per XCD;
    // they are dispatched first and are done long before the clip stage's streaming ends) -- no kernel of its own, nothing on the
    // frame's critical path
    const int orderBlocks = tileOrder.slots > 0 ? kOrderWorkgroups : 0;
    if ((int) blockIdx.x < orderBlocks) // (uniform per workgroup)
    {
        if (blockIdx.y == 0) tileStartOrderBody<256>((int) blockIdx.x, scr, tileOrder);
        return;
    }
    // gs_renderer_set_views: one grid row per camera
    clipDataBlock((int) blockIdx.x - orderBlocks, (int) gridDim.x - orderBlocks, (int) blockIdx.y, (int) gridDim.y, count, scr, camArg, params, position,
                  scaleAndRotation, o, rangeTable, rangeEntries);
}

// The same, sharing its launch with the FIRST column scan of the depth sort (gs_internal.h RadixScanCompanion).  A renderer with Morton
// storage has the depth sort make its own keys, so nothing of the sort waits for the clip data and the clip kernel needs nothing of the
// sort: the scan -- 512 workgroups that each add up one row of a 2 MB table, 5 us of pure latency -- rides in front of the clip kernel's
// grid (its workgroups start first and are gone long before the clip stage's 22 us of streaming end).  One kernel, one dependent launch
// and one chip-wide kernel boundary less per frame; the results are the two kernels' results (same routines, same order of operations).
// Round 6: the clip kernel's workgroups can be dealt into slices, one per scan launch of the sort (kClipScanParts: measured, one slice is
// what pays) -- the grid of a launch is [scan rows | the tile start order's eight builders (first slice only) | clip workgroups
// firstBlock ..]; a clip workgroup does what it does in the plain kernel (its index and the total are the whole kernel's, so the range
// table is cleared once).
__global__ __launch_bounds__(256) void splatClipDataScanKernel(RadixScanArgs scan, int count, Screen scr, ClipCamera camArg,
                                                                const FrameParams* __restrict__ params, const float4* __restrict__ position,
                                                                const float4* __restrict__ scaleAndRotation, ClipOutputs o,
                                                                int32_t* __restrict__ rangeTable, int rangeEntries, int firstBlock, int clipBlocks,
                                                                TileOrderArgs tileOrder)
{
    if ((int) blockIdx.x < scan.rows) // (uniform per workgroup)
    {
        if (blockIdx.y == 0) radixColumnScanRow<256>((int) blockIdx.x, scan.nHost, scan.nDevice, scan.tileElements, scan.stride, scan.table, scan.scanned, scan.totals);
        return;
    }
    const int orderBlocks = tileOrder.slots > 0 ? kOrderWorkgroups : 0;
    if ((int) blockIdx.x < scan.rows + orderBlocks)
    {
        if (blockIdx.y == 0) tileStartOrderBody<256>((int) blockIdx.x - scan.rows, scr, tileOrder);
        return;
    }
    clipDataBlock(firstBlock + (int) blockIdx.x - scan.rows - orderBlocks, clipBlocks, (int) blockIdx.y, (int) gridDim.y, count, scr, camArg, params, position,
                  scaleAndRotation, o, rangeTable, rangeEntries);
}

// Multi-GPU band cull: the same projection, but a splat whose tile AABB does not reach this GPU's tile rows stores nothing.
// Leaves, per slot of 64 splats, the 64-bit mask of the kept ones and the number of kept splats in the workgroup's earlier slots;
// per workgroup, its number of kept splats (the compaction scans these).
// Morton storage (outSlot != nullptr): the kernel walks the splats in the CALLER's order -- position and scaleAndRotation are then the
// caller-ordered copies, masks, counts and clip depths are indexed by the caller's index, and only the two 32-byte records of a kept
// splat go to its storage slot outSlot[index] (whole memory sectors).  The compaction therefore runs in the caller's order without
// gathering anything, the stable depth sort keeps splats of equal depth code in it, and the frame is the one a renderer with the
// caller's storage order gives -- for any number of equal codes (round 3 repaired the order after the sort, up to runs of 1024).
// (Tried and dropped: a cheap conservative bound on the AABB's height first -- trace(Sigma) |J W|_F^2 -- with the candidates of a
// 256- or 1024-splat workgroup compacted in LDS and only those projected.  24 % of the splats of a centre band of the 8-way split
// are candidates, the projection work falls to a third -- and the kernel takes as long as before: 48.4 - 58.6 us for a 4-view chain
// against 47.1 us for this one; loads alone are 20 us of it, and the compacted second phase runs at low occupancy.)
__global__ __launch_bounds__(256) void splatClipBandKernel(int count, Screen scr, ClipCamera camArg, const FrameParams* __restrict__ params,
                                                            const float4* __restrict__ position,
                                                            const float4* __restrict__ scaleAndRotation, ClipOutputs o,
                                                            unsigned long long* __restrict__ bandMask, int32_t* __restrict__ bandSlotPrefix,
                                                            int32_t* __restrict__ bandBlockCounts, int32_t* __restrict__ rangeTable,
                                                            int rangeEntries, const int32_t* __restrict__ outSlot,
                                                            const uint8_t* __restrict__ blockFlags, const BandViewTerms* __restrict__ viewTerms,
                                                            const int32_t* __restrict__ storageToSplat)
{
    __shared__ int slotKept[kBandSlots];
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
    const int index = blockIdx.x * blockDim.x + t;
    const int view = blockIdx.y;
    if (rangeTable != nullptr)
        for (int i = view * (int) (gridDim.x * blockDim.x) + index; i < rangeEntries; i += (int) (gridDim.x * gridDim.y * blockDim.x))
            rangeTable[i] = -1;
    const int blockBase = blockIdx.x * kBandBlock;
    if (blockBase >= count) return; // whole workgroup
    const int slotsPerView = (count + kWave - 1) / kWave, blocksPerView = (count + kBandBlock - 1) / kBandBlock;
    if (blockFlags != nullptr)
    {
        // Block cull (see blockBandTestKernel): the splats are walked in STORAGE order -- along the Morton curve -- and a block of 256
        // neighbours that provably cannot reach the band is not projected at all.  A kept splat sets ITS bit in the caller-ordered keep
        // mask (one fire-and-forget 64-bit atomic OR through storageToSplat; the block test zeroed the mask): the compaction runs in the
        // caller's order, over masks nobody had to gather.
        const bool live = blockFlags[view * blocksPerView + blockIdx.x] != 0; // (uniform)
        if (!live) return;
        // ... and inside a kept block every splat is held to the same bound by itself (its own centre, its own covariance): a wave
        // of 64 Morton neighbours none of which can reach the band skips the projection; the others project their candidates
        const ClipCamera cam = cam_or(params, view, camArg);
        float4 pd = make_float4(0.f, 0.f, 0.f, 0.f), sr = pd;
        bool candidate = false;
        if (index < count)
        {
            pd = position[index];
            sr = scaleAndRotation[index];
            const float c[3] = {pd.x, pd.y, pd.z};
            candidate = !bandBoundRejects(c, c, covarianceEigenBound(sr), viewTerms[view], cam, scr);
        }
        if (__ballot(candidate) == 0ull) return;
        if (candidate && projectSplat<true>(pd, sr, cam, scr, view * count + index, view * count + index, o))
        {
            const int caller = storageToSplat[index];
            atomicOr(&bandMask[view * slotsPerView + (caller >> 6)], 1ull << (caller & 63));
        }
        return;
    }
    const ClipCamera cam = params ? params[view].camera : camArg;
    bool kept = false;
    if (index < count)
        kept = projectSplat<true>(position[index], scaleAndRotation[index], cam, scr, view * count + (outSlot != nullptr ? outSlot[index] : index),
                                  view * count + index, o);
    const unsigned long long mask = __ballot(kept);
    if (lane == 0) slotKept[wave] = __popcll(mask);
    __syncthreads();
    if (lane == 0 && blockBase + wave * kWave < count)
    {
        int prefix = 0;
        for (int s = 0; s < wave; ++s) prefix += slotKept[s];
        bandMask[view * slotsPerView + (blockBase >> 6) + wave] = mask;
        bandSlotPrefix[view * slotsPerView + (blockBase >> 6) + wave] = prefix;
        if (wave == 0) bandBlockCounts[view * blocksPerView + blockIdx.x] = slotKept[0] + slotKept[1] + slotKept[2] + slotKept[3];
    }
}

// The band kernel for a chain of several views (gs_renderer_set_views): ONE lane per splat walks all the chain's cameras -- position and
// scale/rotation are read once and the world-space covariance is computed once, only the camera-dependent half runs per view.  Same
// outputs as `views` grid rows of the kernel above, bit for bit (no contraction in this file).  Round 3 measured this shape at
// 45.9 against 50.2 us at V = 4 and had to drop it: with contraction on, the compiler fused the inlined arithmetic differently in
// the two contexts and half of the conics lost their last bit (profiles/r04_clip_all_views.txt has this round's figures).
__global__ __launch_bounds__(256) void splatClipBandViewsKernel(int count, int views, Screen scr, const FrameParams* __restrict__ params,
                                                                 const float4* __restrict__ position,
                                                                 const float4* __restrict__ scaleAndRotation, ClipOutputs o,
                                                                 unsigned long long* __restrict__ bandMask, int32_t* __restrict__ bandSlotPrefix,
                                                                 int32_t* __restrict__ bandBlockCounts, int32_t* __restrict__ rangeTable,
                                                                 int rangeEntries, const int32_t* __restrict__ outSlot,
                                                                 const uint8_t* __restrict__ blockFlags, const BandViewTerms* __restrict__ viewTerms,
                                                                 const int32_t* __restrict__ storageToSplat)
{
    __shared__ int slotKept[kMaxViews][kBandSlots];
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
    const int index = blockIdx.x * blockDim.x + t;
    if (rangeTable != nullptr)
        for (int i = index; i < rangeEntries; i += (int) (gridDim.x * blockDim.x)) rangeTable[i] = -1;
    const int blockBase = blockIdx.x * kBandBlock;
    if (blockBase >= count) return; // whole workgroup
    if (blockFlags != nullptr)
    {
        // Block cull (see the single-view kernel): storage order, only the views whose test kept this block are looked at; a kept splat
        // sets its bit in its view's caller-ordered keep mask
        const int slotsPerView = (count + kWave - 1) / kWave, blocksPerView = (count + kBandBlock - 1) / kBandBlock;
        bool any = false;
        bool live[kMaxViews];
#pragma unroll
        for (int view = 0; view != kMaxViews; ++view)
        {
            live[view] = view < views && blockFlags[view * blocksPerView + blockIdx.x] != 0; // (uniform)
            any |= live[view];
        }
        if (!any) return;
        // (inside a kept block every splat is held to the same bound by itself: a wave none of whose splats can reach the band in a
        // view skips that view's projection; the world-space covariance is only made by a wave that projects in some view)
        float4 pd = make_float4(0.f, 0.f, 0.f, 0.f), sr = pd;
        float lam = 0.0f;
        if (index < count)
        {
            pd = position[index];
            sr = scaleAndRotation[index];
            lam = covarianceEigenBound(sr);
        }
        const float c[3] = {pd.x, pd.y, pd.z};
        bool candidate[kMaxViews];
        bool anyCandidate = false;
#pragma unroll
        for (int view = 0; view != kMaxViews; ++view)
        {
            candidate[view] = live[view] && index < count && !bandBoundRejects(c, c, lam, viewTerms[view], params[view].camera, scr);
            anyCandidate |= candidate[view];
        }
        if (__ballot(anyCandidate) == 0ull) return;
        const Sym3 S = splatCovariance(sr);
        const int caller = index < count ? storageToSplat[index] : 0;
#pragma unroll
        for (int view = 0; view != kMaxViews; ++view)
        {
            if (view < views && __ballot(candidate[view]) != 0ull)
            {
                if (candidate[view] && projectCovariance<true>(pd, S, params[view].camera, scr, view * count + index, view * count + index, o))
                    atomicOr(&bandMask[view * slotsPerView + (caller >> 6)], 1ull << (caller & 63));
            }
        }
        return;
    }
    float4 pd = make_float4(0.f, 0.f, 0.f, 0.f);
    Sym3 S{};
    int slot = index;
    if (index < count)
    {
        pd = position[index];
        S = splatCovariance(scaleAndRotation[index]);
        if (outSlot != nullptr) slot = outSlot[index];
    }
    unsigned long long masks[kMaxViews];
#pragma unroll
    for (int view = 0; view != kMaxViews; ++view)
    {
        masks[view] = 0ull;
        if (view < views) // (uniform)
        {
            bool kept = false;
            if (index < count) kept = projectCovariance<true>(pd, S, params[view].camera, scr, view * count + slot, view * count + index, o);
            masks[view] = __ballot(kept);
            if (lane == 0) slotKept[view][wave] = __popcll(masks[view]);
        }
    }
    __syncthreads();
    if (lane == 0 && blockBase + wave * kWave < count)
    {
        const int slotsPerView = (count + kWave - 1) / kWave, blocksPerView = (count + kBandBlock - 1) / kBandBlock;
#pragma unroll
        for (int view = 0; view != kMaxViews; ++view)
        {
            if (view < views)
            {
                int prefix = 0;
                for (int s = 0; s < wave; ++s) prefix += slotKept[view][s];
                bandMask[view * slotsPerView + (blockBase >> 6) + wave] = masks[view];
                bandSlotPrefix[view * slotsPerView + (blockBase >> 6) + wave] = prefix;
                if (wave == 0)
                    bandBlockCounts[view * blocksPerView + blockIdx.x] = slotKept[view][0] + slotKept[view][1] + slotKept[view][2] + slotKept[view][3];
            }
        }
    }
}

// ---- Block-level band cull (round 6) --------------------------------------------------------------------------------------------
// Every rank of a multi-GPU split used to run the projection over ALL N splats to learn which ones reach its rows (the replicated
// part of a band frame: 90 of 250 us at BASELINE.json configs[4] / 8).  With the scene stored along the Morton curve a block of 256
// consecutive splats is a small box in space, and one cheap test per block and frame can prove that none of its splats reaches the
// band: the per-splat test then only runs over the blocks that remain.  The test only has to be CONSERVATIVE -- the per-splat keep
// mask (projectCovariance<true>, the list builder's own box routine) stays the decider inside every kept block, so the frame is the
// same bytes -- and it is a bound on exactly that routine (GaussianRender.cu:227-302,316-321,408-436,537-551):
//   * a splat whose centre is outside the frustum emits nothing (extent 0), so only the part of the block's box with clip depth in
//     [-1, 1] counts: d = -view z in [dn, df] (a block that straddles the near plane is cut at it, not kept wholesale);
//   * its centre row: clip y = cotY * vy / d lies between the extremes of the box's view-space y over [dn, df], and in [-1, 1];
//   * the reference's box is the bounding box of the ellipse's ORIENTED RECTANGLE: half height |sin| ex + |cos| ey <= k sqrt(l0 + l1)
//     <= k sqrt(trace(C) + 2 bump + 1e-6) (the 1e-6: the floor under the eigenvalues' radius, see bandBoundRejects), and trace(C) = a' S a + b' S b <= lmax(S) |W|^2 (|J row 0|^2 + |J row 1|^2)
//     <= lmax(S) |W|^2 (cotX^2 + cotY^2 + 2) / dn^2 for a visible centre (|clip x|, |clip y| <= 1 bound the shear terms);
//   * lmax(S) per splat by Gershgorin on the very covariance the kernel computes (un-normalised quaternions included), its maximum
//     per block stored at set_scene; |W|^2 by Gershgorin on W W' (1 for a rigid camera).
// Every float evaluation is padded (1 % on the extent, 1e-3 on the unit bounds, 1e-3 tiles on the row compare) and every rejection is
// a comparison that a NaN fails: a block the bound cannot speak for is kept.
// set_scene: centre box and largest covariance eigenvalue bound of every block of 256 splats in storage order.
__global__ __launch_bounds__(256) void blockBoundsKernel(int count, const float4* __restrict__ position, const float4* __restrict__ scaleAndRotation,
                                                          BlockBound* __restrict__ bounds)
{
    __shared__ float red[7][4];
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
    const int index = blockIdx.x * kBandBlock + t;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, lam = 0.0f;
    if (index < count)
    {
        const float4 pd = position[index];
        lo[0] = hi[0] = pd.x;
        lo[1] = hi[1] = pd.y;
        lo[2] = hi[2] = pd.z;
        lam = covarianceEigenBound(scaleAndRotation[index]);
    }
    // (a NaN bound must survive the reduction: the block is then kept by every test)
    const auto maxNan = [](float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b); };
#pragma unroll
    for (int off = 32; off != 0; off >>= 1)
    {
#pragma unroll
        for (int k = 0; k != 3; ++k)
        {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
        lam = maxNan(lam, __shfl_xor(lam, off));
    }
    if (lane == 0)
    {
#pragma unroll
        for (int k = 0; k != 3; ++k)
        {
            red[k][wave] = lo[k];
            red[3 + k][wave] = hi[k];
        }
        red[6][wave] = lam;
    }
    __syncthreads();
    if (t == 0)
    {
        BlockBound b;
#pragma unroll
        for (int k = 0; k != 3; ++k)
        {
            b.cmin[k] = fminf(fminf(red[k][0], red[k][1]), fminf(red[k][2], red[k][3]));
            b.cmax[k] = fmaxf(fmaxf(red[3 + k][0], red[3 + k][1]), fmaxf(red[3 + k][2], red[3 + k][3]));
        }
        b.lamMax = maxNan(maxNan(red[6][0], red[6][1]), maxNan(red[6][2], red[6][3]));
        b.pad = 0.0f;
        bounds[blockIdx.x] = b;
    }
}

// Per frame: one lane per (block, view) -> flags[view][block] = 1 unless the bound proves that no splat of the block reaches the band;
// lane 0 of every view leaves the view's terms for the band kernel's per-splat test.
__global__ __launch_bounds__(256) void blockBandTestKernel(int blocks, Screen scr, ClipCamera camArg, const FrameParams* __restrict__ params,
                                                            const BlockBound* __restrict__ bounds, uint8_t* __restrict__ flags,
                                                            BandViewTerms* __restrict__ viewTermsOut, unsigned long long* __restrict__ bandMask,
                                                            int slotsPerView)
{
    const int block = blockIdx.x * blockDim.x + threadIdx.x;
    if (block >= blocks) return;
    const int view = blockIdx.y;
    // the caller-ordered keep mask of this frame starts empty: the band kernel's kept splats OR their bits in (four words per 256 splats)
#pragma unroll
    for (int k = 0; k != kBandSlots; ++k)
        if (block * kBandSlots + k < slotsPerView) bandMask[(size_t) view * slotsPerView + block * kBandSlots + k] = 0ull;
    const ClipCamera cam = cam_or(params, view, camArg);
    const BandViewTerms vt = makeBandViewTerms(cam);
    if (block == 0) viewTermsOut[view] = vt;
    const BlockBound b = bounds[block];
    flags[(size_t) view * blocks + block] = bandBoundRejects(b.cmin, b.cmax, b.lamMax, vt, cam, scr) ? 0 : 1;
}

// What the compaction needs besides the caller-ordered keep masks the block-culled band kernel filled: per 64 splats the kept splats of
// the workgroup's earlier slots, per 256 their count (the tail of splatClipBandKernel).  One lane per block of 256 splats.
__global__ __launch_bounds__(256) void bandSlotCountsKernel(int count, const unsigned long long* __restrict__ bandMask,
                                                             int32_t* __restrict__ bandSlotPrefix, int32_t* __restrict__ bandBlockCounts)
{
    const int slotsPerView = (count + kWave - 1) / kWave, blocksPerView = (count + kBandBlock - 1) / kBandBlock;
    const int block = blockIdx.x * blockDim.x + threadIdx.x;
    if (block >= blocksPerView) return;
    const int view = blockIdx.y;
    int prefix = 0;
#pragma unroll
    for (int k = 0; k != kBandSlots; ++k)
    {
        const int slot = block * kBandSlots + k;
        if (slot < slotsPerView)
        {
            bandSlotPrefix[view * slotsPerView + slot] = prefix;
            prefix += __popcll(bandMask[view * slotsPerView + slot]);
        }
    }
    bandBlockCounts[view * blocksPerView + block] = prefix;
}

// Order-preserving compaction of the splats flagged by the band cull: depth key + (virtual) splat index of every kept splat at
// blockOffsets[block of 256 splats] + (kept splats of the block's earlier slots) + (kept lanes below it).  With Morton storage
// masks and clip depths are the caller-ordered ones (see the band kernel) and lane c compacts the splat stored in slot splatToStorage[c].
__global__ __launch_bounds__(256) void compactBandKernel(int count, const unsigned long long* __restrict__ bandMask,
                                                          const int32_t* __restrict__ bandSlotPrefix, const int32_t* __restrict__ blockOffsets,
                                                          const float* __restrict__ clipZ, const int32_t* __restrict__ splatToStorage,
                                                          uint32_t* __restrict__ depthKey, uint32_t* __restrict__ splatOrder,
                                                          const float4* __restrict__ positionCaller, ClipCamera camArg, const FrameParams* __restrict__ params)
{
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= count) return;
    const int view = blockIdx.y;
    const int slotsPerView = (count + kWave - 1) / kWave, blocksPerView = (count + kBandBlock - 1) / kBandBlock;
    const int slot = view * slotsPerView + (index >> 6);
    const int lane = index & (kWave - 1);
    const unsigned long long mask = bandMask[slot]; // wave-uniform: scalar loads
    if ((mask >> lane) & 1ull)
    {
        const int out = view * count + (splatToStorage != nullptr ? splatToStorage[index] : index); // virtual splat (see the clip kernel)
        const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (kWave - lane));
        const int pos = blockOffsets[view * blocksPerView + index / kBandBlock] + bandSlotPrefix[slot] + __popcll(mask & lower);
        // (block cull: the band kernel walked the splats in storage order and stored no clip depth at all -- the key is made here from the
        // caller-ordered position by the routine the clip kernel and the depth sort's key generation share: the same bits)
        depthKey[pos] = positionCaller != nullptr ? depthKeyOfSplat(positionCaller[index], cam_or(params, view, camArg))
                                                  : compactDepthKey(clipZ[view * count + index]);
        splatOrder[pos] = (uint32_t) out;
    }
}

// centre and conic of every splat from the packed raster records (for callers that ask for the renderer's buffers)
__global__ __launch_bounds__(256) void unpackRasterRecordKernel(int count, const float4* __restrict__ rasterRecord, float2* __restrict__ clipXY,
                                                                 float* __restrict__ clipZ, float4* __restrict__ conic)
{
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= count) return;
    const float4 ra = rasterRecord[2 * index], rb = rasterRecord[2 * index + 1];
    clipXY[index] = make_float2(ra.x, ra.y);
    clipZ[index] = rb.y;
    conic[index] = make_float4(ra.z, ra.w, rb.x, 0.0f);
}

// screenEllipse of every splat from the packed tile records (for callers that ask for the renderer's buffers)
__global__ __launch_bounds__(256) void unpackEllipseKernel(int count, const float4* __restrict__ tileRecord, float4* __restrict__ screenEllipse)
{
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= count) return;
    const float4 ra = tileRecord[2 * index], rb = tileRecord[2 * index + 1];
    screenEllipse[index] = make_float4(ra.z, ra.w, rb.x, rb.y);
}

} // namespace

void launchUnpackRasterRecord(hipStream_t stream, int count, const float4* rasterRecord, float2* clipXY, float* clipZ, float4* conic)
{
    if (count > 0)
        hipLaunchKernelGGL(unpackRasterRecordKernel, dim3((count + 255) / 256), dim3(256), 0, stream, count, rasterRecord, clipXY, clipZ, conic);
}

void launchUnpackEllipse(hipStream_t stream, int count, const float4* tileRecord, float4* screenEllipse)
{
    if (count > 0) hipLaunchKernelGGL(unpackEllipseKernel, dim3((count + 255) / 256), dim3(256), 0, stream, count, tileRecord, screenEllipse);
}

void launchBandCompaction(hipStream_t stream, int count, int views, const unsigned long long* bandMask, const int32_t* bandSlotPrefix,
                          const int32_t* bandBlockCounts, int32_t* bandBlockOffsets, int64_t* bandTotal, const float* clipZ,
                          uint32_t* depthKey, uint32_t* splatOrder, const int32_t* splatToStorage, const float4* positionCallerOrNull,
                          const ClipCamera* cameraOrNull, const FrameParams* paramsDeviceOrNull)
{
    if (count <= 0) return;
    const int blocks = (count + kBandBlock - 1) / kBandBlock;
    launchScanInt32(stream, blocks * views, bandBlockCounts, bandBlockOffsets, bandTotal);
    hipLaunchKernelGGL(compactBandKernel, dim3((count + 255) / 256, views), dim3(256), 0, stream, count, bandMask, bandSlotPrefix,
                       bandBlockOffsets, clipZ, splatToStorage, depthKey, splatOrder, positionCallerOrNull, cameraOrNull != nullptr ? *cameraOrNull : ClipCamera{},
                       paramsDeviceOrNull);
}

void launchBlockBounds(hipStream_t stream, int count, const float4* position, const float4* scaleAndRotation, BlockBound* bounds)
{
    if (count > 0)
        hipLaunchKernelGGL(blockBoundsKernel, dim3((count + kBandBlock - 1) / kBandBlock), dim3(256), 0, stream, count, position, scaleAndRotation, bounds);
}

void launchSplatClipData(hipStream_t stream, int count, const Screen& scr, const ClipCamera& cam, const float4* position,
                         const float4* scaleAndRotation, float2* clipXY, float* clipZ, float4* screenEllipse, float4* conic,
                         uint32_t* depthKey, uint32_t* splatOrder, unsigned long long* bandMask, int32_t* bandSlotPrefix,
                         int32_t* bandBlockCounts, const FrameParams* paramsDevice, int32_t* rangeTableToClear, int rangeEntries,
                         float4* tileRecord, float4* rasterRecord, int views, const int32_t* bandOutSlot, ushort4* tileBox,
                         const BandBlockCull* blockCull, const TileOrderArgs* tileOrder)
{
    if (count <= 0) return;
    // views > 1: cameras from paramsDevice[view], results under view * count + splat
    const ClipOutputs o{clipXY, clipZ, screenEllipse, conic, depthKey, splatOrder, tileRecord, rasterRecord, tileBox};
    if (bandMask != nullptr && blockCull != nullptr)
    {
        // block test (+ empty masks) -> projection of the kept blocks in storage order, kept splats OR their bits into the caller-ordered
        // masks -> slot prefixes and block counts for the compaction
        const int blocks = (count + kBandBlock - 1) / kBandBlock, slots = (count + kWave - 1) / kWave;
        hipLaunchKernelGGL(blockBandTestKernel, dim3((blocks + 255) / 256, views), dim3(256), 0, stream, blocks, scr, cam, paramsDevice, blockCull->bounds,
                           blockCull->flags, blockCull->viewTerms, bandMask, slots);
        if (views > 1 && paramsDevice != nullptr)
            hipLaunchKernelGGL(splatClipBandViewsKernel, dim3(blocks), dim3(256), 0, stream, count, views, scr, paramsDevice, position, scaleAndRotation, o,
                               bandMask, nullptr, nullptr, rangeTableToClear, rangeEntries, nullptr, blockCull->flags, blockCull->viewTerms, blockCull->storageToSplat);
        else
            hipLaunchKernelGGL(splatClipBandKernel, dim3(blocks, views), dim3(256), 0, stream, count, scr, cam, paramsDevice, position, scaleAndRotation, o,
                               bandMask, nullptr, nullptr, rangeTableToClear, rangeEntries, nullptr, blockCull->flags, blockCull->viewTerms, blockCull->storageToSplat);
        hipLaunchKernelGGL(bandSlotCountsKernel, dim3((blocks + 255) / 256, views), dim3(256), 0, stream, count, bandMask, bandSlotPrefix, bandBlockCounts);
        return;
    }
#ifndef GS_CLIP_ALL_VIEWS
#define GS_CLIP_ALL_VIEWS 1 // A/B builds: 0 = one grid row per view (round 3)
#endif
    if (bandMask != nullptr && views > 1 && paramsDevice != nullptr && GS_CLIP_ALL_VIEWS)
        hipLaunchKernelGGL(splatClipBandViewsKernel, dim3((count + kBandBlock - 1) / kBandBlock), dim3(256), 0, stream, count, views, scr, paramsDevice,
                           position, scaleAndRotation, o, bandMask, bandSlotPrefix, bandBlockCounts, rangeTableToClear, rangeEntries, bandOutSlot, nullptr, nullptr, nullptr);
    else if (bandMask != nullptr)
        hipLaunchKernelGGL(splatClipBandKernel, dim3((count + kBandBlock - 1) / kBandBlock, views), dim3(256), 0, stream, count, scr, cam,
                           paramsDevice, position, scaleAndRotation, o, bandMask, bandSlotPrefix, bandBlockCounts, rangeTableToClear,
                           rangeEntries, bandOutSlot, nullptr, nullptr, nullptr);
    else
    {
        const TileOrderArgs order = tileOrder != nullptr ? *tileOrder : TileOrderArgs{};
        hipLaunchKernelGGL(splatClipDataKernel, dim3((count + 255) / 256 + (order.slots > 0 ? kOrderWorkgroups : 0), views), dim3(256), 0, stream, count, scr, cam,
                           paramsDevice, position, scaleAndRotation, o, rangeTableToClear, rangeEntries, order);
    }
}

namespace
{
// slices [first, end) of the deferred clip kernel in one launch, behind `scan` (rows == 0: no scan)
void launchClipSlices(hipStream_t stream, ClipWithScan& c, const RadixScanArgs& scan, int first, int end)
{
    const ClipOutputs o{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, c.tileRecord, c.rasterRecord, c.tileBox};
    const int clipBlocks = (c.count + 255) / 256, parts = std::max(c.parts, 1);
    const int firstBlock = (int) ((int64_t) clipBlocks * first / parts), endBlock = (int) ((int64_t) clipBlocks * end / parts);
    TileOrderArgs order = c.tileOrder;
    if (first != 0) order.slots = 0; // (the builder rides with the first slice)
    const int orderBlocks = order.slots > 0 ? kOrderWorkgroups : 0;
    if (scan.rows + orderBlocks + (endBlock - firstBlock) > 0)
        hipLaunchKernelGGL(splatClipDataScanKernel, dim3(scan.rows + orderBlocks + (endBlock - firstBlock), c.views), dim3(256), 0, stream, scan, c.count, c.scr,
                           c.camera, c.params, c.position, c.scaleAndRotation, o, c.rangeTableToClear, c.rangeEntries, firstBlock, clipBlocks, order);
    c.done = end;
}
} // namespace

void launchSplatClipDataWithScan(hipStream_t stream, ClipWithScan& c, const RadixScanArgs& scan)
{
    // (a scan offered after the last slice has gone is launched alone: the grid is then the scan's rows)
    launchClipSlices(stream, c, scan, c.done, std::min(c.done + 1, std::max(c.parts, 1)));
}

void launchSplatClipDataRest(hipStream_t stream, ClipWithScan& c)
{
    if (c.done < std::max(c.parts, 1)) launchClipSlices(stream, c, RadixScanArgs{}, c.done, std::max(c.parts, 1));
}

} // namespace gs
