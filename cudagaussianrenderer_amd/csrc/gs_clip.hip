// gs_clip.hip -- per-splat world -> clip data: centre, depth, 3-sigma screen ellipse, conic.
//
// Replaces evaluateSplatClipDataKernel (GaussianRender.cu:204-329).  Pure streaming kernel:
// 32 B read + 44 B written per splat, one splat per lane with 16-byte coalesced loads/stores.
// The arithmetic follows the reference's order (covariance = (R S)(R S)^T, para-perspective
// Jacobian, J W Sigma^T (J W)^T, trace bump, eigen decomposition, conic = Sigma^-1) on plain floats;
// glm's 3x3 products are written out so the symmetric halves are not recomputed.
//
// One deliberate simplification: the reference stores (cos(atan2(e.y,e.x)), sin(atan2(e.y,e.x)))
// of the NORMALISED principal eigenvector e; that is e itself, which is what is stored here.
#include "gs_aabb.h"
#include "gs_internal.h"

namespace gs
{
namespace
{

struct Sym3
{
    float xx, xy, xz, yy, yz, zz;
};

// Quantised depth of getKey() (GaussianRender.cu:446-454; (uint32) of 2^32 saturated) re-coded on kDepthKeyBits = 27 bits.
// q = floor(n * 2^32) with n a float in [0, 1] has at most 24 significant bits: for q >= 2^23 the bits below (top bit - 23)
// are zero, so (class = top bit position - 22, the 23 bits after the top bit) is a lossless, order-preserving code; q < 2^23
// is its own code (class 0); the saturated 0xFFFFFFFF of n == 1 (the only value with 32 significant bits) gets class 10.
__device__ __forceinline__ uint32_t compactDepthKey(float clipDepth)
{
    const float n01 = fminf(fmaxf((clipDepth + 1.0f) * 0.5f, 0.0f), 1.0f);
    const float scaled = n01 * 4294967296.0f;
    if (scaled >= 4294967296.0f) return 10u << 23;
    const uint32_t q = (uint32_t) scaled;
    if (q < (1u << 23)) return q;
    const int top = 31 - __clz((int) q);
    return ((uint32_t) (top - 22) << 23) | ((q >> (top - 23)) & 0x7FFFFFu);
}

__global__ __launch_bounds__(256) void splatClipDataKernel(int count, Screen scr, ClipCamera camArg, const FrameParams* __restrict__ params,
                                                            const float4* __restrict__ position,
                                                            const float4* __restrict__ scaleAndRotation,
                                                            float2* __restrict__ clipXY, float* __restrict__ clipZ,
                                                            float4* __restrict__ screenEllipse, float4* __restrict__ conicOut,
                                                            uint32_t* __restrict__ depthKey, uint32_t* __restrict__ splatOrder,
                                                            unsigned long long* __restrict__ bandMask,
                                                            int32_t* __restrict__ bandWaveCounts, int32_t* __restrict__ rangeTable,
                                                            int rangeEntries, float4* __restrict__ tileRecord,
                                                            float4* __restrict__ rasterRecord)
{
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    // The renderer has this kernel reset the tile range table to -1 (Demo.cpp:403) on the side: it is the first kernel
    // of the frame and nothing reads the table before the range search, so the frame needs no separate clear.
    if (rangeTable != nullptr)
        for (int i = index; i < rangeEntries; i += (int) (gridDim.x * blockDim.x)) rangeTable[i] = -1;
    if (index >= count) return;
    const ClipCamera cam = params ? params->camera : camArg; // device-resident when the frame is a captured graph

    const float4 pd = position[index];
    const float4 sr = scaleAndRotation[index];

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

    // View-space centre (GaussianRender.cu:227); v[col*4+row].
    const float* v = cam.view;
    const float vx = (v[0] * pd.x + v[4] * pd.y) + (v[8] * pd.z + v[12]);
    const float vy = (v[1] * pd.x + v[5] * pd.y) + (v[9] * pd.z + v[13]);
    const float vz = (v[2] * pd.x + v[6] * pd.y) + (v[10] * pd.z + v[14]);

    // Para-perspective Jacobian (GaussianRender.cu:236-259).
    const float zRcp = 1.0f / vz;
    const float zRcpSqr = zRcp * zRcp;
    const float scaleX = -cam.fovCotX * zRcp;
    const float scaleY = -cam.fovCotY * zRcp;
    const float cotTrsX = cam.fovCotX * vx;
    const float cotTrsY = cam.fovCotY * vy;
    const float shearX = cotTrsX * zRcpSqr;
    const float shearY = cotTrsY * zRcpSqr;

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

    // Clip position = J * (view pos, 1) (GaussianRender.cu:265): x = scaleX*vx + shearX*vz + translationX.
    float clipX = (scaleX * vx + 0.0f * vy) + (shearX * vz + (-cotTrsX * zRcp));
    float clipY = (0.0f * vx + scaleY * vy) + (shearY * vz + (-cotTrsY * zRcp));
    float clipDepth = (0.0f * vx + 0.0f * vy) + (cam.depthScale * vz + cam.depthBias);

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

    // Centre-in-frustum cull (GaussianRender.cu:316-321). step(edge,x) = x < edge ? 0 : 1, so NaN passes, as in glm.
    const bool inside = !(clipX < -1.0f) && !(1.0f < clipX) && !(clipY < -1.0f) && !(1.0f < clipY) && !(clipDepth < -1.0f) &&
                        !(1.0f < clipDepth) && !(lambda1 < 0.0f);
    const float isVisible = inside ? 1.0f : 0.0f;
    // mix(-128, clip, v) = -128*(1-v) + clip*v ; extent *= v  (kept as arithmetic so inf/NaN propagate as in the reference)
    clipX = -128.0f * (1.0f - isVisible) + clipX * isVisible;
    clipY = -128.0f * (1.0f - isVisible) + clipY * isVisible;
    clipDepth = -128.0f * (1.0f - isVisible) + clipDepth * isVisible;
    extentX *= isVisible;
    extentY *= isVisible;

    if (bandMask != nullptr)
    {
        // Band cull for the multi-GPU split: does this splat's tile AABB reach the tile rows this GPU owns?  Same AABB
        // routine as the tile-list builder, so nothing that would emit a pair is dropped. One 64-bit mask per wave.
        const TileAabb box = splatTileAabb(clipX, clipY, cosA, sinA, extentX, extentY, scr);
        const bool kept = box.x1 > box.x0 && box.y1 > box.y0;
        const unsigned long long mask = __ballot(kept);
        if ((threadIdx.x & (kWave - 1)) == 0)
        {
            bandMask[index >> 6] = mask;
            bandWaveCounts[index >> 6] = __popcll(mask);
        }
        // Nothing downstream ever looks at a splat that does not reach the band (the compacted list holds the kept ones only):
        // its 60 bytes of per-splat outputs are not written.  With 8 bands ~80 % of the splats stop here, which takes this
        // replicated, HBM-bound pass from ~100 to ~45 bytes per splat.
        if (!kept) return;
    }
    if (depthKey != nullptr)
    {
        // Quantised depth of getKey() (GaussianRender.cu:446-454) as a sort key of its own (27-bit code, see above), plus
        // the identity permutation: the one-call renderer sorts SPLATS by depth before expanding them to tiles.
        depthKey[index] = compactDepthKey(clipDepth);
        splatOrder[index] = (uint32_t) index;
    }
    if (clipXY != nullptr) clipXY[index] = make_float2(clipX, clipY);
    if (clipZ != nullptr) clipZ[index] = clipDepth;
    if (screenEllipse != nullptr) screenEllipse[index] = make_float4(cosA, sinA, extentX, extentY);
    if (tileRecord != nullptr)
    {
        // the same six numbers once more as ONE 32-byte record: the tile-list passes fetch them through the depth order
        // (random access), where two separate arrays cost two memory sectors per splat
        tileRecord[2 * index] = make_float4(clipX, clipY, cosA, sinA);
        tileRecord[2 * index + 1] = make_float4(extentX, extentY, 0.0f, 0.0f);
    }
    if (conicOut != nullptr) conicOut[index] = make_float4(covC * invDet, -covB * invDet, covA * invDet, 0.0f);
    if (rasterRecord != nullptr)
    {
        // what the raster kernel gathers per list entry besides the colour, as ONE 32-byte record (one memory sector instead of one
        // for the centre and one for the conic)
        rasterRecord[2 * index] = make_float4(clipX, clipY, covC * invDet, -covB * invDet);
        rasterRecord[2 * index + 1] = make_float4(covA * invDet, clipDepth, 0.0f, 0.0f); // (the depth rides along for gs_renderer_buffers)
    }
}

// Order-preserving compaction of the splats flagged by the band cull: depth key + splat index of every kept splat at
// waveOffsets[wave] + (kept lanes below it).
__global__ __launch_bounds__(256) void compactBandKernel(int count, const unsigned long long* __restrict__ bandMask,
                                                          const int32_t* __restrict__ waveOffsets, const float* __restrict__ clipZ,
                                                          uint32_t* __restrict__ depthKey, uint32_t* __restrict__ splatOrder)
{
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= count) return;
    const int lane = index & (kWave - 1);
    const unsigned long long mask = bandMask[index >> 6];
    if ((mask >> lane) & 1ull)
    {
        const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (kWave - lane));
        const int pos = waveOffsets[index >> 6] + __popcll(mask & lower);
        depthKey[pos] = compactDepthKey(clipZ[index]);
        splatOrder[pos] = (uint32_t) index;
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

void launchCompactBand(hipStream_t stream, int count, const unsigned long long* bandMask, const int32_t* waveOffsets, const float* clipZ,
                       uint32_t* depthKey, uint32_t* splatOrder)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(compactBandKernel, dim3((count + 255) / 256), dim3(256), 0, stream, count, bandMask, waveOffsets, clipZ, depthKey,
                       splatOrder);
}

void launchSplatClipData(hipStream_t stream, int count, const Screen& scr, const ClipCamera& cam, const float4* position,
                         const float4* scaleAndRotation, float2* clipXY, float* clipZ, float4* screenEllipse, float4* conic,
                         uint32_t* depthKey, uint32_t* splatOrder, unsigned long long* bandMask, int32_t* bandWaveCounts,
                         const FrameParams* paramsDevice, int32_t* rangeTableToClear, int rangeEntries, float4* tileRecord,
                         float4* rasterRecord)
{
    if (count <= 0) return;
    const dim3 grid((count + 255) / 256), block(256);
    hipLaunchKernelGGL(splatClipDataKernel, grid, block, 0, stream, count, scr, cam, paramsDevice, position, scaleAndRotation, clipXY, clipZ,
                       screenEllipse, conic, depthKey, splatOrder, bandMask, bandWaveCounts, rangeTableToClear, rangeEntries, tileRecord,
                       rasterRecord);
}

} // namespace gs
