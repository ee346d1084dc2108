// gs_depth_key.h -- view-space centre, clip position and the depth-sort key of one splat, shared by the clip kernel (gs_clip.hip) and by
// the key-generating first pass of the depth sort (gs_radix.hip) so that both produce the same bits.  No fp contraction in any of
// these functions, whatever the including file's default (GaussianRender.cu:227-259,265,316-321,446-454).
#pragma once

#include "gs_internal.h"

namespace gs
{

// Quantised depth of getKey() (GaussianRender.cu:446-454; (uint32) of 2^32 saturated) re-coded on kDepthKeyBits = 27 bits.
// q = floor(n * 2^32) with n a float in [0, 1] has at most 24 significant bits: for q >= 2^23 the bits below (top bit - 23)
// are zero, so (class = top bit position - 22, the 23 bits after the top bit) is a lossless, order-preserving code; q < 2^23
// is its own code (class 0); the saturated 0xFFFFFFFF of n == 1 (the only value with 32 significant bits) gets class 10.
__device__ __forceinline__ uint32_t compactDepthKey(float clipDepth)
{
#pragma clang fp contract(off)
    const float n01 = fminf(fmaxf((clipDepth + 1.0f) * 0.5f, 0.0f), 1.0f);
    const float scaled = n01 * 4294967296.0f;
    if (scaled >= 4294967296.0f) return 10u << 23;
    const uint32_t q = (uint32_t) scaled;
    if (q < (1u << 23)) return q;
    const int top = 31 - __clz((int) q);
    return ((uint32_t) (top - 22) << 23) | ((q >> (top - 23)) & 0x7FFFFFu);
}

// The low word of getKey() itself (GaussianRender.cu:446-454): (uint32) of clamp((z + 1) / 2, 0, 1) * 2^32, the 2^32 of depth == far
// saturated to 0xFFFFFFFF explicitly, NaN -> 0 (glm::clamp lets a NaN through; the conversion of a NaN gives 0).  compactDepthKey above
// is a lossless, order-preserving re-coding of exactly this value.
__device__ __forceinline__ uint32_t quantizeDepth(float clipDepth)
{
#pragma clang fp contract(off)
    const float n = (clipDepth + 1.0f) * 0.5f;
    const float lo = (n < 0.0f) ? 0.0f : n; // glm::max(n, 0): (x < y) ? y : x
    const float c = (1.0f < lo) ? 1.0f : lo; // glm::min(lo, 1): (y < x) ? y : x
    const float scaled = c * 4294967296.0f;
    if (scaled != scaled) return 0u;
    if (scaled >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t) scaled;
}

// View-space centre and the para-perspective terms every later step needs (GaussianRender.cu:227-259,265).  ONE routine for the
// band cull's early test, the projection proper and the depth sort's key generation, so that all see the same clip position bit for bit.
struct ViewTerms
{
    float vx, vy, vz, zRcp, zRcpSqr, scaleX, scaleY, cotTrsX, cotTrsY, shearX, shearY;
    float clipX, clipY, clipDepth;
    bool inside; // centre-in-frustum test of GaussianRender.cu:316-321 (without its lambda1 term, which never fails: lambda1 >= 0)
};
__device__ __forceinline__ ViewTerms viewTerms(const float4 pd, const ClipCamera& cam)
{
#pragma clang fp contract(off)
    ViewTerms t;
    // View-space centre (GaussianRender.cu:227); v[col*4+row].
    const float* v = cam.view;
    t.vx = (v[0] * pd.x + v[4] * pd.y) + (v[8] * pd.z + v[12]);
    t.vy = (v[1] * pd.x + v[5] * pd.y) + (v[9] * pd.z + v[13]);
    t.vz = (v[2] * pd.x + v[6] * pd.y) + (v[10] * pd.z + v[14]);
    // Para-perspective Jacobian (GaussianRender.cu:236-259).
    t.zRcp = 1.0f / t.vz;
    t.zRcpSqr = t.zRcp * t.zRcp;
    t.scaleX = -cam.fovCotX * t.zRcp;
    t.scaleY = -cam.fovCotY * t.zRcp;
    t.cotTrsX = cam.fovCotX * t.vx;
    t.cotTrsY = cam.fovCotY * t.vy;
    t.shearX = t.cotTrsX * t.zRcpSqr;
    t.shearY = t.cotTrsY * t.zRcpSqr;
    // Clip position = J * (view pos, 1) (GaussianRender.cu:265): x = scaleX*vx + shearX*vz + translationX.
    t.clipX = (t.scaleX * t.vx + 0.0f * t.vy) + (t.shearX * t.vz + (-t.cotTrsX * t.zRcp));
    t.clipY = (0.0f * t.vx + t.scaleY * t.vy) + (t.shearY * t.vz + (-t.cotTrsY * t.zRcp));
    t.clipDepth = (0.0f * t.vx + 0.0f * t.vy) + (cam.depthScale * t.vz + cam.depthBias);
    // Centre-in-frustum cull (GaussianRender.cu:316-321). step(edge,x) = x < edge ? 0 : 1, so NaN passes, as in glm.
    t.inside = !(t.clipX < -1.0f) && !(1.0f < t.clipX) && !(t.clipY < -1.0f) && !(1.0f < t.clipY) && !(t.clipDepth < -1.0f) &&
               !(1.0f < t.clipDepth);
    return t;
}

// mix(-128, clip, v) = -128*(1-v) + clip*v (GaussianRender.cu:320), kept as arithmetic so that inf / NaN propagate as in the reference.
__device__ __forceinline__ float mixInvisible(float clipValue, float isVisible)
{
#pragma clang fp contract(off)
    return -128.0f * (1.0f - isVisible) + clipValue * isVisible;
}

// The depth-sort key the clip kernel stores for a splat, from its position alone: the visibility that decides between the clip
// depth and -128 is the centre-in-frustum test (the eigenvalue term of GaussianRender.cu:318 never fails).
__device__ __forceinline__ uint32_t depthKeyOfSplat(const float4 pd, const ClipCamera& cam)
{
    const ViewTerms t = viewTerms(pd, cam);
    return compactDepthKey(mixInvisible(t.clipDepth, t.inside ? 1.0f : 0.0f));
}

} // namespace gs
