// gs_overlap_exact.h -- the reference's ellipse / tile-rectangle overlap predicate (GaussianRender.cu:351-405), operation for
// operation: every product, sum and IEEE division rounds as the reference's source (and the CPU oracle) writes it.  Shared by
// the staged builder (gs_tiles.hip: every tile) and by the renderer's builder (gs_tiles_fast.hip: the tiles whose cheap
// decision value lies inside its guard band), so that both produce the same pair list bit for bit.
// No fp contraction in any of these functions, whatever the including file's default.
#pragma once

#include "gs_internal.h"

namespace gs
{

struct EllipseD
{
    float cx, cy, ex, ey, cosA, sinA;
};

// CUDA __saturatef: clamp to [0,1], NaN -> 0.
__device__ __forceinline__ float saturateExact(float x) { return (x != x) ? 0.0f : fminf(fmaxf(x, 0.0f), 1.0f); }

// Clip-space minimum of tile g along one axis (GaussianRender.cu:656-664): product and difference rounded separately.
__device__ __forceinline__ float tileClipMin(int g, float tileClipSize)
{
#pragma clang fp contract(off)
    return (float) g * tileClipSize - 1.0f;
}

// GaussianRender.cu:351-358
__device__ __forceinline__ float2 toEllipseExact(const EllipseD& e, float px, float py)
{
#pragma clang fp contract(off)
    const float dx = px - e.cx, dy = py - e.cy;
    return make_float2((dx * e.cosA + dy * e.sinA) / e.ex, (dy * e.cosA - dx * e.sinA) / e.ey);
}

// GaussianRender.cu:361-372
__device__ __forceinline__ bool intersectsUnitCircleExact(float2 v0, float2 v1)
{
#pragma clang fp contract(off)
    const float dx = v1.x - v0.x, dy = v1.y - v0.y;
    const float lengthSqr = dx * dx + dy * dy;
    const float t = saturateExact((-v0.x * dx + -v0.y * dy) / lengthSqr);
    const float px = v0.x + t * (v1.x - v0.x), py = v0.y + t * (v1.y - v0.y);
    return px * px + py * py < 1.0f;
}

// GaussianRender.cu:375-405
__device__ __forceinline__ bool ellipseRectOverlapExact(const EllipseD& e, float minX, float minY, float maxX, float maxY)
{
#pragma clang fp contract(off)
    bool overlaps = e.cx > minX && e.cx < maxX && e.cy > minY && e.cy < maxY;
    const float2 c = toEllipseExact(e, minX + (maxX - minX) * 0.5f, minY + (maxY - minY) * 0.5f);
    overlaps |= c.x * c.x + c.y * c.y < 1.0f;
    const float2 p0 = toEllipseExact(e, minX, minY);
    const float2 p1 = toEllipseExact(e, maxX, minY);
    const float2 p2 = toEllipseExact(e, maxX, maxY);
    const float2 p3 = toEllipseExact(e, minX, maxY);
    overlaps |= intersectsUnitCircleExact(p0, p1);
    overlaps |= intersectsUnitCircleExact(p1, p2);
    overlaps |= intersectsUnitCircleExact(p2, p3);
    overlaps |= intersectsUnitCircleExact(p3, p0);
    return overlaps;
}

// The same predicate, edge by edge in a rolled loop (the same operations on the same operands give the same bits; the four
// comparisons are OR-ed, so stopping at the first true one changes nothing).  For callers that evaluate it rarely and must not pay
// for it in registers: a handful of live values instead of the straight-line version's four mapped corners and ten divisions in flight.
__device__ __forceinline__ bool ellipseRectOverlapExactRolled(const EllipseD& e, float minX, float minY, float maxX, float maxY)
{
#pragma clang fp contract(off)
    if (e.cx > minX && e.cx < maxX && e.cy > minY && e.cy < maxY) return true;
    const float2 c = toEllipseExact(e, minX + (maxX - minX) * 0.5f, minY + (maxY - minY) * 0.5f);
    if (c.x * c.x + c.y * c.y < 1.0f) return true;
    // corners in the reference's order: (min,min) (max,min) (max,max) (min,max); edge i joins corner i and corner i+1
    float2 pa = toEllipseExact(e, minX, minY);
    const float2 first = pa;
    bool overlaps = false;
#pragma unroll 1
    for (int edge = 0; edge != 4 && !overlaps; ++edge)
    {
        const int b = edge + 1;
        __builtin_amdgcn_sched_barrier(0); // one division sequence after the other: this path is rare and must stay small in registers
        const float2 pb = b == 4 ? first : toEllipseExact(e, (b == 1 || b == 2) ? maxX : minX, b >= 2 ? maxY : minY);
        __builtin_amdgcn_sched_barrier(0);
        overlaps = intersectsUnitCircleExact(pa, pb);
        pa = pb;
    }
    return overlaps;
}

} // namespace gs
