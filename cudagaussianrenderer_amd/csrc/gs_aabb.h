// gs_aabb.h -- tile-space bounding box of a splat's 3-sigma ellipse (GaussianRender.cu:408-436,537-551),
// shared by the renderer's tile-list builder and by the band cull in the clip kernel so that both agree bit for
// bit on which tile rows a splat can touch.  No fp contraction: every operation rounds as written.
#pragma once

#include "gs_internal.h"

namespace gs
{

struct TileAabb
{
    int x0, y0, x1, y1; // [x0,x1) x [y0,y1) in tiles, clamped to the screen and to the tile-row band
};

__device__ __forceinline__ TileAabb splatTileAabb(float cx, float cy, float cosA, float sinA, float extentX, float extentY,
                                                  const Screen& scr)
{
#pragma clang fp contract(off)
    auto gmin = [](float x, float y) { return (y < x) ? y : x; }; // glm::min / glm::max semantics
    auto gmax = [](float x, float y) { return (x < y) ? y : x; };
    auto f2iSat = [](float v) -> int {                            // v_cvt_i32_f32: truncate, saturate, NaN -> 0
        if (v != v) return 0;
        if (v >= 2147483648.0f) return 2147483647;
        if (v <= -2147483648.0f) return -2147483647 - 1;
        return (int) v;
    };
    const float rightX = cosA * extentX, rightY = sinA * extentX;
    const float upX = sinA * extentY, upY = -cosA * extentY;
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
    minX += cx;
    minY += cy;
    maxX += cx;
    maxY += cy;
    TileAabb b;
    b.x0 = min(max(f2iSat(floorf((minX + 1.0f) * 0.5f * scr.tilesPerScreenX)), 0), scr.tilesX);
    b.y0 = min(max(f2iSat(floorf((minY + 1.0f) * 0.5f * scr.tilesPerScreenY)), 0), scr.tilesY);
    b.x1 = min(max(f2iSat(ceilf((maxX + 1.0f) * 0.5f * scr.tilesPerScreenX)), 0), scr.tilesX);
    b.y1 = min(max(f2iSat(ceilf((maxY + 1.0f) * 0.5f * scr.tilesPerScreenY)), 0), scr.tilesY);
    // multi-GPU: only the tile rows of this band (tiles are independent, so this only removes pairs)
    b.y0 = max(b.y0, scr.rowBegin);
    b.y1 = min(b.y1, scr.rowEnd);
    return b;
}

// The reference's box (GaussianRender.cu:408-436) bounds the ellipse's oriented bounding RECTANGLE: half width |cos| ex + |sin| ey.  The
// ellipse itself reaches sqrt((cos ex)^2 + (sin ey)^2) -- for a round splat at 45 degrees 1/sqrt(2) of that -- and a tile outside the
// ellipse's own bounding box cannot pass the overlap predicate.  Testing only the tiles of the intersection changes nothing but the
// number of tests (9.0 M -> 7.x M per frame at C3) as long as the tighter box errs on the large side: the half extents are inflated
// by 1e-4 relative + 1e-6 clip units, forty times what the predicate's own rounding can move a decision (see the guard band), and
// floor / ceil keep every tile the inflated box touches.  A NaN ellipse keeps the reference's box (it emits no pairs anyway).
#ifndef GS_TIGHT_BOX
#define GS_TIGHT_BOX 1
#endif
__device__ __forceinline__ void tightenToEllipse(TileAabb& box, float cx, float cy, float4 se, const Screen& scr)
{
    const float rx = se.x * se.z, ux = se.y * se.w; // x components of the half axes (cos ex, sin ey)
    const float ry = se.y * se.z, uy = se.x * se.w; // y components (sin ex, -cos ey)
    const float hx = sqrtf(fmaf(rx, rx, ux * ux)), hy = sqrtf(fmaf(ry, ry, uy * uy));
    if (!(hx < 1.0e30f && hy < 1.0e30f)) return; // NaN or absurd: the reference's box stands
    const float mx = fmaf(hx, 1.0e-4f, 1.0e-6f) + hx, my = fmaf(hy, 1.0e-4f, 1.0e-6f) + hy;
    const float fx0 = floorf(((cx - mx) + 1.0f) * 0.5f * scr.tilesPerScreenX), fx1 = ceilf(((cx + mx) + 1.0f) * 0.5f * scr.tilesPerScreenX);
    const float fy0 = floorf(((cy - my) + 1.0f) * 0.5f * scr.tilesPerScreenY), fy1 = ceilf(((cy + my) + 1.0f) * 0.5f * scr.tilesPerScreenY);
    // (compared as floats: a bound far off screen stays far off screen; the reference's box is already clamped to the screen)
    if (fx0 > (float) box.x0) box.x0 = (int) fminf(fx0, (float) box.x1);
    if (fx1 < (float) box.x1) box.x1 = (int) fmaxf(fx1, (float) box.x0);
    if (fy0 > (float) box.y0) box.y0 = (int) fminf(fy0, (float) box.y1);
    if (fy1 < (float) box.y1) box.y1 = (int) fmaxf(fy1, (float) box.y0);
}

// The box the renderer's list builder walks for a splat (the reference's box cut down to the ellipse's own bounding box): one routine
// for the builder's prologue and for whoever precomputes it (the clip kernel, for the depth layers' live test).
__device__ __forceinline__ TileAabb splatTileBox(float cx, float cy, float cosA, float sinA, float extentX, float extentY, const Screen& scr)
{
    TileAabb box = splatTileAabb(cx, cy, cosA, sinA, extentX, extentY, scr);
#if GS_TIGHT_BOX
    tightenToEllipse(box, cx, cy, make_float4(cosA, sinA, extentX, extentY), scr);
#endif
    return box;
}

} // namespace gs
