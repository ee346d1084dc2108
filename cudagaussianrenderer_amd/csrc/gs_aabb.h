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

} // namespace gs
