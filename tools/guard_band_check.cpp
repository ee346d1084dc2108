// guard_band_check.cpp -- CPU experiment behind the guard band of the renderer's tile-list builder (gs_tiles_fast.hip).
//
// The builder decides ellipse/tile overlap with a cheap formulation of GaussianRender.cu:375-405 (unit-circle space, hoisted
// affine map, reciprocals, fma) and re-evaluates a tile with the reference's own arithmetic (gs_overlap_exact.h) whenever the
// cheap decision value lies within a guard band around 1.  This program runs both formulations in IEEE float32 on the host
// (fmaf = v_fma_f32, 1.0f / x = the correctly rounded device division, the kernel's v_rcp_f32 modelled as that moved by one ulp) over random ellipses and all the tiles of their
// bounding boxes and reports how far the two decision values are apart, relative to the guard the kernel uses -- the
// evidence for the guard's constant.  Build: g++ -O2 -std=c++17 -ffp-contract=off tools/guard_band_check.cpp -o /tmp/guard
//
// Test infrastructure only; nothing here is linked into the product.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>

namespace
{
struct Ell
{
    float cx, cy, ex, ey, c, s;
};

// ---- the reference's arithmetic (same text as the oracle's / gs_tiles.hip's predicate) -------------------------------------
inline float saturatef(float x) { return (x != x) ? 0.0f : std::fmin(std::fmax(x, 0.0f), 1.0f); }
inline void toE(const Ell& e, float px, float py, float& X, float& Y)
{
    const float dx = px - e.cx, dy = py - e.cy;
    X = (dx * e.c + dy * e.s) / e.ex;
    Y = (dy * e.c - dx * e.s) / e.ey;
}
inline float edgeExact(float ax, float ay, float bx, float by)
{
    const float dx = bx - ax, dy = by - ay;
    const float l = dx * dx + dy * dy;
    const float t = saturatef((-ax * dx + -ay * dy) / l);
    const float px = ax + t * (bx - ax), py = ay + t * (by - ay);
    return px * px + py * py;
}
// smallest of the five squared distances the reference compares with 1 (NaN never wins), and the centre-in-rectangle bit
inline float exactMin(const Ell& e, float minX, float minY, float maxX, float maxY, bool& inside)
{
    inside = e.cx > minX && e.cx < maxX && e.cy > minY && e.cy < maxY;
    float X, Y;
    toE(e, minX + (maxX - minX) * 0.5f, minY + (maxY - minY) * 0.5f, X, Y);
    float m = X * X + Y * Y;
    float p0x, p0y, p1x, p1y, p2x, p2y, p3x, p3y;
    toE(e, minX, minY, p0x, p0y);
    toE(e, maxX, minY, p1x, p1y);
    toE(e, maxX, maxY, p2x, p2y);
    toE(e, minX, maxY, p3x, p3y);
    const float d[4] = {edgeExact(p0x, p0y, p1x, p1y), edgeExact(p1x, p1y, p2x, p2y), edgeExact(p2x, p2y, p3x, p3y),
                        edgeExact(p3x, p3y, p0x, p0y)};
    for (float v : d)
        if (!(m <= v)) m = (v == v) ? v : m; // min that ignores NaN
    return m;
}

// ---- the kernel's cheap formulation (gs_tiles_fast.hip) -------------------------------------------------------------------
inline float med3(float a, float lo, float hi) { return std::fmin(std::fmax(a, lo), hi); }
inline float edgeFast(float qx, float qy, float ex, float ey, float qe, float invEE)
{
    const float t = med3(-qe * invEE, 0.0f, 1.0f);
    const float rx = fmaf(t, ex, qx), ry = fmaf(t, ey, qy);
    return fmaf(rx, rx, ry * ry);
}
struct Hoisted
{
    float m00, m01, m10, m11, invUU, invVV;
};
// v_rcp_f32 is accurate to 1 ulp, not correctly rounded: the host model moves every reciprocal one ulp up or down
inline float rcpApprox(float x, int way) { return std::nextafter(1.0f / x, way > 0 ? INFINITY : -INFINITY); }
inline Hoisted hoist(const Ell& e, float sx, float sy, int way)
{
    Hoisted h;
    const float invEx = rcpApprox(e.ex, way & 1 ? 1 : -1), invEy = rcpApprox(e.ey, way & 2 ? 1 : -1);
    h.m00 = e.c * invEx;
    h.m01 = e.s * invEx;
    h.m10 = -e.s * invEy;
    h.m11 = e.c * invEy;
    const float ux = h.m00 * sx, uy = h.m10 * sx, vx = h.m01 * sy, vy = h.m11 * sy;
    h.invUU = rcpApprox(fmaf(ux, ux, uy * uy), way & 4 ? 1 : -1);
    h.invVV = rcpApprox(fmaf(vx, vx, vy * vy), way & 8 ? 1 : -1);
    return h;
}
inline float fastMin(const Ell& e, const Hoisted& h, float minX, float minY, float sx, float sy, float& scale, float& cond)
{
    const float ux = h.m00 * sx, uy = h.m10 * sx, vx = h.m01 * sy, vy = h.m11 * sy;
    const float uv = fmaf(ux, vx, uy * vy);
    const float dx = minX - e.cx, dy = minY - e.cy;
    const float q0x = fmaf(h.m00, dx, h.m01 * dy), q0y = fmaf(h.m10, dx, h.m11 * dy);
    const float qcx = fmaf(0.5f, ux + vx, q0x), qcy = fmaf(0.5f, uy + vy, q0y);
    const float centre2 = fmaf(qcx, qcx, qcy * qcy);
    const float q0u = fmaf(q0x, ux, q0y * uy), q0v = fmaf(q0x, vx, q0y * vy);
    const float bottom2 = edgeFast(q0x, q0y, ux, uy, q0u, h.invUU);
    const float right2 = edgeFast(q0x + ux, q0y + uy, vx, vy, q0v + uv, h.invVV);
    const float top2 = edgeFast(q0x + vx, q0y + vy, ux, uy, q0u + uv, h.invUU);
    const float left2 = edgeFast(q0x, q0y, vx, vy, q0v, h.invVV);
    scale = std::fabs(q0x) + std::fabs(q0y) + std::fabs(ux) + std::fabs(uy) + std::fabs(vx) + std::fabs(vy);
    // the magnitude the corner's rounding errors really scale with: the terms of the two dot products before they cancel
    cond = (std::fabs(h.m00 * dx) + std::fabs(h.m01 * dy)) + (std::fabs(h.m10 * dx) + std::fabs(h.m11 * dy)) + std::fabs(ux) + std::fabs(uy) + std::fabs(vx) + std::fabs(vy);
    return std::fmin(std::fmin(std::fmin(bottom2, right2), std::fmin(top2, left2)), centre2);
}
// ---- round 4, second formulation: only the two edges NEAREST to the ellipse's centre ---------------------------------------
// The smallest value of the quadratic form over a rectangle that does not hold the centre lies on an edge the centre can see (the
// form grows along every ray from the centre): the left edge when the centre is left of the tile, the right one when it is right of
// it -- likewise bottom / top -- and the rectangle's own centre is never closer than the point where the segment towards it enters
// the rectangle.  For a centre inside the tile's column (row) neither vertical (horizontal) edge is visible and either may be
// taken -- no point of the rectangle is closer than the visible edge's closest one; taking the NEARER one (one compare against the
// tile's midline, which outside the column IS the visible edge) also covers a centre within rounding of a tile boundary: the tile
// it grazes gets a distance of ~0 whichever side of the boundary the arithmetic puts it on.  Two edge distances instead of four plus
// the centre term; the values that are computed are the SAME bits as fastMin's (fma(1, u, q) == q + u).
inline float fastMinVisible(const Ell& e, const Hoisted& h, float minX, float minY, float sx, float sy)
{
    const float ux = h.m00 * sx, uy = h.m10 * sx, vx = h.m01 * sy, vy = h.m11 * sy;
    const float uv = fmaf(ux, vx, uy * vy);
    const float dx = minX - e.cx, dy = minY - e.cy;
    const float q0x = fmaf(h.m00, dx, h.m01 * dy), q0y = fmaf(h.m10, dx, h.m11 * dy);
    const float q0u = fmaf(q0x, ux, q0y * uy), q0v = fmaf(q0x, vx, q0y * vy);
    const float fx = dx < -0.5f * sx ? 1.0f : 0.0f, fy = dy < -0.5f * sy ? 1.0f : 0.0f; // the centre is right of / above the tile's midline
    const float xEdge = edgeFast(fmaf(fx, ux, q0x), fmaf(fx, uy, q0y), vx, vy, fmaf(fx, uv, q0v), h.invVV); // left or right
    const float yEdge = edgeFast(fmaf(fy, vx, q0x), fmaf(fy, vy, q0y), ux, uy, fmaf(fy, uv, q0u), h.invUU); // bottom or top
    return std::fmin(xEdge, yEdge);
}
} // namespace

int main(int argc, char** argv)
{
    const long long ellipses = argc > 1 ? atoll(argv[1]) : 2000000;
    const int widths[] = {1920, 1024, 333, 3840, 256};
    const int heights[] = {1080, 1024, 177, 2160, 256};
    std::mt19937_64 rng(1234);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    // worst |fast - exact| decision value, relative to candidate scales, over tests whose decision value is near 1
    double worstAbs = 0, worstRelS = 0, worstRelS2 = 0;
    long long tests = 0, straddle = 0, near = 0;
    // histogram of |fast - 1| / (1 + S^2) over the straddling cases (decision differs): what the guard must cover
    double worstStraddleRelS = 0, worstStraddleRelS2 = 0;
    long long guardHits[4] = {0, 0, 0, 0}; // tests inside the guard for k = 2e-6, 4e-6, 1e-5 (x (1+S)^2) and 1e-3 flat
    // how the disagreement of the two decision values scales with S, for values within 5 % of the decision: worst |fast - exact| per bin of S
    // the kernel's guard: kGuard (1 + A), A = the cancellation-aware magnitude below; first filter per splat with A* (the corner of a tile
    // whose decision value is near 1 lies within 1.05 x the larger extent + one tile of the centre on either axis), then per tile
    const double kGuard = 1.0e-5;
    long long insideSplatGuard = 0, insideBothGuards = 0, straddleOutside = 0;
    // the visible-edges formulation under the same guard
    long long visStraddleOutside = 0, visDiffersFromFast = 0, visInsideBoth = 0;
    double visWorstVsFastRel = 0, visWorstVsExactRelA = 0;
    constexpr int kBins = 9;
    double binWorst[kBins] = {}, binWorstA[kBins] = {};
    long long binCount[kBins] = {}, binCountA[kBins] = {};
    for (long long i = 0; i != ellipses; ++i)
    {
        const int r = (int) (i % 5);
        const int W = widths[r], H = heights[r];
        const float sx = ((float) 16 / (float) W) * 2.0f, sy = ((float) 16 / (float) H) * 2.0f;
        const float tpsx = (float) W / 16.0f, tpsy = (float) H / 16.0f;
        const int tilesX = (W + 15) / 16, tilesY = (H + 15) / 16;
        Ell e;
        e.cx = -1.1f + 2.2f * U(rng);
        e.cy = -1.1f + 2.2f * U(rng);
        // extents: log-uniform over 1e-4 .. 2 clip units, anisotropy up to 1e3 (ex >= ey as the clip stage stores them)
        const float big = std::exp(std::log(1e-4f) + U(rng) * (std::log(2.0f) - std::log(1e-4f)));
        const float ratio = std::exp(U(rng) * std::log(1000.0f));
        e.ex = big;
        e.ey = std::max(big / ratio, 3e-5f);
        const float ax = U(rng) * 2.0f - 1.0f, ay = U(rng) * 2.0f - 1.0f;
        const float inv = 1.0f / std::sqrt(ax * ax + ay * ay);
        e.c = ax * inv;
        e.s = ay * inv;
        if (i % 97 == 0) { e.c = 1.0f; e.s = 0.0f; }  // axis-aligned
        if (i % 89 == 0) { e.c = 0.0f; e.s = -1.0f; }
        // AABB (gs_aabb.h)
        const float rX = e.c * e.ex, rY = e.s * e.ex, uX = e.s * e.ey, uY = -e.c * e.ey;
        float mnx = 1e12f, mny = 1e12f, mxx = -1e12f, mxy = -1e12f;
        for (int j = 0; j != 4; ++j)
        {
            const int b = j & 1, t = j >> 1;
            const float rr = (float) (b ^ t) * 2.0f - 1.0f, uu = (float) t * 2.0f - 1.0f;
            const float vx = rX * rr + uX * uu, vy = rY * rr + uY * uu;
            mnx = std::min(mnx, vx), mny = std::min(mny, vy), mxx = std::max(mxx, vx), mxy = std::max(mxy, vy);
        }
        mnx += e.cx, mny += e.cy, mxx += e.cx, mxy += e.cy;
        auto cl = [](int v, int lo, int hi) { return std::min(std::max(v, lo), hi); };
        const int x0 = cl((int) std::floor((mnx + 1.0f) * 0.5f * tpsx), 0, tilesX), y0 = cl((int) std::floor((mny + 1.0f) * 0.5f * tpsy), 0, tilesY);
        const int x1 = cl((int) std::ceil((mxx + 1.0f) * 0.5f * tpsx), 0, tilesX), y1 = cl((int) std::ceil((mxy + 1.0f) * 0.5f * tpsy), 0, tilesY);
        if (x1 <= x0 || y1 <= y0) continue;
        const Hoisted h = hoist(e, sx, sy, (int) (i * 2654435761u >> 7));
        // walk at most 4096 tiles of the box (a stride over larger ones)
        const long long area = (long long) (x1 - x0) * (y1 - y0);
        const long long stride = std::max<long long>(1, area / 4096);
        for (long long l = 0; l < area; l += stride)
        {
            const int gx = x0 + (int) (l % (x1 - x0)), gy = y0 + (int) (l / (x1 - x0));
            const float minX = (float) gx * sx - 1.0f, minY = (float) gy * sy - 1.0f;
            bool inside;
            const float me = exactMin(e, minX, minY, minX + sx, minY + sy, inside);
            float S, A;
            const float mf = fastMin(e, h, minX, minY, sx, sy, S, A);
            ++tests;
            const double diff = std::fabs((double) mf - (double) me);
            if (!(me == me) || !(mf == mf) || std::isinf(me) || std::isinf(mf)) continue;
            const double s1 = 1.0 + S, s2 = s1 * s1;
            {
                const float ux = h.m00 * sx, uy = h.m10 * sx, vx = h.m01 * sy, vy = h.m11 * sy;
                const float uvL1 = std::fabs(ux) + std::fabs(uy) + std::fabs(vx) + std::fabs(vy);
                const float reach = 1.05f * std::max(e.ex, e.ey);
                const float aStar = (std::fabs(h.m00) + std::fabs(h.m10)) * (reach + sx) + (std::fabs(h.m01) + std::fabs(h.m11)) * (reach + sy) + uvL1;
                double guardSplat = kGuard * (1.0 + aStar);
                if (!(guardSplat <= 0.05)) guardSplat = INFINITY;
                const float adx = std::fabs(minX - e.cx), ady = std::fabs(minY - e.cy);
                const float aTile = 1.0f + uvL1 + ((std::fabs(h.m00) + std::fabs(h.m10)) * adx + (std::fabs(h.m01) + std::fabs(h.m11)) * ady);
                const double dev = std::fabs((double) mf - 1.0);
                const bool inSplat = !(dev > guardSplat), inTile = !(dev > kGuard * aTile);
                insideSplatGuard += inSplat;
                insideBothGuards += inSplat && inTile;
                if ((me < 1.0f) != (mf < 1.0f) && !inside && !(inSplat && inTile)) ++straddleOutside;
                const float mv = fastMinVisible(e, h, minX, minY, sx, sy);
                if (!inside)
                {
                    // (a tile that holds the centre by the reference's arithmetic is a hit whatever the value: decided per splat)
                    const double devV = std::fabs((double) mv - 1.0);
                    const bool inSplatV = !(devV > guardSplat), inTileV = !(devV > kGuard * aTile);
                    visInsideBoth += inSplatV && inTileV;
                    if ((me < 1.0f) != (mv < 1.0f) && !(inSplatV && inTileV)) ++visStraddleOutside;
                    if (mv != mf) ++visDiffersFromFast;
                    if (std::min(me, mv) < 4.0f && std::isfinite(mv))
                    {
                        visWorstVsFastRel = std::max(visWorstVsFastRel, std::fabs((double) mv - (double) mf) / aTile);
                        visWorstVsExactRelA = std::max(visWorstVsExactRelA, std::fabs((double) mv - (double) me) / aTile);
                    }
                }
            }
            for (int k = 0; k != 3; ++k)
            {
                const double kk[3] = {2e-6, 4e-6, 1e-5};
                if (std::fabs(mf - 1.0) <= kk[k] * s2) ++guardHits[k];
            }
            if (std::fabs(mf - 1.0) <= 1e-3) ++guardHits[3];
            if (std::fabs(me - 1.0f) < 0.05f || std::fabs(mf - 1.0f) < 0.05f)
            {
                int bin = 0;
                for (float edge = 2.0f; bin + 1 < kBins && S >= edge; edge *= 2.0f) ++bin;
                binWorst[bin] = std::max(binWorst[bin], diff);
                ++binCount[bin];
                int binA = 0;
                for (float edge = 2.0f; binA + 1 < kBins && A >= edge; edge *= 2.0f) ++binA;
                binWorstA[binA] = std::max(binWorstA[binA], diff);
                ++binCountA[binA];
            }
            if (std::min(me, mf) < 4.0f)
            {
                ++near;
                worstAbs = std::max(worstAbs, diff);
                worstRelS = std::max(worstRelS, diff / s1);
                worstRelS2 = std::max(worstRelS2, diff / s2);
            }
            if ((me < 1.0f) != (mf < 1.0f) && !inside)
            {
                ++straddle;
                worstStraddleRelS = std::max(worstStraddleRelS, std::fabs(mf - 1.0) / s1);
                worstStraddleRelS2 = std::max(worstStraddleRelS2, std::fabs(mf - 1.0) / s2);
            }
        }
    }
    printf("tests %lld, near the decision %lld, decisions that differ (centre outside) %lld\n", tests, near, straddle);
    printf("worst |fast - exact| near the decision: abs %.3e, /(1+S) %.3e, /(1+S)^2 %.3e\n", worstAbs, worstRelS, worstRelS2);
    printf("worst |fast - 1| among differing decisions: /(1+S) %.3e, /(1+S)^2 %.3e\n", worstStraddleRelS, worstStraddleRelS2);
    printf("tests inside the guard: k=2e-6 %.4f %%, k=4e-6 %.4f %%, k=1e-5 %.4f %% (x (1+S)^2); flat 1e-3 %.4f %%\n",
           100.0 * guardHits[0] / tests, 100.0 * guardHits[1] / tests, 100.0 * guardHits[2] / tests, 100.0 * guardHits[3] / tests);
    printf("kernel's guard (%.1e (1 + A)): tests inside the per-splat filter %.4f %%, inside both (decided exactly) %.4f %%; differing decisions OUTSIDE the guard: %lld\n",
           kGuard, 100.0 * insideSplatGuard / tests, 100.0 * insideBothGuards / tests, straddleOutside);
    printf("nearest-two-edges formulation: value differs from the five-term value in %lld tests, worst |two-edge - five-term| %.3e (1 + A), worst "
           "|two-edge - exact| near the decision %.3e (1 + A); inside both guards %.4f %%; differing decisions OUTSIDE the guard: %lld\n",
           visDiffersFromFast, visWorstVsFastRel, visWorstVsExactRelA, 100.0 * visInsideBoth / tests, visStraddleOutside);
    printf("decision values within 5 %% of 1, by S = |q0|_1 + |u|_1 + |v|_1:\n");
    for (int b = 0; b != kBins; ++b)
    {
        const double lo = b == 0 ? 0.0 : std::ldexp(1.0, b), hi = std::ldexp(1.0, b + 1);
        printf("  S in [%5.0f, %5.0f%s: %9lld tests, worst |fast - exact| %.3e = %.3e (1+S)^2 = %.3e (1+S)\n", lo, hi, b + 1 == kBins ? "+)" : ") ", binCount[b],
               binWorst[b], binWorst[b] / ((1 + lo) * (1 + lo)), binWorst[b] / (1 + lo));
    }
    printf("the same by A = sum of the absolute terms of the corner's two dot products + |u|_1 + |v|_1 (cancellation-aware):\n");
    for (int b = 0; b != kBins; ++b)
    {
        const double lo = b == 0 ? 0.0 : std::ldexp(1.0, b), hi = std::ldexp(1.0, b + 1);
        printf("  A in [%5.0f, %5.0f%s: %9lld tests, worst |fast - exact| %.3e = %.3e (1+A)\n", lo, hi, b + 1 == kBins ? "+)" : ") ", binCountA[b], binWorstA[b],
               binWorstA[b] / (1 + lo));
    }
    return 0;
}
