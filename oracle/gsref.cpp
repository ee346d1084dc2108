/*
 * gsref.cpp -- CPU oracle: serial restatement of the reference's per-frame hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see gsref.h).  Plain C++17, no third-party headers: the handful of
 * glm operations the reference leans on are restated here with the operand order glm 1.0.1 uses,
 * each citing the glm file it follows.  Cites of the form ".cu:NNN" are lines of
 * /root/reference/src/GaussianRender.cu.
 *
 * Build: make -C oracle   (g++ -O2 -ffp-contract=off; no -ffast-math, so every float op is IEEE).
 */
#include "gsref.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <numeric>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

// The reference's compile-time switch `#define EPANECHNIKOV_KERNEL` (GaussianRender.cu:8-9): footprint extent sqrt(7) sigma
// instead of 3 sigma (.cu:298-302) and density 1 - q/7 instead of exp(-q/2) (.cu:977-981).  A run-time switch here.
static int g_epanechnikov = 0;
extern "C" void gsref_set_footprint(int epanechnikov) { g_epanechnikov = epanechnikov ? 1 : 0; }
extern "C" int gsref_get_footprint(void) { return g_epanechnikov; }

namespace
{

// ----------------------------------------------------------------------------------------------
// glm restatements
// ----------------------------------------------------------------------------------------------
struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
// m[col][row], as glm.
struct M3 { float m[3][3]; };
struct M4 { float m[4][4]; };

// glm/detail/func_common.inl: min(x,y) = (y < x) ? y : x ; max(x,y) = (x < y) ? y : x
inline float gmin(float x, float y) { return (y < x) ? y : x; }
inline float gmax(float x, float y) { return (x < y) ? y : x; }
inline int gmini(int x, int y) { return (y < x) ? y : x; }
inline int gmaxi(int x, int y) { return (x < y) ? y : x; }
inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
inline int gclampi(int x, int lo, int hi) { return gmini(gmaxi(x, lo), hi); }
// glm/ext/vector_common.inl:102 single-argument clamp == clamp(x, 0, 1)
inline float gclamp01(float x) { return gclamp(x, 0.0f, 1.0f); }
// glm step(edge, x) = mix(1, 0, x < edge)
inline float gstep(float edge, float x) { return (x < edge) ? 0.0f : 1.0f; }
// glm mix(x, y, a) = x * (1 - a) + y * a
inline float gmix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
inline float dot2(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
// CUDA __saturatef: clamp to [0,1], NaN -> 0.
inline float saturatef(float x) { return (x != x) ? 0.0f : gmin(gmax(x, 0.0f), 1.0f); }

// float -> int32 as the GPU does it (cvt.rzi.s32.f32 / v_cvt_i32_f32): truncation, saturating, NaN -> 0.
inline int32_t f2i_sat(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return std::numeric_limits<int32_t>::max();
    if (v <= -2147483648.0f) return std::numeric_limits<int32_t>::min();
    return (int32_t) v;
}

// glm/detail/type_mat3x3.inl:486-520 operator*(mat3, mat3)
M3 mul(const M3& a, const M3& b)
{
    M3 r;
    for (int c = 0; c != 3; ++c)
        for (int row = 0; row != 3; ++row)
            r.m[c][row] = a.m[0][row] * b.m[c][0] + a.m[1][row] * b.m[c][1] + a.m[2][row] * b.m[c][2];
    return r;
}

M3 transpose(const M3& a)
{
    M3 r;
    for (int c = 0; c != 3; ++c)
        for (int row = 0; row != 3; ++row) r.m[c][row] = a.m[row][c];
    return r;
}

// glm/detail/type_mat4x4.inl:536-572 operator*(mat4, vec4): (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
V4 mul(const M4& m, V4 v)
{
    float r[4];
    for (int row = 0; row != 4; ++row)
    {
        float add0 = m.m[0][row] * v.x + m.m[1][row] * v.y;
        float add1 = m.m[2][row] * v.z + m.m[3][row] * v.w;
        r[row] = add0 + add1;
    }
    return V4{r[0], r[1], r[2], r[3]};
}

M3 upper3(const M4& a)
{
    M3 r;
    for (int c = 0; c != 3; ++c)
        for (int row = 0; row != 3; ++row) r.m[c][row] = a.m[c][row];
    return r;
}

// glm/gtc/quaternion.inl:47-72 mat3_cast
M3 mat3Cast(float qx, float qy, float qz, float qw)
{
    M3 r;
    float qxx = qx * qx, qyy = qy * qy, qzz = qz * qz;
    float qxz = qx * qz, qxy = qx * qy, qyz = qy * qz;
    float qwx = qw * qx, qwy = qw * qy, qwz = qw * qz;
    r.m[0][0] = 1.0f - 2.0f * (qyy + qzz);
    r.m[0][1] = 2.0f * (qxy + qwz);
    r.m[0][2] = 2.0f * (qxz - qwy);
    r.m[1][0] = 2.0f * (qxy - qwz);
    r.m[1][1] = 1.0f - 2.0f * (qxx + qzz);
    r.m[1][2] = 2.0f * (qyz + qwx);
    r.m[2][0] = 2.0f * (qxz + qwy);
    r.m[2][1] = 2.0f * (qyz - qwx);
    r.m[2][2] = 1.0f - 2.0f * (qxx + qyy);
    return r;
}

// glm/gtc/quaternion.inl:81-123 quat_cast(mat3) -> (x,y,z,w)
V4 quatCast(const M3& m)
{
    float fourX = m.m[0][0] - m.m[1][1] - m.m[2][2];
    float fourY = m.m[1][1] - m.m[0][0] - m.m[2][2];
    float fourZ = m.m[2][2] - m.m[0][0] - m.m[1][1];
    float fourW = m.m[0][0] + m.m[1][1] + m.m[2][2];
    int biggest = 0;
    float fourBiggest = fourW;
    if (fourX > fourBiggest) { fourBiggest = fourX; biggest = 1; }
    if (fourY > fourBiggest) { fourBiggest = fourY; biggest = 2; }
    if (fourZ > fourBiggest) { fourBiggest = fourZ; biggest = 3; }
    float biggestVal = std::sqrt(fourBiggest + 1.0f) * 0.5f;
    float mult = 0.25f / biggestVal;
    switch (biggest)
    {
    case 0: return V4{(m.m[1][2] - m.m[2][1]) * mult, (m.m[2][0] - m.m[0][2]) * mult, (m.m[0][1] - m.m[1][0]) * mult, biggestVal};
    case 1: return V4{biggestVal, (m.m[0][1] + m.m[1][0]) * mult, (m.m[2][0] + m.m[0][2]) * mult, (m.m[1][2] - m.m[2][1]) * mult};
    case 2: return V4{(m.m[0][1] + m.m[1][0]) * mult, biggestVal, (m.m[1][2] + m.m[2][1]) * mult, (m.m[2][0] - m.m[0][2]) * mult};
    default: return V4{(m.m[2][0] + m.m[0][2]) * mult, (m.m[1][2] + m.m[2][1]) * mult, biggestVal, (m.m[0][1] - m.m[1][0]) * mult};
    }
}

V3 cross(V3 a, V3 b) { return V3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// glm normalize(v) = v * inversesqrt(dot(v,v)), inversesqrt(x) = 1 / sqrt(x)
V3 normalize3(V3 v)
{
    float inv = 1.0f / std::sqrt(dot3(v, v));
    return V3{v.x * inv, v.y * inv, v.z * inv};
}
V2 normalize2(V2 v)
{
    float inv = 1.0f / std::sqrt(dot2(v, v));
    return V2{v.x * inv, v.y * inv};
}

// ----------------------------------------------------------------------------------------------
// Resolution generalisation (DESIGN.md "Resolution rules"); at 1024x1024 every value equals the
// reference's compile-time constant (Consts.h:4-8).
// ----------------------------------------------------------------------------------------------
constexpr int k_TileSize = 16;
constexpr int k_ChunkSize = 32; // .cu:439 k_WarpSize: raster chunk / termination cadence

struct ScreenInfo
{
    int width, height, tilesX, tilesY, totalTiles;
    float tilesPerScreenX, tilesPerScreenY; // un-rounded W/16, H/16 (.cu:537 uses 64)
    float tileClipSizeX, tileClipSizeY;     // .cu:656-657
    float pixelClipSizeX, pixelClipSizeY;   // .cu:939
    float traceBump;                        // .cu:271-274
};

ScreenInfo makeScreen(int width, int height)
{
    ScreenInfo s;
    s.width = width;
    s.height = height;
    s.tilesX = (width + k_TileSize - 1) / k_TileSize;
    s.tilesY = (height + k_TileSize - 1) / k_TileSize;
    s.totalTiles = s.tilesX * s.tilesY;
    s.tilesPerScreenX = (float) width / (float) k_TileSize;
    s.tilesPerScreenY = (float) height / (float) k_TileSize;
    s.tileClipSizeX = ((float) k_TileSize / (float) width) * 2.0f;
    s.tileClipSizeY = ((float) k_TileSize / (float) height) * 2.0f;
    s.pixelClipSizeX = 2.0f / (float) width;
    s.pixelClipSizeY = 2.0f / (float) height;
    const float k_Pi = 3.14159265359f;
    s.traceBump = (1.0f / k_Pi) * s.pixelClipSizeX * s.pixelClipSizeY;
    return s;
}

M4 loadM4(const float* p)
{
    M4 r;
    for (int c = 0; c != 4; ++c)
        for (int row = 0; row != 4; ++row) r.m[c][row] = p[c * 4 + row];
    return r;
}

double nowMs()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ----------------------------------------------------------------------------------------------
// .cu:351-436 geometry helpers
// ----------------------------------------------------------------------------------------------
struct Ellipse { V2 center, extent, cosSin; };
struct Rect { V2 min, max; };

// .cu:351-358
inline V2 convertToEllipseCoordinates(const Ellipse& e, V2 p)
{
    V2 d{p.x - e.center.x, p.y - e.center.y};
    return V2{(d.x * e.cosSin.x + d.y * e.cosSin.y) / e.extent.x, (d.y * e.cosSin.x - d.x * e.cosSin.y) / e.extent.y};
}

// .cu:361-372
inline bool intersectsUnitCircle(V2 v0, V2 v1)
{
    V2 delta{v1.x - v0.x, v1.y - v0.y};
    float lengthSqr = dot2(delta, delta);
    float t = saturatef(dot2(V2{-v0.x, -v0.y}, delta) / lengthSqr);
    V2 proj{v0.x + t * (v1.x - v0.x), v0.y + t * (v1.y - v0.y)};
    return dot2(proj, proj) < 1.0f;
}

// .cu:375-405
inline bool ellipseRectOverlap(const Ellipse& e, const Rect& r)
{
    bool overlaps = e.center.x > r.min.x && e.center.x < r.max.x && e.center.y > r.min.y && e.center.y < r.max.y;
    // Rect::getCenter (GaussianRender.cuh:33-36): min + (max - min) * 0.5
    V2 c{r.min.x + (r.max.x - r.min.x) * 0.5f, r.min.y + (r.max.y - r.min.y) * 0.5f};
    V2 cl = convertToEllipseCoordinates(e, c);
    overlaps |= dot2(cl, cl) < 1.0f;
    V2 p0 = convertToEllipseCoordinates(e, r.min);
    V2 p1 = convertToEllipseCoordinates(e, V2{r.max.x, r.min.y});
    V2 p2 = convertToEllipseCoordinates(e, r.max);
    V2 p3 = convertToEllipseCoordinates(e, V2{r.min.x, r.max.y});
    overlaps |= intersectsUnitCircle(p0, p1);
    overlaps |= intersectsUnitCircle(p1, p2);
    overlaps |= intersectsUnitCircle(p2, p3);
    overlaps |= intersectsUnitCircle(p3, p0);
    return overlaps;
}

// .cu:408-436 (Ellipse::getPrincipalAxis / getMinorAxis: GaussianRender.cuh:45-52)
inline Rect getAABBRect(const Ellipse& e)
{
    V2 right{e.cosSin.x * e.extent.x, e.cosSin.y * e.extent.x};
    V2 up{e.cosSin.y * e.extent.y, -e.cosSin.x * e.extent.y};
    Rect rect;
    rect.min = V2{1e12f, 1e12f};
    rect.max = V2{-1e12f, -1e12f};
    for (int j = 0; j != 4; ++j)
    {
        int bottomBit = j & 1;
        int topBit = j >> 1;
        float r = (float) (bottomBit ^ topBit) * 2.0f - 1.0f;
        float u = (float) topBit * 2.0f - 1.0f;
        V2 v{right.x * r + up.x * u, right.y * r + up.y * u};
        rect.min = V2{gmin(rect.min.x, v.x), gmin(rect.min.y, v.y)};
        rect.max = V2{gmax(rect.max.x, v.x), gmax(rect.max.y, v.y)};
    }
    rect.min = V2{rect.min.x + e.center.x, rect.min.y + e.center.y};
    rect.max = V2{rect.max.x + e.center.x, rect.max.y + e.center.y};
    return rect;
}

} // namespace

// ----------------------------------------------------------------------------------------------
// encode / decode / key
// ----------------------------------------------------------------------------------------------
extern "C" uint32_t gsref_encode_vec4(const float v[4])
{
    // PlyParser.cpp:155-165: clamp01 then truncating 8-bit pack, x in bits 31..24.
    float c[4];
    for (int i = 0; i != 4; ++i) c[i] = gclamp01(v[i]);
    return ((uint32_t) (c[0] * 255.0f) << 24u) | ((uint32_t) (c[1] * 255.0f) << 16u) | ((uint32_t) (c[2] * 255.0f) << 8u) |
           ((uint32_t) (c[3] * 255.0f));
}

extern "C" void gsref_decode_vec4(uint32_t v, float out[4])
{
    // .cu:55-58
    out[0] = (float) ((v >> 24u) & 0xFF) / 255.0f;
    out[1] = (float) ((v >> 16u) & 0xFF) / 255.0f;
    out[2] = (float) ((v >> 8u) & 0xFF) / 255.0f;
    out[3] = (float) (v & 0xFF) / 255.0f;
}

extern "C" uint64_t gsref_get_key(int32_t tileIndex, float clipDepth)
{
    // .cu:446-454.  (float) 0xFFFFFFFF rounds to 2^32, so depth == far overflows the cast;
    // the GPU conversion saturates, restated explicitly here.
    float scaled = gclamp01((clipDepth + 1.0f) * 0.5f) * 4294967296.0f;
    uint32_t q;
    if (scaled != scaled) q = 0;
    else if (scaled >= 4294967296.0f) q = 0xFFFFFFFFu;
    else q = (uint32_t) scaled;
    return ((uint64_t) (uint32_t) tileIndex << 32) | (uint64_t) q;
}

// ----------------------------------------------------------------------------------------------
// Spherical harmonics (.cu:62-155, generated by sh_gen.py)
// ----------------------------------------------------------------------------------------------
extern "C" void gsref_spherical_harmonics(int l, const float dir[3], const float* sh, int stride, float rgbOut[3])
{
    const float x = dir[0], y = dir[1], z = dir[2];
    float res[3] = {0.0f, 0.0f, 0.0f};
    // Each level is summed left to right into a temporary and then added to the result, as the
    // generated expression `result += b0 * sh_a + b1 * sh_b + ...` does (.cu:72-73,82-85,...).
    int k = 0;
    float level[3];
    bool first = true;
    auto term = [&](float basis) {
        // coefficient k of channel c lives at sh[(3k+c)*stride] (.cu:70,78-80,...)
        for (int c = 0; c != 3; ++c)
        {
            float v = basis * sh[(size_t) (3 * k + c) * stride];
            level[c] = first ? v : level[c] + v;
        }
        first = false;
        ++k;
    };
    auto flush = [&]() {
        for (int c = 0; c != 3; ++c) res[c] += level[c];
        first = true;
    };
    term(0.282094792F);
    flush();
    if (l > 0)
    {
        term(0.488602512F * y);
        term(0.488602512F * z);
        term(0.488602512F * x);
        flush();
        if (l > 1)
        {
            const float xx = x * x, yy = y * y;
            term(1.09254843F * x * y);
            term(1.09254843F * y * z);
            term(-0.946174696F * xx - 0.946174696F * yy + 0.630783131F);
            term(1.09254843F * x * z);
            term(0.546274215F * (x - y) * (x + y));
            flush();
            if (l > 2)
            {
                term(0.295021795F * y * (6.0F * xx - 2.0F * yy));
                term(2.89061144F * x * y * z);
                term(3.6563664F * y * (-0.625F * xx - 0.625F * yy + 0.5F));
                term(0.373176333F * z * (-5.0F * xx - 5.0F * yy + 2.0F));
                term(0.457045799F * x * (-5.0F * xx - 5.0F * yy + 4.0F));
                term(1.44530572F * z * (x - y) * (x + y));
                term(0.59004359F * x * (xx - 3.0F * yy));
                flush();
                if (l > 3)
                {
                    term(2.50334294F * x * y * (xx - yy));
                    term(0.295021795F * y * z * (18.0F * xx - 6.0F * yy));
                    term(1.26156626F * x * y * (-5.25F * xx - 5.25F * yy + 4.5F));
                    term(1.78412412F * y * z * (-2.625F * xx - 2.625F * yy + 1.5F));
                    term(7.40498828F * xx * yy - 4.23142188F * xx + 3.70249414F * xx * xx - 4.23142188F * yy +
                         3.70249414F * yy * yy + 0.846284375F);
                    term(0.669046544F * x * z * (-7.0F * xx - 7.0F * yy + 4.0F));
                    term(-0.473087348F * (x - y) * (x + y) * (7.0F * xx + 7.0F * yy - 6.0F));
                    term(1.77013077F * x * z * (xx - 3.0F * yy));
                    term(-3.75501441F * xx * yy + 0.625835735F * xx * xx + 0.625835735F * yy * yy);
                    flush();
                }
            }
        }
    }
    for (int c = 0; c != 3; ++c) rgbOut[c] = gclamp01(res[c] + 0.5f); // .cu:154
}

extern "C" size_t gsref_realigned_sh_size(int groupSize, int shCount, int splatCount)
{
    int groupCount = (int) std::ceil((float) splatCount / (float) groupSize); // Demo.cpp:172
    return (size_t) groupCount * groupSize * shCount;
}

extern "C" void gsref_realign_spherical_harmonics(const float* src, float* dst, int groupSize, int shCount, int splatCount)
{
    // Demo.cpp:160-194. dst must hold gsref_realigned_sh_size() floats; the tail of the last group is zero.
    size_t total = gsref_realigned_sh_size(groupSize, shCount, splatCount);
    std::memset(dst, 0, total * sizeof(float));
    int groupCount = (int) (total / ((size_t) groupSize * shCount));
    for (int grp = 0; grp != groupCount; ++grp)
    {
        size_t start = (size_t) grp * groupSize * shCount;
        int thisGroupSize = std::min(groupSize, splatCount - grp * groupSize);
        for (int i = 0; i != shCount; ++i)
            for (int k = 0; k != thisGroupSize; ++k) dst[start + (size_t) groupSize * i + k] = src[start + (size_t) shCount * k + i];
    }
}

extern "C" void gsref_evaluate_spherical_harmonics(int32_t count, const float* position4, const float* sh, int shDegree,
                                                   int shCount, const float cameraPosition[3], float* color4)
{
    // .cu:158-182 with blockDim.x == 256 (.cu:188) which is also the realign group size (Demo.cpp:294).
    constexpr int k_Group = 256;
    for (int32_t index = 0; index != count; ++index)
    {
        const float* p = position4 + 4 * (size_t) index;
        float opacity = p[3]; // .cu:168
        V3 d{cameraPosition[0] - p[0], cameraPosition[1] - p[1], cameraPosition[2] - p[2]};
        V3 rayDir = normalize3(d); // .cu:171
        size_t shIdx = (size_t) (index / k_Group) * k_Group * shCount + (size_t) (index % k_Group); // .cu:174
        float dir[3] = {rayDir.x, rayDir.y, rayDir.z};
        float rgb[3];
        gsref_spherical_harmonics(shDegree, dir, sh + shIdx, k_Group, rgb);
        float* c = color4 + 4 * (size_t) index;
        c[0] = rgb[0];
        c[1] = rgb[1];
        c[2] = rgb[2];
        c[3] = opacity; // .cu:180
    }
}

// ----------------------------------------------------------------------------------------------
// Clip data (.cu:204-329)
// ----------------------------------------------------------------------------------------------
extern "C" void gsref_evaluate_splat_clip_data(int32_t count, const float* position4, const float* scaleAndRotation4,
                                               const GsRefCameraData* camera, int width, int height, float* clipXY2,
                                               float* clipZ, float* screenEllipse4, float* conic4)
{
    const ScreenInfo scr = makeScreen(width, height);
    const M4 view = loadM4(camera->view);
    const M3 view3 = upper3(view);
    const V2 fovCotangent{camera->fovCotangent[0], camera->fovCotangent[1]};
    const V2 depthScaleBias{camera->depthScaleBias[0], camera->depthScaleBias[1]};

    for (int32_t index = 0; index != count; ++index)
    {
        const float* pd = position4 + 4 * (size_t) index;
        const float* sr = scaleAndRotation4 + 4 * (size_t) index;

        // .cu:215-224 covariance
        M3 scale3x3{};
        scale3x3.m[0][0] = sr[0];
        scale3x3.m[1][1] = sr[1];
        scale3x3.m[2][2] = sr[2];
        uint32_t bits;
        std::memcpy(&bits, &sr[3], sizeof(bits));
        float dec[4];
        gsref_decode_vec4(bits, dec);
        V4 rv{dec[0] * 2.0f - 1.0f, dec[1] * 2.0f - 1.0f, dec[2] * 2.0f - 1.0f, dec[3] * 2.0f - 1.0f};
        // quat::wxyz(rv.w, rv.x, rv.y, rv.z); not renormalised (.cu:220-221)
        M3 RS = mul(mat3Cast(rv.x, rv.y, rv.z, rv.w), scale3x3);
        M3 splatCovariance = mul(RS, transpose(RS));

        // .cu:227
        V4 vp4 = mul(view, V4{pd[0], pd[1], pd[2], 1.0f});
        V3 viewPosition{vp4.x, vp4.y, vp4.z};

        // .cu:236-244
        float zRcp = 1.0f / viewPosition.z;
        float zRcpSqr = zRcp * zRcp;
        float scaleX = -fovCotangent.x * zRcp;
        float scaleY = -fovCotangent.y * zRcp;
        V2 fovCotTimesTrs{fovCotangent.x * viewPosition.x, fovCotangent.y * viewPosition.y};
        float shearX = fovCotTimesTrs.x * zRcpSqr;
        float shearY = fovCotTimesTrs.y * zRcpSqr;
        float translationX = -fovCotTimesTrs.x * zRcp;
        float translationY = -fovCotTimesTrs.y * zRcp;

        // .cu:248-259
        M4 affine{};
        for (int i = 0; i != 4; ++i) affine.m[i][i] = 1.0f;
        affine.m[0][0] = scaleX;
        affine.m[1][1] = scaleY;
        affine.m[2][2] = depthScaleBias.x;
        affine.m[2][0] = shearX;
        affine.m[2][1] = shearY;
        affine.m[3][0] = translationX;
        affine.m[3][1] = translationY;
        affine.m[3][2] = depthScaleBias.y;

        // .cu:262-265
        M3 viewProjection = mul(upper3(affine), view3);
        M3 clipCovariance = mul(mul(viewProjection, transpose(splatCovariance)), transpose(viewProjection));
        V4 cp4 = mul(affine, V4{viewPosition.x, viewPosition.y, viewPosition.z, 1.0f});
        V3 clipPosition{cp4.x, cp4.y, cp4.z};

        // .cu:271-276
        clipCovariance.m[0][0] += scr.traceBump;
        clipCovariance.m[1][1] += scr.traceBump;

        // .cu:279-288
        float det = clipCovariance.m[0][0] * clipCovariance.m[1][1] - clipCovariance.m[1][0] * clipCovariance.m[1][0];
        float mid = 0.5f * (clipCovariance.m[0][0] + clipCovariance.m[1][1]);
        const float epsilon = 1e-12f;
        float radius = std::sqrt(gmax(epsilon, mid * mid - det));
        float lambda0 = mid + radius;
        float lambda1 = gmax(0.0f, mid - radius);

        // .cu:290 normalize(vec2(b, lambda0 - a)); eigenVector1 (.cu:292) is dead code.
        V2 eigenVector0 = normalize2(V2{clipCovariance.m[1][0], lambda0 - clipCovariance.m[0][0]});

        // .cu:295-301
        float angle = std::atan2(eigenVector0.y, eigenVector0.x);
        // .cu:298-302: 3 sigma for the Gaussian footprint, sqrt(7) with EPANECHNIKOV_KERNEL
        const float extentFactor = g_epanechnikov ? std::sqrt(7.0f) : 3.0f;
        V2 extent{std::sqrt(lambda0) * extentFactor, std::sqrt(lambda1) * extentFactor};

        // .cu:305-307
        float invDet = 1.0f / gmax(epsilon, det);
        V3 conic{clipCovariance.m[1][1] * invDet, -clipCovariance.m[1][0] * invDet, clipCovariance.m[0][0] * invDet};

        // .cu:316-321
        V3 edge{gstep(-1.0f, clipPosition.x) * gstep(clipPosition.x, 1.0f), gstep(-1.0f, clipPosition.y) * gstep(clipPosition.y, 1.0f),
                gstep(-1.0f, clipPosition.z) * gstep(clipPosition.z, 1.0f)};
        float isVisible = edge.x * edge.y * edge.z * gstep(0.0f, lambda1);
        clipPosition = V3{gmix(-128.0f, clipPosition.x, isVisible), gmix(-128.0f, clipPosition.y, isVisible),
                          gmix(-128.0f, clipPosition.z, isVisible)};
        extent = V2{extent.x * isVisible, extent.y * isVisible};

        // .cu:324-327
        clipXY2[2 * (size_t) index + 0] = clipPosition.x;
        clipXY2[2 * (size_t) index + 1] = clipPosition.y;
        clipZ[index] = clipPosition.z;
        float* se = screenEllipse4 + 4 * (size_t) index;
        se[0] = std::cos(angle);
        se[1] = std::sin(angle);
        se[2] = extent.x;
        se[3] = extent.y;
        float* cn = conic4 + 4 * (size_t) index;
        cn[0] = conic.x;
        cn[1] = conic.y;
        cn[2] = conic.z;
        cn[3] = 0.0f;
    }
}

// ----------------------------------------------------------------------------------------------
// Tile list (.cu:456-770): the set { (key(tile, z_i), i) : tile in AABB_i and overlap(ellipse_i, tile) }
// ----------------------------------------------------------------------------------------------
extern "C" int64_t gsref_build_tile_list(int32_t count, const float* clipXY2, const float* clipZ, const float* screenEllipse4,
                                         int width, int height, int tileRowBegin, int tileRowEnd, uint64_t* keys,
                                         int32_t* values, int64_t capacity)
{
    const ScreenInfo scr = makeScreen(width, height);
    if (tileRowBegin == 0 && tileRowEnd == 0) tileRowEnd = scr.tilesY;
    int64_t emitted = 0;
    for (int32_t i = 0; i != count; ++i)
    {
        // .cu:526-534
        Ellipse e;
        e.center = V2{clipXY2[2 * (size_t) i], clipXY2[2 * (size_t) i + 1]};
        const float* se = screenEllipse4 + 4 * (size_t) i;
        e.cosSin = V2{se[0], se[1]};
        e.extent = V2{se[2], se[3]};
        Rect rect = getAABBRect(e);

        // .cu:537-551 (k_TilesPerScreen -> per-axis un-rounded tile counts, clamp to ceil)
        float tx0 = (rect.min.x + 1.0f) * 0.5f * scr.tilesPerScreenX;
        float ty0 = (rect.min.y + 1.0f) * 0.5f * scr.tilesPerScreenY;
        float tx1 = (rect.max.x + 1.0f) * 0.5f * scr.tilesPerScreenX;
        float ty1 = (rect.max.y + 1.0f) * 0.5f * scr.tilesPerScreenY;
        int x0 = gclampi(f2i_sat(std::floor(tx0)), 0, scr.tilesX);
        int y0 = gclampi(f2i_sat(std::floor(ty0)), 0, scr.tilesY);
        int x1 = gclampi(f2i_sat(std::ceil(tx1)), 0, scr.tilesX);
        int y1 = gclampi(f2i_sat(std::ceil(ty1)), 0, scr.tilesY);
        int w = x1 - x0;
        int h = y1 - y0;
        // The reference takes max(0, w*h); a rect with both sizes negative (NaN ellipse, SURVEY hard
        // part 6) yields w*h > 0 work items none of which can pass the overlap test, so it is skipped here.
        if (w <= 0 || h <= 0) continue;

        const float depth = clipZ[i];
        for (int local = 0; local != w * h; ++local)
        {
            // .cu:653-664
            int gx = x0 + local % w;
            int gy = y0 + local / w;
            if (gy < tileRowBegin || gy >= tileRowEnd) continue; // multi-GPU band (no reference analogue)
            V2 tileClip{(float) gx * scr.tileClipSizeX - 1.0f, (float) gy * scr.tileClipSizeY - 1.0f};
            Rect tileRect;
            tileRect.min = tileClip;
            tileRect.max = V2{tileClip.x + scr.tileClipSizeX, tileClip.y + scr.tileClipSizeY};
            if (ellipseRectOverlap(e, tileRect))
            {
                // .cu:675-677
                if (emitted < capacity)
                {
                    keys[emitted] = gsref_get_key(gy * scr.tilesX + gx, depth);
                    values[emitted] = i;
                }
                ++emitted;
            }
        }
    }
    return emitted;
}

// ----------------------------------------------------------------------------------------------
// Sort (.cu:806-855): stable LSD radix sort on bits [0,endBit) == std::stable_sort on the masked key.
// ----------------------------------------------------------------------------------------------
extern "C" int gsref_sort_end_bit(int width, int height)
{
    // .cu:818-819: 32 depth bits + bits needed for the largest tile index (12 at 4096 tiles).
    const ScreenInfo scr = makeScreen(width, height);
    int bits = 1;
    while ((1 << bits) < scr.totalTiles) ++bits;
    return 32 + bits;
}

extern "C" void gsref_sort_tile_list(int64_t size, uint64_t* keys, int32_t* values, int endBit)
{
    const uint64_t mask = (endBit >= 64) ? ~0ull : ((1ull << endBit) - 1ull);
    std::vector<int64_t> order((size_t) size);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return (keys[a] & mask) < (keys[b] & mask); });
    std::vector<uint64_t> k((size_t) size);
    std::vector<int32_t> v((size_t) size);
    for (int64_t i = 0; i != size; ++i)
    {
        k[i] = keys[order[i]];
        v[i] = values[order[i]];
    }
    std::memcpy(keys, k.data(), (size_t) size * sizeof(uint64_t));
    std::memcpy(values, v.data(), (size_t) size * sizeof(int32_t));
}

// ----------------------------------------------------------------------------------------------
// Tile ranges (.cu:857-888)
// ----------------------------------------------------------------------------------------------
extern "C" void gsref_evaluate_tile_ranges(int64_t size, const uint64_t* keys, int32_t* tileRange)
{
    if (size <= 0) return; // Demo.cpp:469 bypass
    for (int64_t index = 0; index != size; ++index)
    {
        if (index == 0)
        {
            tileRange[(keys[0] >> 32) * 2] = 0;
            tileRange[(keys[size - 1] >> 32) * 2 + 1] = (int32_t) size;
        }
        else
        {
            uint64_t prevTile = keys[index - 1] >> 32;
            uint64_t tile = keys[index] >> 32;
            if (tile != prevTile)
            {
                tileRange[prevTile * 2 + 1] = (int32_t) index;
                tileRange[tile * 2] = (int32_t) index;
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Raster (.cu:908-1016)
// ----------------------------------------------------------------------------------------------
// Optional diagnostics: list entries consumed per tile (caller-owned int32[tiles], zeroed by the caller).
static int32_t* g_consumedPerTile = nullptr;
extern "C" void gsref_set_consumed_per_tile_buffer(int32_t* buffer) { g_consumedPerTile = buffer; }

extern "C" int64_t gsref_rasterize_tiles(int width, int height, int tileRowBegin, int tileRowEnd, const int32_t* tileRange,
                                         const int32_t* values, const float* clipXY2, const float* conic4,
                                         const float* color4, uint8_t* backBuffer)
{
    const ScreenInfo scr = makeScreen(width, height);
    if (tileRowBegin == 0 && tileRowEnd == 0) tileRowEnd = scr.tilesY;
    int64_t consumed = 0;
    constexpr int k_Pixels = k_TileSize * k_TileSize;

    for (int tileY = tileRowBegin; tileY < tileRowEnd; ++tileY)
        for (int tileX = 0; tileX != scr.tilesX; ++tileX)
        {
            const int tile = tileY * scr.tilesX + tileX;
            // .cu:917-931
            int firstSplatIndex = tileRange[tile * 2];
            int splatsCount = tileRange[tile * 2 + 1] - firstSplatIndex;
            if (splatsCount == 0) continue;

            float colR[k_Pixels], colG[k_Pixels], colB[k_Pixels], trans[k_Pixels];
            float clipX[k_Pixels], clipY[k_Pixels];
            bool onScreen[k_Pixels];
            for (int t = 0; t != k_Pixels; ++t)
            {
                // .cu:934-943
                int px = tileX * k_TileSize + t % k_TileSize;
                int py = tileY * k_TileSize + t / k_TileSize;
                clipX[t] = (float) px * scr.pixelClipSizeX - 1.0f;
                clipY[t] = (float) py * scr.pixelClipSizeY - 1.0f;
                onScreen[t] = px < width && py < height;
                colR[t] = colG[t] = colB[t] = 0.0f;
                trans[t] = 1.0f;
            }

            while (splatsCount > 0)
            {
                int n = std::min(splatsCount, k_ChunkSize); // .cu:950
                consumed += n;
                if (g_consumedPerTile) g_consumedPerTile[tile] += n;
                for (int i = 0; i != n; ++i)
                {
                    // .cu:957-962
                    int src = values[firstSplatIndex + i];
                    const float cx = clipXY2[2 * (size_t) src], cy = clipXY2[2 * (size_t) src + 1];
                    const float* cn = conic4 + 4 * (size_t) src;
                    const float* sc = color4 + 4 * (size_t) src;
                    for (int t = 0; t != k_Pixels; ++t)
                    {
                        // .cu:972-984
                        float dx_ = clipX[t] - cx;
                        float dy_ = clipY[t] - cy;
                        float q = cn[0] * dx_ * dx_ + cn[2] * dy_ * dy_ + 2.0f * cn[1] * dx_ * dy_;
                        // .cu:977-981
                        float density = g_epanechnikov ? 1.0f - q / 7.0f : std::exp(-0.5f * q);
                        float alpha = sc[3] * saturatef(density);
                        colR[t] += sc[0] * trans[t] * alpha;
                        colG[t] += sc[1] * trans[t] * alpha;
                        colB[t] += sc[2] * trans[t] * alpha;
                        trans[t] *= (1.0f - alpha);
                    }
                }
                // .cu:988-998
                firstSplatIndex += n;
                splatsCount -= n;
                int alive = 0;
                // Pixels of a partial edge tile that fall outside the screen do not vote.
                for (int t = 0; t != k_Pixels; ++t) alive += (onScreen[t] && trans[t] > 0.02f) ? 1 : 0;
                if (alive == 0) break;
            }

            for (int t = 0; t != k_Pixels; ++t)
            {
                if (!onScreen[t]) continue;
                int px = tileX * k_TileSize + t % k_TileSize;
                int py = tileY * k_TileSize + t / k_TileSize;
                // .cu:1007-1015: float -> uchar conversion truncates and saturates on the GPU.
                auto quant = [](float c) -> uint8_t {
                    float v = c * 255.0f;
                    if (v != v) return 0;
                    if (v <= 0.0f) return 0;
                    if (v >= 255.0f) return 255;
                    return (uint8_t) v;
                };
                uint8_t* dst = backBuffer + 4 * ((size_t) py * width + px);
                dst[0] = quant(colR[t]);
                dst[1] = quant(colG[t]);
                dst[2] = quant(colB[t]);
                dst[3] = 255;
            }
        }
    return consumed;
}

// ----------------------------------------------------------------------------------------------
// Frame driver (Demo.cpp:394-479)
// ----------------------------------------------------------------------------------------------
extern "C" int gsref_render(int32_t count, const float* position4, const float* scaleAndRotation4, float* color4,
                            const float* sh, int shDegree, int shCount, const GsRefCameraData* camera, int width,
                            int height, uint8_t* backBuffer, GsRefFrameStats* stats)
{
    const ScreenInfo scr = makeScreen(width, height);
    GsRefFrameStats st{};
    const double t0 = nowMs();

    std::memset(backBuffer, 0, (size_t) width * height * 4);       // Demo.cpp:399
    std::vector<int32_t> tileRange((size_t) scr.totalTiles * 2, -1); // Demo.cpp:403

    std::vector<float> clipXY((size_t) count * 2), clipZ((size_t) count), ellipse((size_t) count * 4), conic((size_t) count * 4);

    double t = nowMs();
    if (shDegree != 0) // Demo.cpp:432
    {
        gsref_evaluate_spherical_harmonics(count, position4, sh, shDegree, shCount, camera->position, color4);
        st.msSphericalHarmonics = nowMs() - t;
    }
    t = nowMs();
    gsref_evaluate_splat_clip_data(count, position4, scaleAndRotation4, camera, width, height, clipXY.data(), clipZ.data(),
                                   ellipse.data(), conic.data());
    st.msClipData = nowMs() - t;

    // Count-then-emit instead of the reference's 8*N provision + doubling (Demo.cpp:325,357-366): same list.
    t = nowMs();
    int64_t size = gsref_build_tile_list(count, clipXY.data(), clipZ.data(), ellipse.data(), width, height, 0, 0, nullptr,
                                         nullptr, 0);
    std::vector<uint64_t> keys((size_t) size);
    std::vector<int32_t> values((size_t) size);
    gsref_build_tile_list(count, clipXY.data(), clipZ.data(), ellipse.data(), width, height, 0, 0, keys.data(), values.data(),
                          size);
    st.msBuildTileList = (nowMs() - t) * 0.5; // one of the two identical passes
    st.tileListSize = size;

    t = nowMs();
    gsref_sort_tile_list(size, keys.data(), values.data(), gsref_sort_end_bit(width, height));
    st.msSortTileList = nowMs() - t;

    t = nowMs();
    gsref_evaluate_tile_ranges(size, keys.data(), tileRange.data());
    st.msTileRanges = nowMs() - t;
    for (int i = 0; i != scr.totalTiles; ++i) st.nonEmptyTiles += (tileRange[2 * i + 1] - tileRange[2 * i]) != 0;

    t = nowMs();
    st.consumedPairs = gsref_rasterize_tiles(width, height, 0, 0, tileRange.data(), values.data(), clipXY.data(), conic.data(),
                                             color4, backBuffer);
    st.msRasterize = nowMs() - t;
    st.msTotal = nowMs() - t0;
    if (stats) *stats = st;
    return 0;
}

// ----------------------------------------------------------------------------------------------
// The same frame on several host threads (the optional all-core CPU baseline).  Orchestration only: every
// stage is the serial routine above on a slice of its input -- splats in contiguous chunks (multiples of the 256-splat SH group),
// the pair list concatenated in chunk order (= the serial emission order), a parallel stable merge sort (the stable order of a key
// sequence is unique), tile rows handed out one at a time -- so the image and the list are the serial frame's, bit for bit
// (tests/test_oracle_golden.py).
// ----------------------------------------------------------------------------------------------
namespace
{
template <typename F>
void forChunks(int threads, int64_t count, int64_t align, F&& body) // body(chunk, begin, end), chunks in order, one thread each
{
    threads = std::max(1, threads);
    int64_t per = (count + threads - 1) / threads;
    per = (per + align - 1) / align * align;
    std::vector<std::thread> pool;
    int chunk = 0;
    for (int64_t begin = 0; begin < count; begin += per, ++chunk)
        pool.emplace_back([=, &body] { body(chunk, begin, std::min(count, begin + per)); });
    for (auto& th : pool) th.join();
}
} // namespace

extern "C" int gsref_render_threads(int32_t count, const float* position4, const float* scaleAndRotation4, float* color4,
                                    const float* sh, int shDegree, int shCount, const GsRefCameraData* camera, int width,
                                    int height, uint8_t* backBuffer, GsRefFrameStats* stats, int threads)
{
    if (threads <= 1) return gsref_render(count, position4, scaleAndRotation4, color4, sh, shDegree, shCount, camera, width, height, backBuffer, stats);
    const ScreenInfo scr = makeScreen(width, height);
    GsRefFrameStats st{};
    const double t0 = nowMs();
    std::memset(backBuffer, 0, (size_t) width * height * 4);
    std::vector<int32_t> tileRange((size_t) scr.totalTiles * 2, -1);
    std::vector<float> clipXY((size_t) count * 2), clipZ((size_t) count), ellipse((size_t) count * 4), conic((size_t) count * 4);

    double t = nowMs();
    if (shDegree != 0)
    {
        // chunks start on a multiple of 256 splats: the blocked coefficient layout of a chunk then starts at sh + begin * shCount
        forChunks(threads, count, 256, [&](int, int64_t b, int64_t e) {
            gsref_evaluate_spherical_harmonics((int32_t) (e - b), position4 + 4 * b, sh + (size_t) b * shCount, shDegree, shCount, camera->position,
                                               color4 + 4 * b);
        });
        st.msSphericalHarmonics = nowMs() - t;
    }
    t = nowMs();
    forChunks(threads, count, 256, [&](int, int64_t b, int64_t e) {
        gsref_evaluate_splat_clip_data((int32_t) (e - b), position4 + 4 * b, scaleAndRotation4 + 4 * b, camera, width, height, clipXY.data() + 2 * b,
                                       clipZ.data() + b, ellipse.data() + 4 * b, conic.data() + 4 * b);
    });
    st.msClipData = nowMs() - t;

    t = nowMs();
    std::vector<int64_t> chunkPairs((size_t) threads + 1, 0), chunkBegin((size_t) threads + 1, 0);
    forChunks(threads, count, 256, [&](int c, int64_t b, int64_t e) {
        chunkPairs[(size_t) c] = gsref_build_tile_list((int32_t) (e - b), clipXY.data() + 2 * b, clipZ.data() + b, ellipse.data() + 4 * b, width, height,
                                                       0, 0, nullptr, nullptr, 0);
    });
    int64_t size = 0;
    for (int c = 0; c <= threads; ++c)
    {
        chunkBegin[(size_t) c] = size;
        size += c < threads ? chunkPairs[(size_t) c] : 0;
    }
    std::vector<uint64_t> keys((size_t) size);
    std::vector<int32_t> values((size_t) size);
    forChunks(threads, count, 256, [&](int c, int64_t b, int64_t e) {
        const int64_t at = chunkBegin[(size_t) c], n = chunkPairs[(size_t) c];
        gsref_build_tile_list((int32_t) (e - b), clipXY.data() + 2 * b, clipZ.data() + b, ellipse.data() + 4 * b, width, height, 0, 0, keys.data() + at,
                              values.data() + at, n);
        for (int64_t j = 0; j != n; ++j) values[(size_t) (at + j)] += (int32_t) b; // chunk-local splat index -> the scene's
    });
    st.msBuildTileList = (nowMs() - t) * 0.5;
    st.tileListSize = size;

    // parallel stable merge sort of the permutation (sorted runs per chunk, then pairwise merges of neighbours)
    t = nowMs();
    {
        const int endBit = gsref_sort_end_bit(width, height);
        const uint64_t mask = (endBit >= 64) ? ~0ull : ((1ull << endBit) - 1ull);
        std::vector<int64_t> order((size_t) size), scratch((size_t) size);
        std::iota(order.begin(), order.end(), 0);
        const auto less = [&](int64_t a, int64_t b) { return (keys[(size_t) a] & mask) < (keys[(size_t) b] & mask); };
        std::vector<int64_t> cut;
        int64_t per = (size + threads - 1) / std::max(threads, 1);
        per = std::max<int64_t>(per, 1);
        for (int64_t b = 0; b < size; b += per) cut.push_back(b);
        cut.push_back(size);
        {
            std::vector<std::thread> pool;
            for (size_t c = 0; c + 1 < cut.size(); ++c)
                pool.emplace_back([&, c] { std::stable_sort(order.begin() + cut[c], order.begin() + cut[c + 1], less); });
            for (auto& th : pool) th.join();
        }
        while (cut.size() > 2)
        {
            std::vector<int64_t> next;
            std::vector<std::thread> pool;
            for (size_t c = 0; c + 1 < cut.size(); c += 2)
            {
                next.push_back(cut[c]);
                if (c + 2 < cut.size())
                    pool.emplace_back([&, c] {
                        std::merge(order.begin() + cut[c], order.begin() + cut[c + 1], order.begin() + cut[c + 1], order.begin() + cut[c + 2],
                                   scratch.begin() + cut[c], less); // std::merge is stable: ties take the left run first
                        std::copy(scratch.begin() + cut[c], scratch.begin() + cut[c + 2], order.begin() + cut[c]);
                    });
            }
            next.push_back(size);
            for (auto& th : pool) th.join();
            cut.swap(next);
        }
        std::vector<uint64_t> k((size_t) size);
        std::vector<int32_t> v((size_t) size);
        forChunks(threads, size, 1, [&](int, int64_t b, int64_t e) {
            for (int64_t i = b; i != e; ++i)
            {
                k[(size_t) i] = keys[(size_t) order[(size_t) i]];
                v[(size_t) i] = values[(size_t) order[(size_t) i]];
            }
        });
        keys.swap(k);
        values.swap(v);
    }
    st.msSortTileList = nowMs() - t;

    t = nowMs();
    gsref_evaluate_tile_ranges(size, keys.data(), tileRange.data());
    st.msTileRanges = nowMs() - t;
    for (int i = 0; i != scr.totalTiles; ++i) st.nonEmptyTiles += (tileRange[2 * i + 1] - tileRange[2 * i]) != 0;

    t = nowMs();
    {
        std::atomic<int> nextRow{0};
        std::atomic<long long> consumed{0};
        std::vector<std::thread> pool;
        for (int w = 0; w != threads; ++w)
            pool.emplace_back([&] {
                for (int row = nextRow.fetch_add(1); row < scr.tilesY; row = nextRow.fetch_add(1))
                    consumed += gsref_rasterize_tiles(width, height, row, row + 1, tileRange.data(), values.data(), clipXY.data(), conic.data(), color4,
                                                      backBuffer);
            });
        for (auto& th : pool) th.join();
        st.consumedPairs = consumed.load();
    }
    st.msRasterize = nowMs() - t;
    st.msTotal = nowMs() - t0;
    if (stats) *stats = st;
    return 0;
}

// (Camera: gsref_camera_from_pose / gsref_camera_from_bounds are at the end of the file, on top of the CameraControls restatement.)

// ----------------------------------------------------------------------------------------------
// PLY ingest (PlyParser.cpp)
// ----------------------------------------------------------------------------------------------
namespace
{
struct PlyError { std::string msg; };

// PlyParser.cpp:15-109
void parseHeader(std::ifstream& inFile, std::vector<std::string>& properties, int& vertexCount)
{
    vertexCount = -1;
    std::string line;
    bool littleEndian = false;
    constexpr int maxLength = 256;
    int iteration = 0;
    while (std::getline(inFile, line))
    {
        std::stringstream ss(line);
        std::string word;
        if (!(ss >> word)) throw PlyError{"Unexpected empty line."};
        if (word == "ply") continue;
        else if (word == "format") littleEndian = (ss >> word) && word == "binary_little_endian";
        else if (word == "element")
        {
            if (!(ss >> word && word == "vertex")) throw PlyError{"Unexpected element type."};
            if (!(ss >> word)) throw PlyError{"Unexpected element count."};
            vertexCount = std::stoi(word);
        }
        else if (word == "property")
        {
            if (!(ss >> word && word == "float")) throw PlyError{"Unexpected property format, expected float."};
            if (!(ss >> word)) throw PlyError{"Unexpected property, missing name."};
            if (std::find(properties.begin(), properties.end(), word) != properties.end())
                throw PlyError{"Duplicated property \"" + word + "\"."};
            properties.push_back(word);
        }
        else if (word == "end_header")
        {
            if (!littleEndian || vertexCount == -1) throw PlyError{"Invalid header. Format or vertex count not found."};
            break;
        }
        ++iteration;
        if (iteration > maxLength) throw PlyError{"Invalid header. End not found."};
    }
}

int indexOf(const std::vector<std::string>& properties, const std::string& property)
{
    auto it = std::find(properties.begin(), properties.end(), property);
    if (it == properties.end()) throw PlyError{"Missing property \"" + property + "\"."};
    return (int) std::distance(properties.begin(), it);
}

// PlyParser.cpp:10-13: the unqualified exp() there binds to ::exp(double), so the whole expression is
// evaluated in double and rounded once on return.
float sigmoid(float x) { return (float) (1.0 / (1.0 + std::exp(-(double) x))); }
} // namespace

extern "C" int gsref_parse_ply(const char* path, GsRefPly* out, char* err, size_t errLen)
{
    std::memset(out, 0, sizeof(*out));
    try
    {
        // PlyParser.cpp:111-143
        std::ifstream inFile(path, std::ios_base::binary);
        std::vector<std::string> properties;
        int vertexCount = -1;
        parseHeader(inFile, properties, vertexCount);
        const size_t numProps = properties.size();
        std::vector<std::vector<float>> data(numProps, std::vector<float>((size_t) std::max(vertexCount, 0)));
        for (int i = 0; i != vertexCount; ++i)
            for (size_t j = 0; j != numProps; ++j)
            {
                float val = 0.0f;
                inFile.read(reinterpret_cast<char*>(&val), sizeof(float));
                data[j][i] = val;
            }

        // PlyParser.cpp:182-196
        static const char* required[14] = {"x", "y", "z", "rot_0", "rot_1", "rot_2", "rot_3", "scale_0", "scale_1", "scale_2",
                                           "f_dc_0", "f_dc_1", "f_dc_2", "opacity"};
        int offsets[14];
        for (int i = 0; i != 14; ++i) offsets[i] = indexOf(properties, required[i]);

        // PlyParser.cpp:204-243
        int extra = 0;
        std::vector<int> shOffsets;
        for (;;)
        {
            auto it = std::find(properties.begin(), properties.end(), "f_rest_" + std::to_string(extra));
            if (it == properties.end()) break;
            shOffsets.push_back((int) std::distance(properties.begin(), it));
            ++extra;
        }
        int expected = 0;
        int degree = 0;
        while (expected < extra)
        {
            expected += (2 * (degree + 1) + 1) * 3;
            ++degree;
        }
        if (expected != extra)
            throw PlyError{"Expected degree " + std::to_string(degree) + ", " + std::to_string(expected) +
                           " extra spherical harmonics. Found " + std::to_string(extra) + "."};
        const int shCount = extra + 3;

        out->vertexCount = vertexCount;
        out->shDegree = degree;
        out->shCount = shCount;
        const size_t n = (size_t) vertexCount;
        out->position4 = (float*) std::malloc(std::max<size_t>(1, n * 4 * sizeof(float)));
        out->scaleAndRotation4 = (float*) std::malloc(std::max<size_t>(1, n * 4 * sizeof(float)));
        out->color4 = (float*) std::malloc(std::max<size_t>(1, n * 4 * sizeof(float)));

        if (degree != 0)
        {
            // PlyParser.cpp:245-287: [f_dc rgb][f_rest channel-major] -> [basis][rgb]
            out->sh = (float*) std::malloc(std::max<size_t>(1, n * shCount * sizeof(float)));
            const int rgbShCount = extra / 3;
            for (size_t i = 0; i != n; ++i)
            {
                float* ptr = out->sh + (size_t) shCount * i;
                for (int j = 0; j != 3; ++j) ptr[j] = data[offsets[10 + j]][i];
                for (int j = 0; j != rgbShCount; ++j)
                {
                    ptr[3 + j * 3 + 0] = data[shOffsets[j]][i];
                    ptr[3 + j * 3 + 1] = data[shOffsets[rgbShCount + j]][i];
                    ptr[3 + j * 3 + 2] = data[shOffsets[rgbShCount * 2 + j]][i];
                }
            }
        }

        // PlyParser.cpp:289-336 (boundsMax starts at numeric_limits<float>::min(), a quirk kept on purpose)
        float bmin[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
        float bmax[3] = {std::numeric_limits<float>::min(), std::numeric_limits<float>::min(), std::numeric_limits<float>::min()};
        for (size_t i = 0; i != n; ++i)
        {
            float tr[3] = {data[offsets[0]][i], data[offsets[1]][i], data[offsets[2]][i]};
            float qw = data[offsets[3]][i], qx = data[offsets[4]][i], qy = data[offsets[5]][i], qz = data[offsets[6]][i];
            float sc[3] = {data[offsets[7]][i], data[offsets[8]][i], data[offsets[9]][i]};
            float dc[3] = {data[offsets[10]][i], data[offsets[11]][i], data[offsets[12]][i]};
            float opacity = data[offsets[13]][i];

            // glm::normalize(quat) (glm/ext/quaternion_geometric.inl): len = sqrt(dot(q,q)); if len <= 0 -> (1,0,0,0); q * (1/len)
            // dot(quat) = (w*w + x*x) + (y*y + z*z) (glm/detail/type_quat.inl:17-24)
            float len = std::sqrt((qw * qw + qx * qx) + (qy * qy + qz * qz));
            if (len <= 0.0f) { qw = 1.0f; qx = qy = qz = 0.0f; }
            else
            {
                float oneOverLen = 1.0f / len;
                qw *= oneOverLen; qx *= oneOverLen; qy *= oneOverLen; qz *= oneOverLen;
            }
            for (int k = 0; k != 3; ++k) sc[k] = std::exp(sc[k]);
            opacity = sigmoid(opacity);
            for (int k = 0; k != 3; ++k)
            {
                bmin[k] = gmin(bmin[k], tr[k]);
                bmax[k] = gmax(bmax[k], tr[k]);
            }
            const float SH_C0 = 0.28209479177387814f;
            float enc[4] = {(qx + 1.0f) * 0.5f, (qy + 1.0f) * 0.5f, (qz + 1.0f) * 0.5f, (qw + 1.0f) * 0.5f};
            uint32_t packed = gsref_encode_vec4(enc);
            float packedF;
            std::memcpy(&packedF, &packed, sizeof(float));
            float* p = out->position4 + 4 * i;
            p[0] = tr[0]; p[1] = tr[1]; p[2] = tr[2]; p[3] = opacity;
            float* s = out->scaleAndRotation4 + 4 * i;
            s[0] = sc[0]; s[1] = sc[1]; s[2] = sc[2]; s[3] = packedF;
            float* c = out->color4 + 4 * i;
            c[0] = dc[0] * SH_C0 + 0.5f; c[1] = dc[1] * SH_C0 + 0.5f; c[2] = dc[2] * SH_C0 + 0.5f; c[3] = opacity;
        }
        for (int k = 0; k != 3; ++k)
        {
            out->boundsMin[k] = bmin[k];
            out->boundsMax[k] = bmax[k];
        }
        return vertexCount;
    }
    catch (const PlyError& e)
    {
        if (err && errLen) std::snprintf(err, errLen, "%s", e.msg.c_str());
        gsref_free_ply(out);
        return -1;
    }
    catch (const std::exception& e) // std::stoi failures etc.
    {
        if (err && errLen) std::snprintf(err, errLen, "%s", e.what());
        gsref_free_ply(out);
        return -1;
    }
}

extern "C" void gsref_free_ply(GsRefPly* ply)
{
    std::free(ply->position4);
    std::free(ply->scaleAndRotation4);
    std::free(ply->color4);
    std::free(ply->sh);
    ply->position4 = ply->scaleAndRotation4 = ply->color4 = ply->sh = nullptr;
}

// ----------------------------------------------------------------------------------------------
// Test hooks: the glm restatements above, one by one, so that tests/test_oracle_golden.py can hold each of them bit for bit to
// what glm 1.0.1 itself returns (tests/golden/glm_ops.json, written by tests/golden/make_golden.py from oracle/glm_ops_driver.cpp
// compiled against the reference's vendored glm).  Matrices are column-major, m[col][row] at index 3*col + row (4*col + row).
// ----------------------------------------------------------------------------------------------
namespace
{
M3 loadM3(const float* p)
{
    M3 r;
    for (int c = 0; c != 3; ++c)
        for (int row = 0; row != 3; ++row) r.m[c][row] = p[c * 3 + row];
    return r;
}
void storeM3(const M3& a, float* p)
{
    for (int c = 0; c != 3; ++c)
        for (int row = 0; row != 3; ++row) p[c * 3 + row] = a.m[c][row];
}
} // namespace

extern "C" void gsref_glm_mat3_cast(const float xyzw[4], float out[9]) { storeM3(mat3Cast(xyzw[0], xyzw[1], xyzw[2], xyzw[3]), out); }
extern "C" void gsref_glm_mat3_mul(const float a[9], const float b[9], float out[9]) { storeM3(mul(loadM3(a), loadM3(b)), out); }
extern "C" void gsref_glm_mat3_transpose(const float a[9], float out[9]) { storeM3(transpose(loadM3(a)), out); }
extern "C" void gsref_glm_mat3_sandwich(const float a[9], const float b[9], float out[9])
{
    const M3 A = loadM3(a), B = loadM3(b);
    storeM3(mul(mul(A, transpose(B)), transpose(A)), out); // as the clip stage writes it (.cu:264)
}
extern "C" void gsref_glm_mat4_mul_vec4(const float m[16], const float v[4], float out[4])
{
    const V4 r = mul(loadM4(m), V4{v[0], v[1], v[2], v[3]});
    out[0] = r.x, out[1] = r.y, out[2] = r.z, out[3] = r.w;
}
extern "C" void gsref_glm_mat3_of_mat4(const float m[16], float out[9]) { storeM3(upper3(loadM4(m)), out); }
extern "C" void gsref_glm_normalize2(const float v[2], float out[2])
{
    const V2 r = normalize2(V2{v[0], v[1]});
    out[0] = r.x, out[1] = r.y;
}
extern "C" void gsref_glm_normalize3(const float v[3], float out[3])
{
    const V3 r = normalize3(V3{v[0], v[1], v[2]});
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
extern "C" float gsref_glm_atan2(float y, float x) { return std::atan2(y, x); }
extern "C" float gsref_glm_max(float a, float b) { return gmax(a, b); }
extern "C" float gsref_glm_min(float a, float b) { return gmin(a, b); }
extern "C" float gsref_glm_step(float edge, float x) { return gstep(edge, x); }
extern "C" float gsref_glm_mix(float x, float y, float a) { return gmix(x, y, a); }
extern "C" float gsref_glm_clamp01(float x) { return gclamp01(x); }
extern "C" int gsref_glm_clampi(int x, int lo, int hi) { return gclampi(x, lo, hi); }
extern "C" float gsref_glm_length2(const float v[2]) { return dot2(V2{v[0], v[1]}, V2{v[0], v[1]}); }
extern "C" float gsref_glm_dot2(const float a[2], const float b[2]) { return dot2(V2{a[0], a[1]}, V2{b[0], b[1]}); }
extern "C" float gsref_glm_sqrt(float x) { return std::sqrt(x); }
extern "C" float gsref_glm_cos(float x) { return std::cos(x); }
extern "C" float gsref_glm_sin(float x) { return std::sin(x); }
extern "C" float gsref_glm_floor(float x) { return std::floor(x); }
extern "C" float gsref_glm_ceil(float x) { return std::ceil(x); }

// ----------------------------------------------------------------------------------------------
// Interactive camera: CameraControls.cpp / CameraControls.h restated without the window.
//
// Pinning status: update() polls GLFW, which this image lacks, so the reference's update() cannot run here and a session as a
// whole is PARITY UNPINNED.  Everything update() is made of IS pinned, bit for bit (tests/golden/camera_controls.json):
//   * against the REFERENCE's own code, compiled where it lies (oracle/upstream_driver.cpp up_cc_*): getView / getProjection /
//     getViewProjection at arbitrary poses, the private getWorldSpaceRay, removeRoll, rayIntersectsPlane, getPlane, projectOnPlane,
//     the whole state setBounds leaves;
//   * against glm 1.0.1 itself (oracle/glm_ops_driver.cpp): angleAxis, quat * quat, quat * vec3, quat_cast, cross, dot,
//     mat4 * mat4, inverse(mat4).
// What remains hand-restated is the statement order of update() (CameraControls.cpp:148-253), cited line by line below.
// ----------------------------------------------------------------------------------------------
namespace
{
// tanf at RUN time, as the reference's binary calls it for its member m_FieldOfView (CameraControls.cpp:88-91,142; Demo.cpp:383): a
// compile-time fold of tan(30 deg) is correctly rounded and differs from glibc's tanf in the last bit.
float runtimeTan(float x)
{
    volatile float v = x;
    return std::tan(v);
}
// glm/ext/quaternion_trigonometric.inl:30-36; quaternions are V4 (x, y, z, w) here
V4 angleAxis(float angle, V3 v)
{
    float a = angle;
    float s = std::sin(a * 0.5f);
    return V4{v.x * s, v.y * s, v.z * s, std::cos(a * 0.5f)};
}
// glm/detail/type_quat.inl:298-308 (operator*= with p = left operand)
V4 quatMul(V4 p, V4 q)
{
    V4 r;
    r.w = p.w * q.w - p.x * q.x - p.y * q.y - p.z * q.z;
    r.x = p.w * q.x + p.x * q.w + p.y * q.z - p.z * q.y;
    r.y = p.w * q.y + p.y * q.w + p.z * q.x - p.x * q.z;
    r.z = p.w * q.z + p.z * q.w + p.x * q.y - p.y * q.x;
    return r;
}
// glm/detail/type_quat.inl:359-366: v + ((uv * q.w) + uuv) * 2
V3 quatRotate(V4 q, V3 v)
{
    V3 quatVector{q.x, q.y, q.z};
    V3 uv = cross(quatVector, v);
    V3 uuv = cross(quatVector, uv);
    return V3{v.x + ((uv.x * q.w) + uuv.x) * 2.0f, v.y + ((uv.y * q.w) + uuv.y) * 2.0f, v.z + ((uv.z * q.w) + uuv.z) * 2.0f};
}
// glm/detail/type_mat4x4.inl:630-648
M4 mul(const M4& a, const M4& b)
{
    M4 r;
    for (int c = 0; c != 4; ++c)
        for (int row = 0; row != 4; ++row)
        {
            float acc = a.m[0][row] * b.m[c][0];
            acc = acc + a.m[1][row] * b.m[c][1];
            acc = acc + a.m[2][row] * b.m[c][2];
            acc = acc + a.m[3][row] * b.m[c][3];
            r.m[c][row] = acc;
        }
    return r;
}
// glm/detail/func_matrix.inl, compute_inverse<4, 4, T, Q, Aligned>::call
M4 inverse(const M4& in)
{
    const float(*m)[4] = in.m;
    float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    V4 Fac0{Coef00, Coef00, Coef02, Coef03}, Fac1{Coef04, Coef04, Coef06, Coef07}, Fac2{Coef08, Coef08, Coef10, Coef11};
    V4 Fac3{Coef12, Coef12, Coef14, Coef15}, Fac4{Coef16, Coef16, Coef18, Coef19}, Fac5{Coef20, Coef20, Coef22, Coef23};
    V4 Vec0{m[1][0], m[0][0], m[0][0], m[0][0]}, Vec1{m[1][1], m[0][1], m[0][1], m[0][1]};
    V4 Vec2{m[1][2], m[0][2], m[0][2], m[0][2]}, Vec3{m[1][3], m[0][3], m[0][3], m[0][3]};
    auto mulv = [](V4 a, V4 b) { return V4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; };
    auto subv = [](V4 a, V4 b) { return V4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; };
    auto addv = [](V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; };
    V4 Inv0 = addv(subv(mulv(Vec1, Fac0), mulv(Vec2, Fac1)), mulv(Vec3, Fac2));
    V4 Inv1 = addv(subv(mulv(Vec0, Fac0), mulv(Vec2, Fac3)), mulv(Vec3, Fac4));
    V4 Inv2 = addv(subv(mulv(Vec0, Fac1), mulv(Vec1, Fac3)), mulv(Vec3, Fac5));
    V4 Inv3 = addv(subv(mulv(Vec0, Fac2), mulv(Vec1, Fac4)), mulv(Vec2, Fac5));
    V4 SignA{+1.0f, -1.0f, +1.0f, -1.0f}, SignB{-1.0f, +1.0f, -1.0f, +1.0f};
    V4 col[4] = {mulv(Inv0, SignA), mulv(Inv1, SignB), mulv(Inv2, SignA), mulv(Inv3, SignB)};
    V4 Row0{col[0].x, col[1].x, col[2].x, col[3].x};
    V4 Dot0 = mulv(V4{m[0][0], m[0][1], m[0][2], m[0][3]}, Row0);
    float Dot1 = (Dot0.x + Dot0.y) + (Dot0.z + Dot0.w);
    float OneOverDeterminant = 1.0f / Dot1;
    M4 r;
    for (int c = 0; c != 4; ++c)
    {
        r.m[c][0] = col[c].x * OneOverDeterminant;
        r.m[c][1] = col[c].y * OneOverDeterminant;
        r.m[c][2] = col[c].z * OneOverDeterminant;
        r.m[c][3] = col[c].w * OneOverDeterminant;
    }
    return r;
}
void storeM4(const M4& a, float* p)
{
    for (int c = 0; c != 4; ++c)
        for (int row = 0; row != 4; ++row) p[c * 4 + row] = a.m[c][row];
}

// CameraControls.cpp:3-13
bool rayIntersectsPlane(V3 rayOrigin, V3 rayDirection, V4 plane, float& t)
{
    float denom = dot3(rayDirection, V3{plane.x, plane.y, plane.z});
    if (denom == 0.0f) return false;
    t = -(dot3(rayOrigin, V3{plane.x, plane.y, plane.z}) + plane.w) / denom;
    return true;
}
// CameraControls.cpp:15-18
V4 getPlane(V3 normal, V3 point) { return V4{normal.x, normal.y, normal.z, -dot3(normal, point)}; }
// CameraControls.cpp:20-23 (vec - scalar: glm subtracts the scalar from every component)
V3 projectOnPlane(V3 vec, V3 planeNormal)
{
    float d = dot3(planeNormal, vec);
    return V3{vec.x - d, vec.y - d, vec.z - d};
}
const V3 k_Up{0.0f, 1.0f, 0.0f}, k_Right{1.0f, 0.0f, 0.0f}, k_Back{0.0f, 0.0f, 1.0f}; // CameraControls.cpp:25-27
// CameraControls.cpp:29-42
V4 removeRoll(V4 rotation)
{
    M3 rot = mat3Cast(rotation.x, rotation.y, rotation.z, rotation.w);
    V3 right{rot.m[0][0], rot.m[0][1], rot.m[0][2]};
    V3 up{rot.m[1][0], rot.m[1][1], rot.m[1][2]};
    right = normalize3(projectOnPlane(right, k_Up));
    up = normalize3(projectOnPlane(up, right));
    V3 forward = cross(right, up);
    M3 basis;
    basis.m[0][0] = right.x; basis.m[0][1] = right.y; basis.m[0][2] = right.z;
    basis.m[1][0] = up.x; basis.m[1][1] = up.y; basis.m[1][2] = up.z;
    basis.m[2][0] = forward.x; basis.m[2][1] = forward.y; basis.m[2][2] = forward.z;
    return quatCast(basis);
}

enum class CcMode { None, Pan, Orbit, Drag };                 // CameraControls.h:10-16
enum class CcMouseButton { None, Left, Middle, Right };       // CameraControls.h:18-24
} // namespace

struct GsRefCameraControls
{
    CcMode m_Mode = CcMode::None;
    bool m_MousePositionIsValid = false;
    CcMouseButton m_MouseButton = CcMouseButton::None;
    V2 m_PointerPosition{0.0f, 0.0f};
    V3 m_StartPanPosition{0.0f, 0.0f, 0.0f};
    bool m_StartPanPositionIsValid = false;
    V2 m_ScreenSize{1.0f, 1.0f};
    float m_Near = 0.1f, m_Far = 100.0f, m_FieldOfView = 60.0f * 0.01745329251994329576923690768489f; // CameraControls.h:35-37
    float m_Speed = 0.0f;                          // (these five are uninitialised in the reference until setBounds)
    V3 m_AnchorPosition{0.0f, 0.0f, 0.0f};
    V4 m_FloorPlane{0.0f, 1.0f, 0.0f, 0.0f};
    V4 m_Rotation{0.0f, 0.0f, 0.0f, 1.0f};
    V3 m_Position{0.0f, 0.0f, 0.0f};

    float getAspect() const { return m_ScreenSize.x / m_ScreenSize.y; }
    // CameraControls.cpp:79-86
    M4 getView() const
    {
        M3 r3 = mat3Cast(m_Rotation.x, m_Rotation.y, m_Rotation.z, m_Rotation.w);
        M4 rotation{}; // glm::mat4_cast = mat4(mat3_cast(q))
        for (int c = 0; c != 3; ++c)
            for (int row = 0; row != 3; ++row) rotation.m[c][row] = r3.m[c][row];
        rotation.m[3][3] = 1.0f;
        M4 identity{};
        for (int k = 0; k != 4; ++k) identity.m[k][k] = 1.0f;
        M4 translation = identity; // glm::translate, glm/ext/matrix_transform.inl:10-15
        for (int row = 0; row != 4; ++row)
            translation.m[3][row] = identity.m[0][row] * m_Position.x + identity.m[1][row] * m_Position.y + identity.m[2][row] * m_Position.z + identity.m[3][row];
        return inverse(mul(translation, rotation));
    }
    // CameraControls.cpp:88-91, glm/ext/matrix_clip_space.inl:249-262
    M4 getProjection() const
    {
        float tanHalfFovy = runtimeTan(m_FieldOfView / 2.0f);
        M4 r{};
        r.m[0][0] = 1.0f / (getAspect() * tanHalfFovy);
        r.m[1][1] = 1.0f / tanHalfFovy;
        r.m[2][2] = -(m_Far + m_Near) / (m_Far - m_Near);
        r.m[2][3] = -1.0f;
        r.m[3][2] = -(2.0f * m_Far * m_Near) / (m_Far - m_Near);
        return r;
    }
    M4 getViewProjection() const { return mul(getProjection(), getView()); } // CameraControls.cpp:93-96
    // CameraControls.cpp:98-114
    void getWorldSpaceRay(V2 pointerPosition, V3& rayOrigin, V3& rayDirection) const
    {
        M4 viewProjectionInverse = inverse(getViewProjection());
        V2 pointerPositionClip{(pointerPosition.x / m_ScreenSize.x) * 2.0f - 1.0f, (pointerPosition.y / m_ScreenSize.y) * 2.0f - 1.0f};
        pointerPositionClip.y *= -1.0f;
        V4 from = mul(viewProjectionInverse, V4{pointerPositionClip.x, pointerPositionClip.y, -1.0f, 1.0f});
        V4 to = mul(viewProjectionInverse, V4{pointerPositionClip.x, pointerPositionClip.y, 1.0f, 1.0f});
        from = V4{from.x / from.w, from.y / from.w, from.z / from.w, from.w / from.w};
        to = V4{to.x / to.w, to.y / to.w, to.z / to.w, to.w / to.w};
        rayOrigin = V3{from.x, from.y, from.z};
        V4 d{to.x - from.x, to.y - from.y, to.z - from.z, to.w - from.w};
        // glm::normalize(vec4) = v * inversesqrt(dot(v, v)), dot of vec4 = (x + y) + (z + w) (glm/detail/func_geometric.inl:58-65)
        float inv = 1.0f / std::sqrt((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w));
        rayDirection = V3{d.x * inv, d.y * inv, d.z * inv};
    }
    // CameraControls.cpp:132-146
    void setBounds(V3 min, V3 max)
    {
        V3 size{max.x - min.x, max.y - min.y, max.z - min.z};
        V3 center{min.x + size.x * 0.5f, min.y + size.y * 0.5f, min.z + size.z * 0.5f};
        float maxSize = gmax(size.x, gmax(size.y, size.z));
        m_Speed = maxSize * 0.02f;
        V3 offset = normalize3(V3{0.0f, maxSize * 0.5f, maxSize * 0.5f});
        float dist = std::sqrt(2.0f) * maxSize * 0.5f / runtimeTan(m_FieldOfView * 0.5f);
        m_Position = V3{center.x + offset.x * dist, center.y + offset.y * dist, center.z + offset.z * dist};
        // glm::quatLookAtRH(-offset, k_Up), glm/gtc/quaternion.inl:179-189
        V3 direction{-offset.x, -offset.y, -offset.z};
        V3 col2{-direction.x, -direction.y, -direction.z};
        V3 right = cross(k_Up, col2);
        float invLen = 1.0f / std::sqrt(gmax(0.00001f, dot3(right, right)));
        V3 col0{right.x * invLen, right.y * invLen, right.z * invLen};
        V3 col1 = cross(col2, col0);
        M3 basis;
        basis.m[0][0] = col0.x; basis.m[0][1] = col0.y; basis.m[0][2] = col0.z;
        basis.m[1][0] = col1.x; basis.m[1][1] = col1.y; basis.m[1][2] = col1.z;
        basis.m[2][0] = col2.x; basis.m[2][1] = col2.y; basis.m[2][2] = col2.z;
        m_Rotation = quatCast(basis);
        m_FloorPlane = getPlane(k_Up, center);
        m_AnchorPosition = center;
    }
    // CameraControls.cpp:148-253; the three glfw polls are the arguments
    void update(float dt, CcMouseButton mouseButton, double xpos, double ypos, V3 movement)
    {
        if (mouseButton != m_MouseButton) // :152-173
        {
            switch (mouseButton)
            {
            case CcMouseButton::None: m_Mode = CcMode::None; break;
            case CcMouseButton::Left: m_Mode = CcMode::Drag; break;
            case CcMouseButton::Middle: m_Mode = CcMode::Orbit; break;
            case CcMouseButton::Right:
                m_Mode = CcMode::Pan;
                m_StartPanPositionIsValid = false;
                break;
            }
            m_MouseButton = mouseButton;
        }
        bool pointerIsInside = xpos >= 0 || xpos < m_ScreenSize.x || ypos >= 0 || ypos < m_ScreenSize.y; // :178 (always true)
        V2 pointerPosition{(float) xpos, (float) ypos};
        V3 rayOrigin{0.0f, 0.0f, 0.0f}, rayDirection{0.0f, 0.0f, 0.0f};
        getWorldSpaceRay(pointerPosition, rayOrigin, rayDirection);
        V2 pointerDelta = m_MousePositionIsValid ? V2{pointerPosition.x - m_PointerPosition.x, pointerPosition.y - m_PointerPosition.y} : V2{0.0f, 0.0f};
        m_MousePositionIsValid = true;
        switch (m_Mode)
        {
        case CcMode::None: { // :195-202
            float t = 0.0f;
            if (rayIntersectsPlane(rayOrigin, rayDirection, m_FloorPlane, t))
                m_AnchorPosition = V3{rayOrigin.x + rayDirection.x * t, rayOrigin.y + rayDirection.y * t, rayOrigin.z + rayDirection.z * t};
        }
        break;
        case CcMode::Drag: { // :206-211
            V2 yawAndPitch{m_FieldOfView * getAspect() * pointerDelta.x / m_ScreenSize.x, m_FieldOfView * pointerDelta.y / m_ScreenSize.y};
            V4 yaw = angleAxis(yawAndPitch.x, k_Up);
            V4 pitch = angleAxis(yawAndPitch.y, k_Right);
            m_Rotation = removeRoll(quatMul(quatMul(m_Rotation, yaw), pitch));
        }
        break;
        case CcMode::Orbit: { // :215-227
            V2 yawAndPitch{m_FieldOfView * getAspect() * pointerDelta.x / m_ScreenSize.x, m_FieldOfView * pointerDelta.y / m_ScreenSize.y};
            V3 right = quatRotate(m_Rotation, k_Right);
            V4 pitchRot = angleAxis(-yawAndPitch.y, right);
            V4 yawRot = angleAxis(-yawAndPitch.x, k_Up);
            V4 deltaRotation = quatMul(yawRot, pitchRot);
            V4 rotation = removeRoll(quatMul(deltaRotation, m_Rotation));
            V3 anchorToCamera = quatRotate(deltaRotation, V3{m_Position.x - m_AnchorPosition.x, m_Position.y - m_AnchorPosition.y, m_Position.z - m_AnchorPosition.z});
            m_Rotation = rotation;
            m_Position = V3{m_AnchorPosition.x + anchorToCamera.x, m_AnchorPosition.y + anchorToCamera.y, m_AnchorPosition.z + anchorToCamera.z};
        }
        break;
        case CcMode::Pan: { // :230-248
            V4 plane = getPlane(quatRotate(m_Rotation, k_Back), m_AnchorPosition);
            float t = 0.0f;
            if (rayIntersectsPlane(rayOrigin, rayDirection, plane, t))
            {
                V3 hit{rayOrigin.x + rayDirection.x * t, rayOrigin.y + rayDirection.y * t, rayOrigin.z + rayDirection.z * t};
                if (m_StartPanPositionIsValid)
                {
                    V3 delta{hit.x - m_StartPanPosition.x, hit.y - m_StartPanPosition.y, hit.z - m_StartPanPosition.z};
                    m_Position = V3{m_Position.x - delta.x, m_Position.y - delta.y, m_Position.z - delta.z};
                }
                else if (pointerIsInside)
                {
                    m_StartPanPosition = hit;
                    m_StartPanPositionIsValid = true;
                }
            }
        }
        break;
        }
        V3 step = quatRotate(m_Rotation, V3{movement.x * dt, movement.y * dt, movement.z * dt}); // :251
        m_Position = V3{m_Position.x + step.x, m_Position.y + step.y, m_Position.z + step.z};
        m_PointerPosition = pointerPosition;
    }
};

extern "C" GsRefCameraControls* gsref_cc_create(float screenWidth, float screenHeight)
{
    GsRefCameraControls* c = new GsRefCameraControls();
    c->m_ScreenSize = V2{screenWidth, screenHeight};
    return c;
}
extern "C" void gsref_cc_destroy(GsRefCameraControls* c) { delete c; }
extern "C" void gsref_cc_set_bounds(GsRefCameraControls* c, const float bmin[3], const float bmax[3])
{
    c->setBounds(V3{bmin[0], bmin[1], bmin[2]}, V3{bmax[0], bmax[1], bmax[2]});
}
extern "C" void gsref_cc_set_pose(GsRefCameraControls* c, const float position[3], const float rotationXYZW[4])
{
    c->m_Position = V3{position[0], position[1], position[2]};
    c->m_Rotation = V4{rotationXYZW[0], rotationXYZW[1], rotationXYZW[2], rotationXYZW[3]};
}
// mouseButtons: bit 0 left, 1 middle, 2 right (getMouseButton's priority, CameraControls.cpp:116-130); keys: bit 0 W, 1 A, 2 S, 3 D, 4 Q, 5 E (:44-72)
extern "C" void gsref_cc_update(GsRefCameraControls* c, float dt, uint32_t mouseButtons, uint32_t keys, double cursorX, double cursorY)
{
    CcMouseButton b = CcMouseButton::None;
    if (mouseButtons & 1u) b = CcMouseButton::Left;
    else if (mouseButtons & 2u) b = CcMouseButton::Middle;
    else if (mouseButtons & 4u) b = CcMouseButton::Right;
    V3 result{0.0f, 0.0f, 0.0f};
    if (keys & 1u) result.z -= 1.0f;
    if (keys & 2u) result.x -= 1.0f;
    if (keys & 4u) result.z += 1.0f;
    if (keys & 8u) result.x += 1.0f;
    if (keys & 16u) result.y += 1.0f;
    if (keys & 32u) result.y -= 1.0f;
    c->update(dt, b, cursorX, cursorY, result);
}
// state[16]: position 3, rotation xyzw 4, anchor 3, floor plane 4, speed, mode
extern "C" void gsref_cc_state(const GsRefCameraControls* c, float state[16])
{
    const float s[16] = {c->m_Position.x, c->m_Position.y, c->m_Position.z, c->m_Rotation.x, c->m_Rotation.y, c->m_Rotation.z, c->m_Rotation.w,
                         c->m_AnchorPosition.x, c->m_AnchorPosition.y, c->m_AnchorPosition.z, c->m_FloorPlane.x, c->m_FloorPlane.y, c->m_FloorPlane.z,
                         c->m_FloorPlane.w, c->m_Speed, (float) (int) c->m_Mode};
    std::memcpy(state, s, sizeof(s));
}
extern "C" void gsref_cc_camera_data(const GsRefCameraControls* c, GsRefCameraData* out)
{
    // Demo.cpp:377-392
    std::memset(out, 0, sizeof(*out));
    storeM4(c->getView(), out->view);
    storeM4(c->getProjection(), out->projection);
    storeM4(c->getViewProjection(), out->viewProjection);
    float cotangentY = 1.0f / runtimeTan(c->m_FieldOfView * 0.5f);
    float cotangentX = cotangentY / c->getAspect();
    out->fovCotangent[0] = cotangentX;
    out->fovCotangent[1] = cotangentY;
    out->depthScaleBias[0] = -2.0f / (c->m_Far - c->m_Near);
    out->depthScaleBias[1] = -(c->m_Far + c->m_Near) / (c->m_Far - c->m_Near);
    out->position[0] = c->m_Position.x, out->position[1] = c->m_Position.y, out->position[2] = c->m_Position.z;
    out->aspect = c->getAspect();
}

// Default pose and CameraData of a given pose (CameraControls.cpp:79-96,132-146; CameraControls.h:35-37; Demo.cpp:377-392): the
// restatement above, so that view (through glm's general inverse), projection and the cotangents (run-time tanf) carry the bits the
// reference's binary produces (tests/golden/camera_default_pose.json: equal, not close).
extern "C" void gsref_camera_from_pose(const float position[3], const float rotationXYZW[4], int width, int height, GsRefCameraData* out)
{
    GsRefCameraControls c;
    c.m_ScreenSize = V2{(float) width, (float) height};
    c.m_Position = V3{position[0], position[1], position[2]};
    c.m_Rotation = V4{rotationXYZW[0], rotationXYZW[1], rotationXYZW[2], rotationXYZW[3]};
    gsref_cc_camera_data(&c, out);
}
extern "C" void gsref_camera_from_bounds(const float bmin[3], const float bmax[3], int width, int height, GsRefCameraData* out)
{
    GsRefCameraControls c;
    c.m_ScreenSize = V2{(float) width, (float) height};
    c.setBounds(V3{bmin[0], bmin[1], bmin[2]}, V3{bmax[0], bmax[1], bmax[2]});
    gsref_cc_camera_data(&c, out);
}

// ---- hooks with the signatures of oracle/upstream_driver.cpp's up_cc_* (held to the REFERENCE's results, tests/golden/camera_controls.json)
namespace
{
GsRefCameraControls posed(const float position[3], const float rotationXYZW[4], const float screen[2])
{
    GsRefCameraControls c;
    c.m_ScreenSize = V2{screen[0], screen[1]};
    c.m_Position = V3{position[0], position[1], position[2]};
    c.m_Rotation = V4{rotationXYZW[0], rotationXYZW[1], rotationXYZW[2], rotationXYZW[3]};
    return c;
}
} // namespace
extern "C" void gsref_cc_matrices(const float position[3], const float rotationXYZW[4], const float screen[2], float view[16],
                                  float projection[16], float viewProjection[16])
{
    GsRefCameraControls c = posed(position, rotationXYZW, screen);
    storeM4(c.getView(), view);
    storeM4(c.getProjection(), projection);
    storeM4(c.getViewProjection(), viewProjection);
}
extern "C" void gsref_cc_world_ray(const float position[3], const float rotationXYZW[4], const float screen[2], const float pointer[2],
                                   float origin[3], float direction[3])
{
    GsRefCameraControls c = posed(position, rotationXYZW, screen);
    V3 o, d;
    c.getWorldSpaceRay(V2{pointer[0], pointer[1]}, o, d);
    origin[0] = o.x, origin[1] = o.y, origin[2] = o.z;
    direction[0] = d.x, direction[1] = d.y, direction[2] = d.z;
}
extern "C" void gsref_cc_remove_roll(const float rotationXYZW[4], float out[4])
{
    V4 q = removeRoll(V4{rotationXYZW[0], rotationXYZW[1], rotationXYZW[2], rotationXYZW[3]});
    out[0] = q.x, out[1] = q.y, out[2] = q.z, out[3] = q.w;
}
extern "C" int gsref_cc_ray_plane(const float origin[3], const float direction[3], const float plane[4], float* t)
{
    return rayIntersectsPlane(V3{origin[0], origin[1], origin[2]}, V3{direction[0], direction[1], direction[2]}, V4{plane[0], plane[1], plane[2], plane[3]}, *t) ? 1 : 0;
}
extern "C" void gsref_cc_get_plane(const float normal[3], const float point[3], float out[4])
{
    V4 p = getPlane(V3{normal[0], normal[1], normal[2]}, V3{point[0], point[1], point[2]});
    out[0] = p.x, out[1] = p.y, out[2] = p.z, out[3] = p.w;
}
extern "C" void gsref_cc_project_on_plane(const float vec[3], const float normal[3], float out[3])
{
    V3 r = projectOnPlane(V3{vec[0], vec[1], vec[2]}, V3{normal[0], normal[1], normal[2]});
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
extern "C" void gsref_cc_set_bounds_state(const float bmin[3], const float bmax[3], const float screen[2], float position[3], float rotationXYZW[4],
                                          float anchor[3], float floorPlane[4], float* speed)
{
    GsRefCameraControls c;
    c.m_ScreenSize = V2{screen[0], screen[1]};
    c.setBounds(V3{bmin[0], bmin[1], bmin[2]}, V3{bmax[0], bmax[1], bmax[2]});
    position[0] = c.m_Position.x, position[1] = c.m_Position.y, position[2] = c.m_Position.z;
    rotationXYZW[0] = c.m_Rotation.x, rotationXYZW[1] = c.m_Rotation.y, rotationXYZW[2] = c.m_Rotation.z, rotationXYZW[3] = c.m_Rotation.w;
    anchor[0] = c.m_AnchorPosition.x, anchor[1] = c.m_AnchorPosition.y, anchor[2] = c.m_AnchorPosition.z;
    floorPlane[0] = c.m_FloorPlane.x, floorPlane[1] = c.m_FloorPlane.y, floorPlane[2] = c.m_FloorPlane.z, floorPlane[3] = c.m_FloorPlane.w;
    *speed = c.m_Speed;
}
// ---- the glm primitives update() adds to the pinned set (tests/golden/glm_ops.json)
extern "C" void gsref_glm_angle_axis(float angle, const float axis[3], float out[4])
{
    V4 q = angleAxis(angle, V3{axis[0], axis[1], axis[2]});
    out[0] = q.x, out[1] = q.y, out[2] = q.z, out[3] = q.w;
}
extern "C" void gsref_glm_quat_mul(const float p[4], const float q[4], float out[4])
{
    V4 r = quatMul(V4{p[0], p[1], p[2], p[3]}, V4{q[0], q[1], q[2], q[3]});
    out[0] = r.x, out[1] = r.y, out[2] = r.z, out[3] = r.w;
}
extern "C" void gsref_glm_quat_rotate(const float q[4], const float v[3], float out[3])
{
    V3 r = quatRotate(V4{q[0], q[1], q[2], q[3]}, V3{v[0], v[1], v[2]});
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
extern "C" void gsref_glm_quat_cast(const float m[9], float out[4])
{
    V4 q = quatCast(loadM3(m));
    out[0] = q.x, out[1] = q.y, out[2] = q.z, out[3] = q.w;
}
extern "C" void gsref_glm_cross(const float a[3], const float b[3], float out[3])
{
    V3 r = cross(V3{a[0], a[1], a[2]}, V3{b[0], b[1], b[2]});
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
extern "C" float gsref_glm_dot3(const float a[3], const float b[3]) { return dot3(V3{a[0], a[1], a[2]}, V3{b[0], b[1], b[2]}); }
extern "C" void gsref_glm_mat4_mul(const float a[16], const float b[16], float out[16]) { storeM4(mul(loadM4(a), loadM4(b)), out); }
extern "C" void gsref_glm_mat4_inverse(const float a[16], float out[16]) { storeM4(inverse(loadM4(a)), out); }

extern "C" int gsref_hardware_concurrency(void) { return (int) std::thread::hardware_concurrency(); }
