// gs_camera.cpp -- the reference's interactive camera without a window.
//
// Replaces (file:line of /root/reference):
//   gs_camera_controls_create/destroy  <- CameraControls::CameraControls                 src/CameraControls.h:46-49
//   gs_camera_controls_set_bounds      <- CameraControls::setBounds                      src/CameraControls.cpp:132-146
//   gs_camera_controls_update          <- CameraControls::update (+ getMovement, getMouseButton, getWorldSpaceRay, removeRoll)
//                                                                                      src/CameraControls.cpp:29-77,98-130,148-253
//   gs_camera_controls_camera_data     <- getPosition/getView/getProjection/getViewProjection + the CameraData fill of the frame loop
//                                                                                      src/CameraControls.cpp:74-96, src/Demo.cpp:377-392
//
// The reference polls GLFW inside update() (glfwGetMouseButton, glfwGetCursorPos, glfwGetKey); here the caller hands the same
// three things in as a GsCameraInput, so a headless host can replay a recorded session (tools/gsplat_demo --events) and a
// windowed host can fill it from whatever toolkit it has.  Everything else -- the mode switch on a button CHANGE, the anchor that
// follows the pointer over the floor plane while no button is down, drag / orbit / pan, the roll removal, WASDQE movement along
// the camera's axes -- follows the reference statement by statement, in glm 1.0.1's operand order (each helper cites the glm file
// it restates), so that a session gives the same poses bit for bit.  Two of the reference's quirks are kept because they are
// behaviour: projectOnPlane subtracts the SCALAR dot product from every component (CameraControls.cpp:20-23), and "pointer is
// inside" is an OR of four comparisons and therefore always true (:189).
//
// Host code only (no HIP); built with -ffp-contract=off.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>

#include "gsplat_hip.h"

namespace
{
struct Vec2 { float x, y; };
struct Vec3 { float x, y, z; };
struct Vec4 { float x, y, z, w; };
struct Quat { float x, y, z, w; };
struct Mat3 { Vec3 c[3]; };       // columns
struct Mat4 { float m[4][4]; };   // m[column][row]

inline Vec3 add(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 sub(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 scale(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
// glm/detail/func_geometric.inl:48-55 (dot of vec3: tmp = a * b; tmp.x + tmp.y + tmp.z)
inline float dot(Vec3 a, Vec3 b)
{
    const float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
// glm/detail/func_geometric.inl:68-79
inline Vec3 cross(Vec3 x, Vec3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
// glm normalize(v) = v * inversesqrt(dot(v, v)), inversesqrt(x) = 1 / sqrt(x)
inline Vec3 normalize(Vec3 v) { return scale(v, 1.0f / std::sqrt(dot(v, v))); }

// glm/ext/quaternion_trigonometric.inl:30-36
inline Quat angleAxis(float angle, Vec3 axis)
{
    const float s = std::sin(angle * 0.5f);
    return {axis.x * s, axis.y * s, axis.z * s, std::cos(angle * 0.5f)};
}
// glm/detail/type_quat.inl:298-308
inline Quat mul(Quat p, Quat q)
{
    Quat r;
    r.w = p.w * q.w - p.x * q.x - p.y * q.y - p.z * q.z;
    r.x = p.w * q.x + p.x * q.w + p.y * q.z - p.z * q.y;
    r.y = p.w * q.y + p.y * q.w + p.z * q.x - p.x * q.z;
    r.z = p.w * q.z + p.z * q.w + p.x * q.y - p.y * q.x;
    return r;
}
// glm/detail/type_quat.inl:359-366
inline Vec3 rotate(Quat q, Vec3 v)
{
    const Vec3 qv{q.x, q.y, q.z};
    const Vec3 uv = cross(qv, v);
    const Vec3 uuv = cross(qv, uv);
    return add(v, scale(add(scale(uv, q.w), uuv), 2.0f));
}
// glm/gtc/quaternion.inl:47-72
inline Mat3 toMat3(Quat q)
{
    const float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z;
    const float qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z;
    const float qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
    Mat3 r;
    r.c[0] = {1.0f - 2.0f * (qyy + qzz), 2.0f * (qxy + qwz), 2.0f * (qxz - qwy)};
    r.c[1] = {2.0f * (qxy - qwz), 1.0f - 2.0f * (qxx + qzz), 2.0f * (qyz + qwx)};
    r.c[2] = {2.0f * (qxz + qwy), 2.0f * (qyz - qwx), 1.0f - 2.0f * (qxx + qyy)};
    return r;
}
// glm/gtc/quaternion.inl:81-123: the largest of the four squared components is taken by its square root, the rest by division
inline Quat toQuat(const Mat3& m)
{
    const float m00 = m.c[0].x, m01 = m.c[0].y, m02 = m.c[0].z;
    const float m10 = m.c[1].x, m11 = m.c[1].y, m12 = m.c[1].z;
    const float m20 = m.c[2].x, m21 = m.c[2].y, m22 = m.c[2].z;
    const float four[4] = {m00 + m11 + m22, m00 - m11 - m22, m11 - m00 - m22, m22 - m00 - m11}; // w, x, y, z
    int biggest = 0;
    for (int k = 1; k != 4; ++k)
        if (four[k] > four[biggest]) biggest = k;
    const float val = std::sqrt(four[biggest] + 1.0f) * 0.5f;
    const float mult = 0.25f / val;
    switch (biggest)
    {
    case 0: return {(m12 - m21) * mult, (m20 - m02) * mult, (m01 - m10) * mult, val};
    case 1: return {val, (m01 + m10) * mult, (m20 + m02) * mult, (m12 - m21) * mult};
    case 2: return {(m01 + m10) * mult, val, (m12 + m21) * mult, (m20 - m02) * mult};
    default: return {(m20 + m02) * mult, (m12 + m21) * mult, val, (m01 - m10) * mult};
    }
}

// glm/detail/type_mat4x4.inl:630-648: column c of a * b = a0 * b[c][0] + a1 * b[c][1] + a2 * b[c][2] + a3 * b[c][3], left to right
inline Mat4 mul(const Mat4& a, const Mat4& b)
{
    Mat4 r;
    for (int c = 0; c != 4; ++c)
        for (int row = 0; row != 4; ++row)
            r.m[c][row] = ((a.m[0][row] * b.m[c][0] + a.m[1][row] * b.m[c][1]) + a.m[2][row] * b.m[c][2]) + a.m[3][row] * b.m[c][3];
    return r;
}
// glm/detail/type_mat4x4.inl:536-572: (m0 * v0 + m1 * v1) + (m2 * v2 + m3 * v3)
inline Vec4 mul(const Mat4& m, Vec4 v)
{
    float r[4];
    for (int row = 0; row != 4; ++row) r[row] = (m.m[0][row] * v.x + m.m[1][row] * v.y) + (m.m[2][row] * v.z + m.m[3][row] * v.w);
    return {r[0], r[1], r[2], r[3]};
}
// glm/detail/func_matrix.inl compute_inverse<4, 4>: cofactors of 2x2 sub-determinants, signs by multiplication with +-1,
// determinant from the first row, one reciprocal
Mat4 inverse(const Mat4& in)
{
    const auto& m = in.m;
    const float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3], c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3], c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    const float c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3], c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3], c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    const float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2], c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2], c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    const float c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3], c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3], c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    const float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2], c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2], c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    const float c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1], c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1], c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
    const float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    const float v0[4] = {m[1][0], m[0][0], m[0][0], m[0][0]}, v1[4] = {m[1][1], m[0][1], m[0][1], m[0][1]};
    const float v2[4] = {m[1][2], m[0][2], m[0][2], m[0][2]}, v3[4] = {m[1][3], m[0][3], m[0][3], m[0][3]};
    const float signA[4] = {1.0f, -1.0f, 1.0f, -1.0f}, signB[4] = {-1.0f, 1.0f, -1.0f, 1.0f};
    Mat4 inv;
    for (int k = 0; k != 4; ++k)
    {
        inv.m[0][k] = ((v1[k] * f0[k] - v2[k] * f1[k]) + v3[k] * f2[k]) * signA[k];
        inv.m[1][k] = ((v0[k] * f0[k] - v2[k] * f3[k]) + v3[k] * f4[k]) * signB[k];
        inv.m[2][k] = ((v0[k] * f1[k] - v1[k] * f3[k]) + v3[k] * f5[k]) * signA[k];
        inv.m[3][k] = ((v0[k] * f2[k] - v1[k] * f4[k]) + v2[k] * f5[k]) * signB[k];
    }
    const float d0 = m[0][0] * inv.m[0][0], d1 = m[0][1] * inv.m[1][0], d2 = m[0][2] * inv.m[2][0], d3 = m[0][3] * inv.m[3][0];
    const float oneOverDeterminant = 1.0f / ((d0 + d1) + (d2 + d3));
    for (int c = 0; c != 4; ++c)
        for (int row = 0; row != 4; ++row) inv.m[c][row] *= oneOverDeterminant;
    return inv;
}

// tanf at RUN time, as the reference's binary calls it for its member m_FieldOfView: a compile-time fold of tan(30 deg) is
// correctly rounded and differs from glibc's tanf in the last bit
float runtimeTan(float x)
{
    volatile float v = x;
    return std::tan(v);
}

// CameraControls.cpp:3-23
inline Vec4 planeThrough(Vec3 normal, Vec3 point) { return {normal.x, normal.y, normal.z, -dot(normal, point)}; }
inline bool rayMeetsPlane(Vec3 origin, Vec3 direction, Vec4 plane, float& t)
{
    const Vec3 n{plane.x, plane.y, plane.z};
    const float denominator = dot(direction, n);
    if (denominator == 0.0f) return false;
    t = -(dot(origin, n) + plane.w) / denominator;
    return true;
}
inline Vec3 minusScalar(Vec3 v, float s) { return {v.x - s, v.y - s, v.z - s}; } // the reference's projectOnPlane, as it is

constexpr Vec3 kUp{0.0f, 1.0f, 0.0f}, kRight{1.0f, 0.0f, 0.0f}, kBack{0.0f, 0.0f, 1.0f};

// CameraControls.cpp:29-42
Quat withoutRoll(Quat rotation)
{
    const Mat3 basis = toMat3(rotation);
    Vec3 right = basis.c[0], up = basis.c[1];
    right = normalize(minusScalar(right, dot(kUp, right)));
    up = normalize(minusScalar(up, dot(right, up)));
    const Vec3 forward = cross(right, up);
    return toQuat(Mat3{{right, up, forward}});
}

enum Mode : int32_t { kNone = 0, kPan = 1, kOrbit = 2, kDrag = 3 };          // CameraControls.h:10-16
enum Button : int32_t { kNoButton = 0, kLeft = 1, kMiddle = 2, kRightButton = 3 }; // CameraControls.h:18-24
} // namespace

struct GsCameraControls
{
    int32_t mode = kNone;
    bool pointerKnown = false;
    int32_t button = kNoButton;
    Vec2 pointer{0.0f, 0.0f};
    Vec3 panStart{0.0f, 0.0f, 0.0f};
    bool panStartKnown = false;
    Vec2 screen{1.0f, 1.0f};
    float nearZ = 0.1f, farZ = 100.0f, fov = 60.0f * 0.01745329251994329576923690768489f; // CameraControls.h:35-37 (glm::radians)
    // (the reference leaves the next five uninitialised until setBounds; here a camera at the origin with the identity rotation)
    float speed = 0.0f;
    Vec3 anchor{0.0f, 0.0f, 0.0f};
    Vec4 floorPlane{0.0f, 1.0f, 0.0f, 0.0f};
    Quat rotation{0.0f, 0.0f, 0.0f, 1.0f};
    Vec3 position{0.0f, 0.0f, 0.0f};

    float aspect() const { return screen.x / screen.y; } // CameraControls.h:50-53
    // CameraControls.cpp:79-86: inverse(translate(1, position) * mat4_cast(rotation)) -- through the general 4x4 inverse, as there
    Mat4 view() const
    {
        const Mat3 r = toMat3(rotation);
        Mat4 rot{}, trans{};
        for (int c = 0; c != 3; ++c)
        {
            rot.m[c][0] = r.c[c].x;
            rot.m[c][1] = r.c[c].y;
            rot.m[c][2] = r.c[c].z;
        }
        rot.m[3][3] = 1.0f;
        for (int k = 0; k != 4; ++k) trans.m[k][k] = 1.0f;
        // glm/ext/matrix_transform.inl:10-15: column 3 = m0 * v0 + m1 * v1 + m2 * v2 + m3
        const float v[3] = {position.x, position.y, position.z};
        for (int row = 0; row != 4; ++row)
            trans.m[3][row] = ((trans.m[0][row] * v[0] + trans.m[1][row] * v[1]) + trans.m[2][row] * v[2]) + (row == 3 ? 1.0f : 0.0f);
        return inverse(mul(trans, rot));
    }
    // glm::perspective = perspectiveRH_NO, glm/ext/matrix_clip_space.inl:249-262
    Mat4 projection() const
    {
        const float tanHalfFovy = runtimeTan(fov / 2.0f);
        Mat4 p{};
        p.m[0][0] = 1.0f / (aspect() * tanHalfFovy);
        p.m[1][1] = 1.0f / tanHalfFovy;
        p.m[2][2] = -(farZ + nearZ) / (farZ - nearZ);
        p.m[2][3] = -1.0f;
        p.m[3][2] = -(2.0f * farZ * nearZ) / (farZ - nearZ);
        return p;
    }
    Mat4 viewProjection() const { return mul(projection(), view()); } // CameraControls.cpp:93-96

    // CameraControls.cpp:98-114
    void worldSpaceRay(Vec2 pointerPosition, Vec3& origin, Vec3& direction) const
    {
        const Mat4 inv = inverse(viewProjection());
        Vec2 clip{(pointerPosition.x / screen.x) * 2.0f - 1.0f, (pointerPosition.y / screen.y) * 2.0f - 1.0f};
        clip.y *= -1.0f; // mouse coordinates start at the top
        Vec4 from = mul(inv, Vec4{clip.x, clip.y, -1.0f, 1.0f});
        Vec4 to = mul(inv, Vec4{clip.x, clip.y, 1.0f, 1.0f});
        from = {from.x / from.w, from.y / from.w, from.z / from.w, from.w / from.w};
        to = {to.x / to.w, to.y / to.w, to.z / to.w, to.w / to.w};
        origin = {from.x, from.y, from.z};
        // normalize() of the four-component difference (glm/detail/func_geometric.inl:58-65: (x + y) + (z + w)), first three kept
        const Vec4 d{to.x - from.x, to.y - from.y, to.z - from.z, to.w - from.w};
        const float inv4 = 1.0f / std::sqrt((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w));
        direction = {d.x * inv4, d.y * inv4, d.z * inv4};
    }
};

extern "C" GsCameraControls* gs_camera_controls_create(float screenWidth, float screenHeight)
{
    GsCameraControls* c = new (std::nothrow) GsCameraControls();
    if (c) c->screen = {screenWidth, screenHeight};
    return c;
}

extern "C" void gs_camera_controls_destroy(GsCameraControls* c) { delete c; }

extern "C" void gs_camera_controls_set_bounds(GsCameraControls* c, const float bmin[3], const float bmax[3])
{
    // CameraControls.cpp:132-146
    const Vec3 lo{bmin[0], bmin[1], bmin[2]}, hi{bmax[0], bmax[1], bmax[2]};
    const Vec3 size = sub(hi, lo);
    const Vec3 center = add(lo, scale(size, 0.5f));
    const float inner = (size.y < size.z) ? size.z : size.y; // glm::max(x, y) = (x < y) ? y : x
    const float maxSize = (size.x < inner) ? inner : size.x;
    c->speed = maxSize * 0.02f;
    const Vec3 offset = normalize(Vec3{0.0f, maxSize * 0.5f, maxSize * 0.5f});
    const float dist = std::sqrt(2.0f) * maxSize * 0.5f / runtimeTan(c->fov * 0.5f);
    c->position = add(center, scale(offset, dist));
    // glm::quatLookAt(-offset, up) = quatLookAtRH, glm/gtc/quaternion.inl:179-189
    const Vec3 direction{-offset.x, -offset.y, -offset.z};
    const Vec3 col2{-direction.x, -direction.y, -direction.z};
    const Vec3 right = cross(kUp, col2);
    const float rr = dot(right, right);
    const Vec3 col0 = scale(right, 1.0f / std::sqrt((0.00001f < rr) ? rr : 0.00001f));
    const Vec3 col1 = cross(col2, col0);
    c->rotation = toQuat(Mat3{{col0, col1, col2}});
    c->floorPlane = planeThrough(kUp, center);
    c->anchor = center;
}

extern "C" void gs_camera_controls_update(GsCameraControls* c, const GsCameraInput* in, float dt)
{
    // CameraControls.cpp:116-130: left wins over middle wins over right
    int32_t button = kNoButton;
    if (in->mouseButtons & GS_MOUSE_LEFT) button = kLeft;
    else if (in->mouseButtons & GS_MOUSE_MIDDLE) button = kMiddle;
    else if (in->mouseButtons & GS_MOUSE_RIGHT) button = kRightButton;
    // :150-173 the mode follows a CHANGE of the button
    if (button != c->button)
    {
        switch (button)
        {
        case kNoButton: c->mode = kNone; break;
        case kLeft: c->mode = kDrag; break;
        case kMiddle: c->mode = kOrbit; break;
        default:
            c->mode = kPan;
            c->panStartKnown = false;
            break;
        }
        c->button = button;
    }

    // :175-189
    const double xpos = in->cursorX, ypos = in->cursorY;
    const bool pointerIsInside = xpos >= 0 || xpos < c->screen.x || ypos >= 0 || ypos < c->screen.y;
    const Vec2 pointer{(float) xpos, (float) ypos};
    Vec3 rayOrigin{0.0f, 0.0f, 0.0f}, rayDirection{0.0f, 0.0f, 0.0f};
    c->worldSpaceRay(pointer, rayOrigin, rayDirection);
    const Vec2 delta = c->pointerKnown ? Vec2{pointer.x - c->pointer.x, pointer.y - c->pointer.y} : Vec2{0.0f, 0.0f};
    c->pointerKnown = true;

    // yaw and pitch of the two rotating modes (:207,216): (fov * aspect, fov) * delta / screen, component by component
    const Vec2 yawAndPitch{((c->fov * c->aspect()) * delta.x) / c->screen.x, (c->fov * delta.y) / c->screen.y};
    switch (c->mode)
    {
    case kNone: { // :195-202 the anchor follows the pointer over the floor plane
        float t = 0.0f;
        if (rayMeetsPlane(rayOrigin, rayDirection, c->floorPlane, t)) c->anchor = add(rayOrigin, scale(rayDirection, t));
        break;
    }
    case kDrag: { // :206-211 look around
        const Quat yaw = angleAxis(yawAndPitch.x, kUp);
        const Quat pitch = angleAxis(yawAndPitch.y, kRight);
        c->rotation = withoutRoll(mul(mul(c->rotation, yaw), pitch));
        break;
    }
    case kOrbit: { // :215-227 turn about the anchor
        const Vec3 right = rotate(c->rotation, kRight);
        const Quat pitchRotation = angleAxis(-yawAndPitch.y, right);
        const Quat yawRotation = angleAxis(-yawAndPitch.x, kUp);
        const Quat deltaRotation = mul(yawRotation, pitchRotation);
        const Quat rotation = withoutRoll(mul(deltaRotation, c->rotation));
        const Vec3 anchorToCamera = rotate(deltaRotation, sub(c->position, c->anchor));
        c->rotation = rotation;
        c->position = add(c->anchor, anchorToCamera);
        break;
    }
    default: { // :230-248 pan in the plane through the anchor that faces the camera
        const Vec4 plane = planeThrough(rotate(c->rotation, kBack), c->anchor);
        float t = 0.0f;
        if (rayMeetsPlane(rayOrigin, rayDirection, plane, t))
        {
            const Vec3 hit = add(rayOrigin, scale(rayDirection, t));
            if (c->panStartKnown) c->position = sub(c->position, sub(hit, c->panStart));
            else if (pointerIsInside)
            {
                c->panStart = hit;
                c->panStartKnown = true;
            }
        }
        break;
    }
    }

    // :44-72, :251 W A S D Q E along the camera's axes
    Vec3 movement{0.0f, 0.0f, 0.0f};
    if (in->keys & GS_KEY_W) movement.z -= 1.0f;
    if (in->keys & GS_KEY_A) movement.x -= 1.0f;
    if (in->keys & GS_KEY_S) movement.z += 1.0f;
    if (in->keys & GS_KEY_D) movement.x += 1.0f;
    if (in->keys & GS_KEY_Q) movement.y += 1.0f;
    if (in->keys & GS_KEY_E) movement.y -= 1.0f;
    c->position = add(c->position, rotate(c->rotation, scale(movement, dt)));
    c->pointer = pointer;
}

extern "C" void gs_camera_controls_get_state(const GsCameraControls* c, GsCameraControlsState* out)
{
    std::memset(out, 0, sizeof(*out));
    const float p[3] = {c->position.x, c->position.y, c->position.z}, a[3] = {c->anchor.x, c->anchor.y, c->anchor.z};
    const float q[4] = {c->rotation.x, c->rotation.y, c->rotation.z, c->rotation.w};
    std::memcpy(out->position, p, sizeof(p));
    std::memcpy(out->rotationXYZW, q, sizeof(q));
    std::memcpy(out->anchorPosition, a, sizeof(a));
    out->speed = c->speed;
    out->mode = c->mode;
}

extern "C" void gs_camera_controls_set_pose(GsCameraControls* c, const float position[3], const float rotationXYZW[4])
{
    c->position = {position[0], position[1], position[2]};
    c->rotation = {rotationXYZW[0], rotationXYZW[1], rotationXYZW[2], rotationXYZW[3]};
}

extern "C" void gs_camera_controls_camera_data(const GsCameraControls* c, GsCameraData* out)
{
    // Demo.cpp:377-392
    std::memset(out, 0, sizeof(*out));
    const Mat4 view = c->view(), projection = c->projection(), viewProjection = mul(projection, view);
    std::memcpy(out->view, view.m, sizeof(view.m));
    std::memcpy(out->projection, projection.m, sizeof(projection.m));
    std::memcpy(out->viewProjection, viewProjection.m, sizeof(viewProjection.m));
    const float cotangentY = 1.0f / runtimeTan(c->fov * 0.5f);
    out->fovCotangent[0] = cotangentY / c->aspect();
    out->fovCotangent[1] = cotangentY;
    out->depthScaleBias[0] = -2.0f / (c->farZ - c->nearZ);
    out->depthScaleBias[1] = -(c->farZ + c->nearZ) / (c->farZ - c->nearZ);
    out->position[0] = c->position.x;
    out->position[1] = c->position.y;
    out->position[2] = c->position.z;
    out->aspect = c->aspect();
}

// Default pose / CameraData of a given pose without a controls object (CameraControls.cpp:79-96,132-146 + Demo.cpp:377-392): the
// code above, so the matrices and cotangents carry the bits the reference's binary produces.
extern "C" void gs_camera_from_pose(const float position[3], const float rotationXYZW[4], int width, int height, GsCameraData* out)
{
    GsCameraControls c;
    c.screen = {(float) width, (float) height};
    gs_camera_controls_set_pose(&c, position, rotationXYZW);
    gs_camera_controls_camera_data(&c, out);
}

extern "C" void gs_camera_from_bounds(const float boundsMin[3], const float boundsMax[3], int width, int height, GsCameraData* out)
{
    GsCameraControls c;
    c.screen = {(float) width, (float) height};
    gs_camera_controls_set_bounds(&c, boundsMin, boundsMax);
    gs_camera_controls_camera_data(&c, out);
}
