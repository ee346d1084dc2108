/*
 * glm_ops_driver.cpp -- C entry points around glm 1.0.1 ITSELF (the header-only library the reference vendors under
 * external/glm-1.0.1 and whose calls make up the arithmetic of evaluateSplatClipDataKernel, GaussianRender.cu:215-320,
 * the overlap helpers :351-436 and getKey :452).  The Makefile compiles this file against the headers where they lie
 * in /root/reference; nothing of glm or of the reference is copied here -- every function below only CALLS the glm
 * primitive named in its comment.
 *
 * TEST INFRASTRUCTURE ONLY, authoring container only: oracle/_ref/libglmops.so is used by tests/golden/make_golden.py
 * to write tests/golden/glm_ops.json (inputs + glm's outputs, as bit patterns), which pins the hand-written
 * restatements of these primitives in gsref.cpp (gsref_glm_* hooks) bit for bit.
 *
 * Built like the oracle (g++ -O2 -ffp-contract=off, glm's scalar code path: no GLM_FORCE_INTRINSICS), with the
 * reference's own glm configuration apart from GLM_FORCE_CUDA (GaussianRender.cuh:3-5,12-15).
 */
#define GLM_FORCE_SWIZZLE
#define GLM_ENABLE_EXPERIMENTAL
#include <glm/glm.hpp>
#include <glm/gtc/matrix_access.hpp>
#include <glm/gtc/matrix_transform.hpp>
#include <glm/gtx/norm.hpp>
#include <glm/gtx/quaternion.hpp>

namespace
{
glm::mat3 load3(const float* p)
{
    glm::mat3 m;
    for (int c = 0; c != 3; ++c)
        for (int r = 0; r != 3; ++r) m[c][r] = p[c * 3 + r];
    return m;
}
glm::mat4 load4(const float* p)
{
    glm::mat4 m;
    for (int c = 0; c != 4; ++c)
        for (int r = 0; r != 4; ++r) m[c][r] = p[c * 4 + r];
    return m;
}
void store3(const glm::mat3& m, float* p)
{
    for (int c = 0; c != 3; ++c)
        for (int r = 0; r != 3; ++r) p[c * 3 + r] = m[c][r];
}
} // namespace

extern "C"
{
// glm::quat::wxyz + glm::mat3_cast (.cu:221,223); the quaternion is used as it is, not normalised
void glmops_mat3_cast(const float xyzw[4], float out[9]) { store3(glm::mat3_cast(glm::quat::wxyz(xyzw[3], xyzw[0], xyzw[1], xyzw[2])), out); }
// mat3 * mat3 (.cu:223,224,262,264)
void glmops_mat3_mul(const float a[9], const float b[9], float out[9]) { store3(load3(a) * load3(b), out); }
// glm::transpose(mat3) (.cu:224,264)
void glmops_mat3_transpose(const float a[9], float out[9]) { store3(glm::transpose(load3(a)), out); }
// a * transpose(b) * transpose(a): the clip-space covariance product as one expression (.cu:264; left to right)
void glmops_mat3_sandwich(const float a[9], const float b[9], float out[9])
{
    const glm::mat3 A = load3(a), B = load3(b);
    store3(A * glm::transpose(B) * glm::transpose(A), out);
}
// mat4 * vec4 (.cu:227,265)
void glmops_mat4_mul_vec4(const float m[16], const float v[4], float out[4])
{
    const glm::vec4 r = load4(m) * glm::vec4(v[0], v[1], v[2], v[3]);
    out[0] = r.x, out[1] = r.y, out[2] = r.z, out[3] = r.w;
}
// (glm::mat3) mat4 (.cu:262)
void glmops_mat3_of_mat4(const float m[16], float out[9]) { store3((glm::mat3) load4(m), out); }
// glm::normalize(vec2) (.cu:290), (0,0) included
void glmops_normalize2(const float v[2], float out[2])
{
    const glm::vec2 r = glm::normalize(glm::vec2(v[0], v[1]));
    out[0] = r.x, out[1] = r.y;
}
// glm::normalize(vec3) (.cu:171)
void glmops_normalize3(const float v[3], float out[3])
{
    const glm::vec3 r = glm::normalize(glm::vec3(v[0], v[1], v[2]));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
// glm::atan(y, x) (.cu:295)
float glmops_atan2(float y, float x) { return glm::atan(y, x); }
// glm::max / glm::min on floats (.cu:285,288,305,429,430)
float glmops_max(float a, float b) { return glm::max(a, b); }
float glmops_min(float a, float b) { return glm::min(a, b); }
// glm::step(edge, x) (.cu:316,318), as the vec3 overload the kernel uses
float glmops_step(float edge, float x) { return glm::step(glm::vec3(edge), glm::vec3(x)).x; }
// glm::mix(vec3, vec3, float) (.cu:320)
float glmops_mix(float x, float y, float a) { return glm::mix(glm::vec3(x), glm::vec3(y), a).y; }
// single-argument glm::clamp (.cu:154,452)
float glmops_clamp01(float x) { return glm::clamp(x); }
// glm::clamp(int, int, int) (.cu:541-544)
int glmops_clampi(int x, int lo, int hi) { return glm::clamp(x, lo, hi); }
// glm::length2(vec2), glm::dot(vec2, vec2) (.cu:364,367,371,389)
float glmops_length2(const float v[2]) { return glm::length2(glm::vec2(v[0], v[1])); }
float glmops_dot2(const float a[2], const float b[2]) { return glm::dot(glm::vec2(a[0], a[1]), glm::vec2(b[0], b[1])); }
// glm::sqrt / cos / sin / floor / ceil on floats (.cu:285,299-301,326,541-544)
float glmops_sqrt(float x) { return glm::sqrt(x); }
float glmops_cos(float x) { return glm::cos(x); }
float glmops_sin(float x) { return glm::sin(x); }
float glmops_floor(float x) { return glm::floor(x); }
float glmops_ceil(float x) { return glm::ceil(x); }

// ---- the glm calls CameraControls::update() adds (CameraControls.cpp:29-42,79-114,195-251); quaternions travel as (x, y, z, w)
// glm::angleAxis(angle, axis) (CameraControls.cpp:208-209,218-219)
void glmops_angle_axis(float angle, const float axis[3], float out[4])
{
    const glm::quat q = glm::angleAxis(angle, glm::vec3(axis[0], axis[1], axis[2]));
    out[0] = q.x, out[1] = q.y, out[2] = q.z, out[3] = q.w;
}
// quat * quat (CameraControls.cpp:210,220-221)
void glmops_quat_mul(const float p[4], const float q[4], float out[4])
{
    const glm::quat r = glm::quat::wxyz(p[3], p[0], p[1], p[2]) * glm::quat::wxyz(q[3], q[0], q[1], q[2]);
    out[0] = r.x, out[1] = r.y, out[2] = r.z, out[3] = r.w;
}
// quat * vec3 (CameraControls.cpp:217,223,231,251)
void glmops_quat_rotate(const float q[4], const float v[3], float out[3])
{
    const glm::vec3 r = glm::quat::wxyz(q[3], q[0], q[1], q[2]) * glm::vec3(v[0], v[1], v[2]);
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
// glm::quat(mat3) (CameraControls.cpp:41)
void glmops_quat_cast(const float m[9], float out[4])
{
    const glm::quat q = glm::quat(load3(m));
    out[0] = q.x, out[1] = q.y, out[2] = q.z, out[3] = q.w;
}
// glm::cross, glm::dot on vec3 (CameraControls.cpp:5,11,17,22,39)
void glmops_cross(const float a[3], const float b[3], float out[3])
{
    const glm::vec3 r = glm::cross(glm::vec3(a[0], a[1], a[2]), glm::vec3(b[0], b[1], b[2]));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
float glmops_dot3(const float a[3], const float b[3]) { return glm::dot(glm::vec3(a[0], a[1], a[2]), glm::vec3(b[0], b[1], b[2])); }
// mat4 * mat4, glm::inverse(mat4) (CameraControls.cpp:85,95,102)
void glmops_mat4_mul(const float a[16], const float b[16], float out[16])
{
    const glm::mat4 r = load4(a) * load4(b);
    for (int c = 0; c != 4; ++c)
        for (int k = 0; k != 4; ++k) out[c * 4 + k] = r[c][k];
}
void glmops_mat4_inverse(const float a[16], float out[16])
{
    const glm::mat4 r = glm::inverse(load4(a));
    for (int c = 0; c != 4; ++c)
        for (int k = 0; k != 4; ++k) out[c * 4 + k] = r[c][k];
}
}
