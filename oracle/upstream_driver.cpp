/*
 * upstream_driver.cpp -- C entry points around the REFERENCE's own PlyParser.cpp and
 * CameraControls.cpp, which the Makefile compiles from /root/reference where they lie.
 *
 * TEST INFRASTRUCTURE ONLY, and only in the authoring container: the resulting
 * oracle/_ref/libgsupstream.so is used by tests/golden/make_golden.py to pin the CPU restatement
 * (gsref.cpp) and to generate the committed fixtures.  Nothing of the reference is copied here:
 * this file only calls its public functions (PlyParser.h:11-21, CameraControls.h:50-75).
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "CameraControls.h"
#include "PlyParser.h"

extern "C" uint32_t up_encode_vec4(const float v[4])
{
    return encodeVec4(glm::vec4(v[0], v[1], v[2], v[3]));
}

struct UpPly
{
    int32_t vertexCount, shDegree, shCount;
    float boundsMin[3], boundsMax[3];
    float *position4, *scaleAndRotation4, *color4, *sh;
};

static float* dupBytes(const void* src, size_t bytes)
{
    float* p = (float*) std::malloc(bytes ? bytes : 1);
    if (bytes) std::memcpy(p, src, bytes);
    return p;
}

extern "C" int up_parse_ply(const char* path, UpPly* out, char* err, size_t errLen)
{
    std::memset(out, 0, sizeof(*out));
    try
    {
        std::vector<float4> position, scaleAndRotation, color;
        std::vector<float> sh;
        int degree = 0, shCount = 0;
        glm::vec3 bmin, bmax;
        int n = parsePly(path, position, scaleAndRotation, color, sh, degree, shCount, bmin, bmax);
        out->vertexCount = n;
        out->shDegree = degree;
        out->shCount = shCount;
        for (int k = 0; k != 3; ++k)
        {
            out->boundsMin[k] = bmin[k];
            out->boundsMax[k] = bmax[k];
        }
        out->position4 = dupBytes(position.data(), position.size() * sizeof(float4));
        out->scaleAndRotation4 = dupBytes(scaleAndRotation.data(), scaleAndRotation.size() * sizeof(float4));
        out->color4 = dupBytes(color.data(), color.size() * sizeof(float4));
        out->sh = sh.empty() ? nullptr : dupBytes(sh.data(), sh.size() * sizeof(float));
        return n;
    }
    catch (const std::exception& e)
    {
        if (err && errLen) std::snprintf(err, errLen, "%s", e.what());
        return -1;
    }
}

extern "C" void up_free_ply(UpPly* p)
{
    std::free(p->position4);
    std::free(p->scaleAndRotation4);
    std::free(p->color4);
    std::free(p->sh);
    std::memset(p, 0, sizeof(*p));
}

/* Default pose: CameraControls(window=nullptr, screenSize).setBounds(min,max) then the getters the
 * frame loop uses (Demo.cpp:377-392). */
extern "C" void up_camera_default(const float bmin[3], const float bmax[3], float screenW, float screenH, float position[3],
                                  float view[16], float projection[16], float viewProjection[16], float params[4])
{
    CameraControls cam(nullptr, glm::vec2(screenW, screenH));
    cam.setBounds(glm::vec3(bmin[0], bmin[1], bmin[2]), glm::vec3(bmax[0], bmax[1], bmax[2]));
    glm::vec3 p = cam.getPosition();
    glm::mat4 v = cam.getView(), pr = cam.getProjection(), vp = cam.getViewProjection();
    for (int k = 0; k != 3; ++k) position[k] = p[k];
    std::memcpy(view, &v[0][0], 16 * sizeof(float));
    std::memcpy(projection, &pr[0][0], 16 * sizeof(float));
    std::memcpy(viewProjection, &vp[0][0], 16 * sizeof(float));
    params[0] = cam.getAspect();
    params[1] = cam.getFieldOfView();
    params[2] = cam.getNear();
    params[3] = cam.getFar();
}

/* ---- the building blocks of CameraControls::update(), called where they lie --------------------------------
 * update() itself polls GLFW (glfwGetMouseButton / glfwGetCursorPos / glfwGetKey), which this image does not have, so it is
 * never called; what it is made of can be: the four free functions of CameraControls.cpp (external linkage: removeRoll,
 * rayIntersectsPlane, getPlane, projectOnPlane), the private getWorldSpaceRay and the view / projection getters of a camera at
 * an ARBITRARY pose.  The pose members are private and only setBounds writes them, so this file is compiled with
 * -fno-access-control (oracle/Makefile): the class, its layout and its code are the reference's, untouched. */
bool rayIntersectsPlane(const glm::vec3& rayOrigin, const glm::vec3& rayDirection, const glm::vec4& plane, float& t); /* CameraControls.cpp:3 */
glm::vec4 getPlane(const glm::vec3& normal, const glm::vec3& point);                                                   /* :15 */
glm::vec3 projectOnPlane(const glm::vec3& vec, const glm::vec3& planeNormal);                                          /* :20 */
glm::quat removeRoll(glm::quat rotation);                                                                              /* :29 */

static CameraControls posedCamera(const float position[3], const float rotationXYZW[4], const float screen[2])
{
    CameraControls cam(nullptr, glm::vec2(screen[0], screen[1]));
    cam.m_Position = glm::vec3(position[0], position[1], position[2]);
    cam.m_Rotation = glm::quat::wxyz(rotationXYZW[3], rotationXYZW[0], rotationXYZW[1], rotationXYZW[2]);
    return cam;
}

extern "C" void up_cc_matrices(const float position[3], const float rotationXYZW[4], const float screen[2], float view[16],
                               float projection[16], float viewProjection[16])
{
    const CameraControls cam = posedCamera(position, rotationXYZW, screen);
    const glm::mat4 v = cam.getView(), p = cam.getProjection(), vp = cam.getViewProjection();
    std::memcpy(view, &v[0][0], 16 * sizeof(float));
    std::memcpy(projection, &p[0][0], 16 * sizeof(float));
    std::memcpy(viewProjection, &vp[0][0], 16 * sizeof(float));
}

extern "C" void up_cc_world_ray(const float position[3], const float rotationXYZW[4], const float screen[2], const float pointer[2],
                                float origin[3], float direction[3])
{
    const CameraControls cam = posedCamera(position, rotationXYZW, screen);
    glm::vec3 o, d;
    cam.getWorldSpaceRay(glm::vec2(pointer[0], pointer[1]), o, d);
    for (int k = 0; k != 3; ++k) origin[k] = o[k], direction[k] = d[k];
}

extern "C" void up_cc_remove_roll(const float rotationXYZW[4], float out[4])
{
    const glm::quat q = removeRoll(glm::quat::wxyz(rotationXYZW[3], rotationXYZW[0], rotationXYZW[1], rotationXYZW[2]));
    out[0] = q.x, out[1] = q.y, out[2] = q.z, out[3] = q.w;
}

extern "C" int up_cc_ray_plane(const float origin[3], const float direction[3], const float plane[4], float* t)
{
    return rayIntersectsPlane(glm::vec3(origin[0], origin[1], origin[2]), glm::vec3(direction[0], direction[1], direction[2]),
                              glm::vec4(plane[0], plane[1], plane[2], plane[3]), *t)
               ? 1
               : 0;
}

extern "C" void up_cc_get_plane(const float normal[3], const float point[3], float out[4])
{
    const glm::vec4 p = getPlane(glm::vec3(normal[0], normal[1], normal[2]), glm::vec3(point[0], point[1], point[2]));
    out[0] = p.x, out[1] = p.y, out[2] = p.z, out[3] = p.w;
}

extern "C" void up_cc_project_on_plane(const float vec[3], const float normal[3], float out[3])
{
    const glm::vec3 r = projectOnPlane(glm::vec3(vec[0], vec[1], vec[2]), glm::vec3(normal[0], normal[1], normal[2]));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}

/* setBounds' whole state (position, rotation, anchor, floor plane, speed): CameraControls.cpp:132-146 */
extern "C" void up_cc_set_bounds(const float bmin[3], const float bmax[3], const float screen[2], float position[3], float rotationXYZW[4],
                                 float anchor[3], float floorPlane[4], float* speed)
{
    CameraControls cam(nullptr, glm::vec2(screen[0], screen[1]));
    cam.setBounds(glm::vec3(bmin[0], bmin[1], bmin[2]), glm::vec3(bmax[0], bmax[1], bmax[2]));
    for (int k = 0; k != 3; ++k) position[k] = cam.m_Position[k], anchor[k] = cam.m_AnchorPosition[k];
    rotationXYZW[0] = cam.m_Rotation.x, rotationXYZW[1] = cam.m_Rotation.y, rotationXYZW[2] = cam.m_Rotation.z, rotationXYZW[3] = cam.m_Rotation.w;
    for (int k = 0; k != 4; ++k) floorPlane[k] = cam.m_FloorPlane[k];
    *speed = cam.m_Speed;
}
