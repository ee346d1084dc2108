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
