/*
 * gsref.h -- C interface of the CPU oracle ("libgsref").
 *
 * TEST INFRASTRUCTURE ONLY.  This is a serial CPU restatement of the reference's
 * per-frame hot path (etienne-p/CUDAGaussianRenderer, src/GaussianRender.cu) plus its
 * PLY ingest and default camera.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product (libgsplat_hip.so and the
 * cudagaussianrenderer_amd package) never links, imports or calls anything in here.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - PLY ingest / encodeVec4 / default camera pose: PINNED against the reference's own
 *     PlyParser.cpp and CameraControls.cpp compiled from /root/reference (oracle/_ref,
 *     tests/golden/ply_*.json, camera_default_pose.json).
 *   - Interactive camera (CameraControls::update): its building blocks (view / projection at arbitrary poses,
 *     getWorldSpaceRay, removeRoll, rayIntersectsPlane, getPlane, projectOnPlane, setBounds' state) are PINNED against the
 *     reference's own CameraControls.cpp (tests/golden/camera_controls.json) and the glm calls it adds against glm itself;
 *     update() polls GLFW and cannot run here, so the ~60 statements that string the blocks together are restated by hand:
 *     a session as a whole is PARITY UNPINNED.
 *   - SH basis: PINNED against the closed forms printed by the reference's sh_gen.py
 *     (tests/golden/sh_basis.json).
 *   - Kernel stages (clip data, tile list, sort, ranges, raster): the reference ships no
 *     tests, fixtures or golden images and its only implementation is CUDA, which cannot be
 *     built here -> PARITY UNPINNED for those stages; the restatement follows the cited
 *     line ranges of GaussianRender.cu and is checked by analytic known answers only.
 *
 * All matrices are glm layout: 16 floats, column-major, m[col*4+row].
 * All "float4" arrays are flat float arrays with 4 floats per splat.
 */
#ifndef GSREF_H
#define GSREF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Field-for-field the reference's CameraData (GaussianRender.cuh:17-26), 224 bytes. */
typedef struct GsRefCameraData
{
    float viewProjection[16];
    float projection[16];
    float view[16];
    float fovCotangent[2];
    float depthScaleBias[2];
    float position[3];
    float aspect;
} GsRefCameraData;

/* Per-stage wall-clock of one gsref_render() call, milliseconds, plus list sizes. */
typedef struct GsRefFrameStats
{
    double msSphericalHarmonics;
    double msClipData;
    double msBuildTileList;
    double msSortTileList;
    double msTileRanges;
    double msRasterize;
    double msTotal;
    int64_t tileListSize;     /* P   : (splat,tile) pairs emitted                     */
    int64_t consumedPairs;    /* P_c : list entries the raster loaded before stopping */
    int64_t nonEmptyTiles;
} GsRefFrameStats;

/* GaussianRender.cu:55-58 / PlyParser.cpp:155-165 */
uint32_t gsref_encode_vec4(const float v[4]);
void gsref_decode_vec4(uint32_t bits, float out[4]);

/* GaussianRender.cu:446-454 (with the explicit 0xFFFFFFFF clamp of SURVEY hard part 5). */
uint64_t gsref_get_key(int32_t tileIndex, float clipDepth);

/* GaussianRender.cu:62-155: colour = clamp(sum_k Y_k(dir) * sh[(3k+c)*stride] + 0.5, 0, 1). */
void gsref_spherical_harmonics(int l, const float dir[3], const float* sh, int stride, float rgbOut[3]);

/* Demo.cpp:160-194 */
void gsref_realign_spherical_harmonics(const float* src, float* dst, int groupSize, int shCount, int splatCount);
size_t gsref_realigned_sh_size(int groupSize, int shCount, int splatCount);

/* GaussianRender.cu:158-182.  sh is the realigned (group of 256) layout. */
void gsref_evaluate_spherical_harmonics(int32_t count, const float* position4, const float* sh, int shDegree,
                                        int shCount, const float cameraPosition[3], float* color4);

/* GaussianRender.cu:204-329, generalised to (width,height) by the rules in DESIGN.md. */
void gsref_evaluate_splat_clip_data(int32_t count, const float* position4, const float* scaleAndRotation4,
                                    const GsRefCameraData* camera, int width, int height, float* clipXY2,
                                    float* clipZ, float* screenEllipse4, float* conic4);

/* GaussianRender.cu:456-770 as the set of section 3.4 of SURVEY.md, emitted in
 * (splat index, local tile index) order.  Only tile rows [tileRowBegin, tileRowEnd) are emitted
 * (0,0 = all rows).  Returns the number of pairs the list would hold (may exceed capacity);
 * at most `capacity` pairs are written. */
int64_t gsref_build_tile_list(int32_t count, const float* clipXY2, const float* clipZ, const float* screenEllipse4,
                              int width, int height, int tileRowBegin, int tileRowEnd, uint64_t* keys,
                              int32_t* values, int64_t capacity);

/* GaussianRender.cu:806-855: stable sort on key bits [0, endBit). In place. */
void gsref_sort_tile_list(int64_t size, uint64_t* keys, int32_t* values, int endBit);
int gsref_sort_end_bit(int width, int height);

/* GaussianRender.cu:857-888. tileRange must be pre-set to -1 by the caller (Demo.cpp:403). */
void gsref_evaluate_tile_ranges(int64_t size, const uint64_t* keys, int32_t* tileRange);

/* GaussianRender.cu:908-1016. backBuffer (width*height*4 bytes) must be zeroed by the caller (Demo.cpp:399).
 * Only tile rows [tileRowBegin, tileRowEnd) are rasterised (0,0 = all).  Returns P_c. */
int64_t gsref_rasterize_tiles(int width, int height, int tileRowBegin, int tileRowEnd, const int32_t* tileRange,
                              const int32_t* values, const float* clipXY2, const float* conic4,
                              const float* color4, uint8_t* backBuffer);

/* Demo.cpp:394-479 as one call.  shDegree==0 -> colour is used as is.  Scratch is allocated inside.
 * Returns 0 on success. */
int gsref_render(int32_t count, const float* position4, const float* scaleAndRotation4, float* color4,
                 const float* sh, int shDegree, int shCount, const GsRefCameraData* camera, int width,
                 int height, uint8_t* backBuffer, GsRefFrameStats* stats);
/* The same frame on `threads` host threads (the optional all-core baseline of SURVEY.md section 8(d)): every stage is the serial routine on
 * a slice of its input, so image, list and counts are the serial frame's bit for bit. */
int gsref_render_threads(int32_t count, const float* position4, const float* scaleAndRotation4, float* color4, const float* sh,
                         int shDegree, int shCount, const GsRefCameraData* camera, int width, int height, uint8_t* backBuffer,
                         GsRefFrameStats* stats, int threads);

/* CameraControls.cpp:132-146 + 79-96 + Demo.cpp:377-392: default pose for a bounding box. */
void gsref_camera_from_bounds(const float boundsMin[3], const float boundsMax[3], int width, int height,
                              GsRefCameraData* out);
/* Same projection constants for an arbitrary pose: position + unit quaternion (x,y,z,w). */
void gsref_camera_from_pose(const float position[3], const float rotationXYZW[4], int width, int height,
                            GsRefCameraData* out);

/* PlyParser.cpp:15-339.  Returns vertex count (>=0) or -1 with a message in err (the reference throws
 * std::invalid_argument with the same text). Output arrays are malloc'ed; free with gsref_free. */
typedef struct GsRefPly
{
    int32_t vertexCount;
    int32_t shDegree;
    int32_t shCount;
    float boundsMin[3];
    float boundsMax[3];
    float* position4;
    float* scaleAndRotation4;
    float* color4;
    float* sh; /* [vertex][shCount], NOT realigned; NULL when degree 0 */
} GsRefPly;
int gsref_parse_ply(const char* path, GsRefPly* out, char* err, size_t errLen);
void gsref_free_ply(GsRefPly* ply);

int gsref_hardware_concurrency(void);

/* GaussianRender.cu:8-9 EPANECHNIKOV_KERNEL as a run-time switch (0 = Gaussian footprint, the default). */
void gsref_set_footprint(int epanechnikov);
int gsref_get_footprint(void);

#ifdef __cplusplus
}
#endif
#endif
