/*
 * gsplat_hip.h -- C ABI of libgsplat_hip.so, the MI355X (gfx950) 3D-Gaussian-splat rasterizer.
 *
 * Drop-in boundary for the per-frame pipeline of etienne-p/CUDAGaussianRenderer
 * (src/GaussianRender.cuh:118-143 + the frame loop src/Demo.cpp:394-479) and its PLY ingest
 * (src/PlyParser.h:11-21).  Plain pointers and sizes only; every function cites the reference
 * interface it replaces.  A header-only C++ facade with the reference's own names
 * (setGlobalArgs, evaluateSplatClipData, DoubleBuffer<T>, DeviceBuffer<T>, ...) is in
 * GaussianRender.hpp; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions kept from the reference:
 *   - the caller owns all device memory; the library keeps raw pointers until the next set-args call;
 *   - the staged calls keep their arguments in per-host-thread state (the reference keeps them in file-scope __constant__
 *     symbols): one host thread runs one frame at a time, exactly as in the reference; different threads may each drive
 *     their own GsRenderer concurrently (device, stream and global/tile-list args are thread-local; everything that shapes a
 *     renderer's frames -- footprint, pipeline, storage order, depth layers, views, band -- is renderer state; the process-wide
 *     switches gs_set_footprint and gs_set_staged_builder are the defaults of the staged calls, as the reference's #define is);
 *   - a failed HIP API call prints "HIP error at <file> <line>: <msg>" and exit(EXIT_FAILURE)
 *     (src/CudaErrorCheck.cpp:2-11); a failed kernel launch prints "kernel error <stage>" and continues;
 *   - tileRange must be pre-set to 0xFF and the back buffer zeroed by the caller each frame
 *     (Demo.cpp:399,403); evaluateTileRange must be skipped when the list is empty (Demo.cpp:469).
 * Additions (absent in the reference because they are compile-time there): width/height, the
 * tile-row band used by the multi-GPU split, and the stream the stages are enqueued on.
 * Diagnostics, A/B switches, counters and trace hooks are NOT part of the boundary: they live in gsplat_hip_debug.h.
 *
 * Matrices are glm layout: 16 floats, column-major, m[col*4+row].
 */
#ifndef GSPLAT_HIP_H
#define GSPLAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_TILE_SIZE 16            /* Consts.h:5  k_TileSize                                   */
#define GS_DEFAULT_SCREEN_SIZE 1024 /* Consts.h:4  k_ScreenSize (used when width/height are 0) */
#define GS_SH_GROUP_SIZE 256       /* Demo.cpp:294 / GaussianRender.cu:188: SH realign group   */

/* ---- POD argument structs ------------------------------------------------------------------ */

/* GaussianRender.cuh:17-26 CameraData, field for field (224 bytes). */
typedef struct GsCameraData
{
    float viewProjection[16];
    float projection[16];
    float view[16];
    float fovCotangent[2];
    float depthScaleBias[2];
    float position[3];
    float aspect;
} GsCameraData;

/* GaussianRender.cuh:56-74 GlobalArgs, field for field, then the additions. Device pointers. */
typedef struct GsGlobalArgs
{
    GsCameraData cameraData;
    float* position;              /* float4[N]  xyz + opacity                                  */
    float* scaleAndRotation;      /* float4[N]  scale xyz + 8-bit x4 packed quaternion         */
    float* color;                 /* float4[N]  rgb + opacity                                  */
    int32_t sphericalHarmonicsDegree;
    int32_t sphericalHarmonicsCount;
    float* sphericalHarmonics;    /* float[ceil(N/256)*256*shCount], realigned (Demo.cpp:160-194) */
    float* positionClipSpaceXY;   /* float2[N]                                                 */
    float* positionClipSpaceZ;    /* float[N]                                                  */
    float* conic;                 /* float4[N]                                                 */
    float* screenEllipse;         /* float4[N]  cos, sin, extent.x, extent.y                   */
    int32_t splatCount;
    uint8_t* backBuffer;          /* uchar4[width*height], row 0 = clip y -1                   */
    int32_t* tileRange;           /* int32[2*tiles]                                            */
    /* additions */
    int32_t width;                /* 0 -> 1024                                                 */
    int32_t height;               /* 0 -> 1024                                                 */
    int32_t tileRowBegin;         /* multi-GPU band [begin,end) of tile rows; 0,0 -> all rows  */
    int32_t tileRowEnd;
} GsGlobalArgs;

/* GaussianRender.cuh:77-83 TileListArgs. */
typedef struct GsTileListArgs
{
    uint64_t* keys;
    int32_t* values;
    int32_t size;
    int32_t capacity;
} GsTileListArgs;

/* GaussianRender.cuh:87-116 DoubleBuffer<T> as plain structs. */
typedef struct GsDoubleBufferU64
{
    uint64_t* buffers[2];
    int32_t selector;
} GsDoubleBufferU64;

typedef struct GsDoubleBufferI32
{
    int32_t* buffers[2];
    int32_t selector;
} GsDoubleBufferI32;

/* Utilities.h:155-187 CudaTimer: event pair; stop() blocks until the stage has finished. */
typedef struct GsTimer GsTimer;
GsTimer* gs_timer_create(void);
void gs_timer_destroy(GsTimer* timer);
void gs_timer_start(GsTimer* timer);
void gs_timer_stop(GsTimer* timer);
float gs_timer_elapsed_ms(const GsTimer* timer);
/* Addition: a non-blocking timer only records events; elapsed is read after gs_synchronize(). */
void gs_timer_set_blocking(GsTimer* timer, int blocking);

/* ---- the eight staged host functions (GaussianRender.cuh:118-143) ---------------------------- */

void gs_set_global_args(const GsGlobalArgs* globalArgs);            /* setGlobalArgs      .cu:45-48   */
void gs_set_tile_list_args(const GsTileListArgs* tileListArgs);     /* setTileListArgs    .cu:50-53   */
void gs_evaluate_spherical_harmonics(GsTimer* timer, int32_t count); /* .cu:184-201                    */
void gs_evaluate_splat_clip_data(GsTimer* timer, int32_t count);    /* .cu:331-348                    */
/* buildTileList .cu:776-802: returns min(pairs emitted, capacity). numBlocks is accepted for
 * signature compatibility and ignored (the grid is sized from the splat count).  The list is the reference's SET of
 * (getKey(tile, depth), splat) pairs; the order in which the reference's kernel appends them is undefined (atomics, .cu:674,689) and
 * this one emits them in the order the sort leaves them -- see gs_set_staged_builder.  Needs library-owned scratch (grown on demand,
 * per host thread). */
int32_t gs_build_tile_list(GsTimer* timer, int32_t numBlocks, int32_t tileListCapacity);
/* Addition: how gs_build_tile_list emits.
 * SORTED (default): the renderer's two-level sort behind the reference's two calls -- the splats are sorted by getKey's depth first
 * (three radix passes over N elements), the pairs are emitted in that order with the first pass of the tile sort inside the emit pass,
 * the second pass ends in the caller's value buffer and the 64-bit keys are written beside it: the list leaves buildTileList in
 * (tile, depth, splat index) order.  gs_sort_tile_list then looks before it sorts (one pass over the keys): a stable sort of a
 * list that is already in order on the compared bits is the identity, so it returns with the selectors untouched; any other list is
 * sorted as before.  Same sorted list as REFERENCE_ORDER + sort, bit for bit (keys and values), at 0.23 instead of 0.62 ms for the two
 * calls at BASELINE.json configs[2] (profiles/r05_staged_path.txt).  A list cut off by the capacity is a sorted subset of the full
 * list (the reference's is an arbitrary subset, .cu:700-703).  FOOTPRINT: this builder keeps library-owned scratch per host thread,
 * sized by the CALLER'S capacity, not by the demand -- 16 bytes per splat + 16 (20 above 65 536 tiles) bytes per list entry of
 * tileListCapacity: 0.13 GB at the reference's 8 entries per splat for 1 M splats, 0.64 GB at a capacity of 40 M -- freed by
 * gs_release_thread_scratch.  When that memory cannot be had the call prints one line and emits in REFERENCE_ORDER instead (which
 * needs no scratch beyond 8 bytes per 64 splats); it does not end the process.
 * REFERENCE_ORDER: (splat, local tile) emission order (gs_tiles.hip, the order the oracle emits in) and an unconditional sort. */
#define GS_STAGED_BUILDER_REFERENCE_ORDER 0
#define GS_STAGED_BUILDER_SORTED 1
void gs_set_staged_builder(int mode);
int gs_get_staged_builder(void);
/* Addition: the tile start order of an interactive camera for the staged loop (the literal drop-in host IS the reference's viewer,
 * Demo.cpp:371-479).  PREVIOUS_FRAME: gs_rasterize_tile starts its tiles longest-first by the entries they consumed in the calling
 * thread's previous frame; the table is made by the first eight workgroups of gs_evaluate_splat_clip_data's kernel, withdrawn by itself
 * when consecutive frames differ and not built when the camera (GsGlobalArgs.cameraData.view) turned by more than ~3 degrees -- see
 * gs_renderer_set_tile_start_order below for the mechanism and the numbers (raster 62 -> 50 us at 1-degree steps; same image bytes).  Keeps
 * per-thread scratch of 8 bytes per tile + 4 per grid slot (freed by gs_release_thread_scratch).  Default SCREEN (GS_TILE_START_ORDER_*
 * are defined below).  Process-wide switch, per-thread history. */
void gs_set_staged_tile_start_order(int mode);
int gs_get_staged_tile_start_order(void);
/* Addition: frees the scratch the staged calls of the CALLING host thread allocated behind the scenes (state is per host thread, like
 * the reference's file-scope symbols are per process); the next staged call allocates again.  Waits for the thread's stream first. */
void gs_release_thread_scratch(void);
/* Addition: pairs the last gs_build_tile_list would have emitted with unlimited capacity. */
int64_t gs_last_tile_list_demand(void);
/* sortTileList .cu:806-855: stable sort on key bits [0, 32+ceil(log2(tiles))) (rocPRIM radix sort; skipped when the list is
 * already in order on those bits, see gs_set_staged_builder). May free and re-allocate *deviceTempStorage (hipFree/hipMalloc), as
 * the reference does with cudaFree/cudaMalloc. */
void gs_sort_tile_list(GsTimer* timer, int32_t tileListSize, void** deviceTempStorage, size_t* tempStorageSizeInBytes,
                       GsDoubleBufferU64* keys, GsDoubleBufferI32* values);
void gs_evaluate_tile_range(GsTimer* timer, int32_t tileListSize);  /* .cu:891-906                    */
void gs_rasterize_tile(GsTimer* timer);                             /* .cu:1019-1034                  */
/* The reference's compile-time option `#define EPANECHNIKOV_KERNEL` (GaussianRender.cu:8-9, README.md:133-136) as a
 * run-time switch: footprint extent sqrt(7) sigma instead of 3 sigma (.cu:298-302), density 1 - q/7 instead of exp(-q/2)
 * (.cu:977-981).  Process-wide, like the reference's #define: it applies to the staged calls and is the DEFAULT of every renderer;
 * a renderer can be given a footprint of its own (gs_renderer_set_footprint below), so that two host threads with different
 * footprints do not share a switch. */
#define GS_FOOTPRINT_GAUSSIAN 0
#define GS_FOOTPRINT_EPANECHNIKOV 1
void gs_set_footprint(int footprint);
int gs_get_footprint(void);

/* ---- stream / device (additions) ------------------------------------------------------------- */
void gs_set_device(int device);
void gs_set_stream(void* hipStream); /* NULL = the null stream, as the reference */
void* gs_get_stream(void);
void gs_synchronize(void);           /* Demo.cpp:478 cudaDeviceSynchronize -> stream synchronize */
int gs_device_info(char* name, size_t nameLen, int* computeUnits, size_t* totalMemBytes);

/* ---- DeviceBuffer<T> (DeviceBuffer.cu:8-127) -------------------------------------------------- */
void* gs_device_alloc(size_t bytes);
void gs_device_free(void* ptr);
void gs_device_upload(void* dst, const void* srcHost, size_t bytes);
void gs_device_download(void* dstHost, const void* src, size_t bytes);
void gs_device_memset(void* ptr, int value, size_t bytes); /* enqueued on the current stream */

/* ---- host-side helpers of the path -------------------------------------------------------------- */
uint32_t gs_encode_vec4(const float v[4]);                        /* PlyParser.cpp:155-165 */
int gs_sort_end_bit(int width, int height);                       /* .cu:818-819 generalised */
void gs_tile_counts(int width, int height, int* tilesX, int* tilesY);
size_t gs_realigned_sh_size(int groupSize, int shCount, int splatCount);
void gs_realign_spherical_harmonics(const float* src, float* dst, int groupSize, int shCount, int splatCount); /* Demo.cpp:160-194 */
/* CameraControls.cpp:132-146 default pose + Demo.cpp:377-392 CameraData fill (near .1, far 100, fov 60 deg). */
void gs_camera_from_bounds(const float boundsMin[3], const float boundsMax[3], int width, int height, GsCameraData* out);
void gs_camera_from_pose(const float position[3], const float rotationXYZW[4], int width, int height, GsCameraData* out);

/* ---- interactive camera without a window: class CameraControls, CameraControls.h:7-76 ------------------
 * The reference polls GLFW inside CameraControls::update (glfwGetMouseButton / glfwGetCursorPos / glfwGetKey,
 * CameraControls.cpp:44-72,116-130,176-177); here the caller hands the same three things in per frame, so a headless host can
 * replay a recorded session and a windowed one can fill them from its toolkit.  Same state machine, same arithmetic in glm's
 * operand order: a session gives the reference's poses bit for bit (csrc/gs_camera.cpp).  Host code, no GPU involved. */
enum { GS_MOUSE_LEFT = 1, GS_MOUSE_MIDDLE = 2, GS_MOUSE_RIGHT = 4 };                              /* GLFW_MOUSE_BUTTON_LEFT/MIDDLE/RIGHT pressed */
enum { GS_KEY_W = 1, GS_KEY_A = 2, GS_KEY_S = 4, GS_KEY_D = 8, GS_KEY_Q = 16, GS_KEY_E = 32 };    /* GLFW_KEY_* pressed, CameraControls.cpp:44-72 */
typedef struct GsCameraInput
{
    uint32_t mouseButtons; /* GS_MOUSE_* bits; left wins over middle wins over right (CameraControls.cpp:116-130): drag / orbit / pan */
    uint32_t keys;         /* GS_KEY_* bits */
    double cursorX;        /* glfwGetCursorPos: pixels from the window's top-left corner */
    double cursorY;
} GsCameraInput;
typedef struct GsCameraControlsState
{
    float position[3];       /* m_Position */
    float rotationXYZW[4];   /* m_Rotation */
    float anchorPosition[3]; /* m_AnchorPosition: what orbit turns about; follows the pointer over the floor plane while no button is down */
    float speed;             /* m_Speed (set by setBounds; unused by update, as in the reference) */
    int32_t mode;            /* 0 none, 1 pan, 2 orbit, 3 drag (CameraControls.h:10-16) */
} GsCameraControlsState;
typedef struct GsCameraControls GsCameraControls;
GsCameraControls* gs_camera_controls_create(float screenWidth, float screenHeight);                /* CameraControls.h:46-49 */
void gs_camera_controls_destroy(GsCameraControls* controls);
void gs_camera_controls_set_bounds(GsCameraControls* controls, const float boundsMin[3], const float boundsMax[3]); /* CameraControls.cpp:132-146 */
void gs_camera_controls_update(GsCameraControls* controls, const GsCameraInput* input, float dt); /* CameraControls.cpp:148-253 */
void gs_camera_controls_camera_data(const GsCameraControls* controls, GsCameraData* out);         /* getters + Demo.cpp:377-392 */
void gs_camera_controls_get_state(const GsCameraControls* controls, GsCameraControlsState* out);
/* Addition (the reference only places the camera through setBounds): start a session from a saved pose. */
void gs_camera_controls_set_pose(GsCameraControls* controls, const float position[3], const float rotationXYZW[4]);

/* ---- GL interop: CudaGraphicsResource (Utilities.h:113-153), SURVEY 8(f4) --------------------------
 * The one CUDA-specific piece of the reference's presentation: its GL pixel-unpack buffer is registered with the device
 * (Demo.cpp:311,338), mapped once per frame (:396) and the mapped pointer is the frame's back buffer (:419); the texture upload and the
 * quad of Demo.cpp:481-525 are plain GL and stay with the host unchanged.  A GL buffer id goes in, a device pointer comes out -- what
 * GsGlobalArgs.backBuffer and gs_renderer_set_back_buffer take; the host needs no HIP header.  A GL context must be current on the
 * calling thread (the reference's glfwMakeContextCurrent, Demo.cpp:230).  The GPU pool this library is tested on has no window system:
 * what is tested there is that a registration without a context is refused cleanly; the mapped path is compiled, not run. */
typedef struct GsGraphicsResource GsGraphicsResource;
#define GS_GRAPHICS_REGISTER_FLAGS_NONE 0          /* cudaGraphicsRegisterFlagsNone (Demo.cpp:338) */
#define GS_GRAPHICS_REGISTER_FLAGS_READ_ONLY 1     /* cudaGraphicsRegisterFlagsReadOnly */
#define GS_GRAPHICS_REGISTER_FLAGS_WRITE_DISCARD 2 /* cudaGraphicsRegisterFlagsWriteDiscard */
/* Utilities.h:139-142 cudaGraphicsGLRegisterBuffer.  The reference ignores that call's result and ends at the first map
 * (checkCudaErrors, :127); here a failed registration prints its reason and returns NULL -- and mapping NULL ends the process the same way. */
GsGraphicsResource* gs_graphics_gl_register_buffer(uint32_t glBufferId, uint32_t flags);
/* Utilities.h:122-130 Binding(): cudaGraphicsMapResources (on the library's current stream; the reference: stream 0) +
 * cudaGraphicsResourceGetMappedPointer.  Errors print and exit, like checkCudaErrors. */
void* gs_graphics_map(GsGraphicsResource* resource, size_t* sizeInBytesOrNull);
void gs_graphics_unmap(GsGraphicsResource* resource);      /* Utilities.h:131-134 ~Binding(): unchecked there and here */
void gs_graphics_unregister(GsGraphicsResource* resource); /* Utilities.h:143-146; unmaps first if still mapped; NULL is a no-op */

/* Demo.cpp:119-143 generateRandomGaussians with a seeded PRNG (splitmix64) instead of std::rand. */
void gs_generate_random_gaussians(int32_t count, uint64_t seed, float minScale, float maxScale, const float minPosition[3],
                                  const float maxPosition[3], float* position4, float* scaleAndRotation4, float* color4);

/* ---- PLY ingest (PlyParser.h:13-21 parsePly) ------------------------------------------------- */
typedef struct GsPly
{
    int32_t vertexCount;
    int32_t sphericalHarmonicsDegree;
    int32_t sphericalHarmonicsCount;
    float boundsMin[3];
    float boundsMax[3];
    float* position;            /* host, float4[N] */
    float* scaleAndRotation;    /* host, float4[N] */
    float* color;               /* host, float4[N] */
    float* sphericalHarmonics;  /* host, [N][shCount] (not realigned), NULL at degree 0 */
} GsPly;
/* Returns the vertex count, or -1 and the reference's std::invalid_argument text in err. */
int gs_parse_ply(const char* filePath, GsPly* out, char* err, size_t errLen);
void gs_free_ply(GsPly* ply);

/* ---- one-call renderer: the frame loop body Demo.cpp:352-479 as render(splats, camera) ------- */
typedef struct GsRenderer GsRenderer;

typedef struct GsFrameStats
{
    float msSphericalHarmonics, msClipData, msBuildTileList, msSortTileList, msTileRanges, msRasterize; /* Demo.cpp:149-157 */
    float msFrame;            /* first enqueue -> last kernel done, GPU time */
    int64_t tileListSize;     /* P   */
    int64_t consumedPairs;    /* P_c (only with gs_set_collect_raster_stats) */
    int32_t tileListCapacity;
    int32_t grewTileList;     /* 1 when the list buffers were re-allocated this frame */
} GsFrameStats;

/* Limits (each refused instead of letting a 32-bit index wrap; the reference's own interface is int32 throughout): at most 65 536
 * pixels per axis (gs_renderer_create returns NULL with a message), at most 2^30 splats x views per chain (gs_renderer_set_scene /
 * gs_renderer_share_scene return -1), at most 0x7FF00000 list entries per frame (gs_renderer_render returns -2).  Tested up to
 * 16 384 x 16 384, 200 M splats and 1.72e9 entries (profiles/r06_large_lists.txt). */
GsRenderer* gs_renderer_create(int width, int height);
void gs_renderer_destroy(GsRenderer* r);
/* The whole of Demo.cpp:278-479 for ONE frame as one call (SURVEY.md section 8(b)'s fused convenience): the scene as parsePly returns it
 * (host arrays; sphericalHarmonics as [splat][count] rows, NULL with degree 0), a camera, a size -- upload, render, and the RGBA8 image
 * (row 0 = clip y -1, GaussianRender.cu:1014) copied to rgbaOutHost when that is not NULL.  Returns 0, or the code of the call that
 * failed.  A renderer is created and destroyed inside: a loop over cameras keeps a GsRenderer instead (gs_renderer_set_scene once). */
typedef struct GsScene
{
    int32_t splatCount;
    const float* position;         /* float4[n]: xyz, opacity */
    const float* scaleAndRotation; /* float4[n]: scale xyz, packed quaternion */
    const float* color;            /* float4[n] */
    const float* sphericalHarmonics;
    int32_t sphericalHarmonicsDegree, sphericalHarmonicsCount;
} GsScene;
int gs_render(const GsScene* scene, const GsCameraData* camera, int width, int height, uint8_t* rgbaOutHostOrNull);
/* Multi-GPU: this renderer only builds/sorts/rasterises tile rows [begin,end). 0,0 = all. */
void gs_renderer_set_tile_rows(GsRenderer* r, int tileRowBegin, int tileRowEnd);
/* Two pipelines give the same frame. REFERENCE runs the eight staged calls in the reference's order (one
 * radix sort of all pairs on the 64-bit tile|depth key). TWO_LEVEL (default) sorts the splats by the 32-bit
 * depth key, emits the pairs in that order keyed by tile index, and sorts them stably on the tile bits only:
 * same list order (ties by splat index), ~5x less radix-sort traffic. */
/* Band cull (TWO_LEVEL pipeline): when the renderer owns a strict sub-band of tile rows, only the splats whose tile
 * AABB reaches the band are SH-evaluated, depth-sorted and expanded. AUTO = on exactly then; ON/OFF force it. */
#define GS_BAND_CULL_AUTO 0
#define GS_BAND_CULL_ON 1
#define GS_BAND_CULL_OFF 2
void gs_renderer_set_band_cull(GsRenderer* r, int mode);
#define GS_PIPELINE_REFERENCE 0
#define GS_PIPELINE_TWO_LEVEL 1
/* Returns 0; -1 (nothing changed) for REFERENCE while the renderer holds a scene stored in Morton order (below): that pipeline
 * breaks depth ties by storage order.  A scene set while REFERENCE is selected is always stored in the caller's order. */
int gs_renderer_set_pipeline(GsRenderer* r, int pipeline);
/* This renderer's footprint: GS_FOOTPRINT_GAUSSIAN / GS_FOOTPRINT_EPANECHNIKOV, or -1 (default) = the process-wide setting at the
 * time of each frame (gs_set_footprint).  Renderer state: frames of different renderers, driven from different host threads, may use
 * different footprints at the same time.  A captured frame is re-captured when its footprint changes. */
void gs_renderer_set_footprint(GsRenderer* r, int footprintOrMinus1);
int gs_renderer_footprint(const GsRenderer* r); /* the footprint the next frame uses */
/* 1 when the renderer's frames evaluate SH colours on demand inside the raster kernel instead of running the SH pass over all
 * splats (two-level pipeline, SH degree 1..4, two pixels per lane; GS_NO_LAZY_SH=1 in the environment turns it
 * off).  Same colours bit for bit (one shared routine); a frame consumes only the front of its lists -- 88 K distinct splats
 * of 1 M at BASELINE.json configs[2] -- so most of the reference's evaluateSphericalHarmonics work is never needed.
 * GsRendererBuffers.color is brought up to date (one SH pass) when gs_renderer_buffers is called. */
int gs_renderer_colours_on_demand(const GsRenderer* r);
/* Addition: DEPTH LAYERS for captured frames.  A frame lists every (splat, tile) pair of the scene, but its tiles stop blending long
 * before their lists end (GaussianRender.cu:995: every pixel below T = 0.02) -- at BASELINE.json configs[2] 0.39 M of 5.45 M pairs are
 * ever consumed.  With `layers` > 1 the depth-sorted splats are listed, sorted and blended in that many layers, nearest first (layer l
 * = sorted splats [fraction[l-1], fraction[l]) of the frame's; `fractions` = the layers - 1 ascending upper bounds, NULL = defaults),
 * and a later layer lists no pair of a tile that an earlier layer's votes already stopped.  A tile that is still open when a layer's
 * list ends leaves its per-pixel sums and transmittance and goes on in the next layer that has entries for it, its votes still at
 * multiples of 32 entries of its whole list: the image is the same bytes as with one layer.  What changes: the pair list the renderer
 * hands out (gs_renderer_buffers: keys, values, ranges, tileListSize) is the LAST layer's only (gs_renderer_listed_pairs: all layers).  Takes effect on
 * captured frames of the two-level pipeline whose tile sort is a two-pass sort of 16-bit indices (list fusion on); tile-row bands with
 * their cull included (the layers' slices of the culled list are cut on the device).
 * layers = 1 (the default): one list per frame, as the reference has it.
 * layers = GS_DEPTH_LAYERS_CHOOSE (0): the renderer chooses from the previous frame's pair total, for a host that renders ONE frame at a
 * time -- one layer below 16 M pairs per frame, two from there (27 M pairs: -16 % frame time), three from 40 M (BASELINE.json configs[4]
 * on one GPU: 1 258 -> 965 us of kernels per frame, profiles/r05_depth_layers.txt).
 * layers = GS_DEPTH_LAYERS_CHOOSE_PIPELINED (-1): the same for a host that keeps SEVERAL frames in flight on the GPU (several renderers on
 * their own streams, as bench.py's timed loop does): two layers already from 4.5 M pairs per frame -- BASELINE.json configs[2] (5.45 M):
 * +4 % frames/s with four frames in flight (10.9 M: +22 %, configs[4]: 870 -> 1 500 frames/s), where a lone frame would lose 14 % (ten
 * more launches on one stream; at 10.9 / 12.8 M pairs it still loses 3 / 14 %), hence the two forms.  Under -1 a scene below 8 M pairs
 * whose first layered frame lists more than half of its pairs (low opacities: the tiles hardly stop) goes back to one layer and is not
 * tried again for 256 frames (doubling). */
#define GS_DEPTH_LAYERS_CHOOSE 0
#define GS_DEPTH_LAYERS_CHOOSE_PIPELINED (-1)
int gs_renderer_set_depth_layers(GsRenderer* r, int layers, const float* fractions);
int gs_renderer_depth_layers(const GsRenderer* r);        /* the setting (0 / -1 = the renderer chooses) */
int gs_renderer_depth_layers_in_use(const GsRenderer* r); /* layers the last captured frame was made of */
/* Pairs the last captured frame listed: summed over its depth layers (gs_last_tile_list_demand then holds the largest layer's total,
 * which is what the list capacity must hold); the frame's pair total with one layer.  Waits for the renderer's stream. */
int64_t gs_renderer_listed_pairs(GsRenderer* r);
/* Addition: the order in which the raster's tiles START.  SCREEN (default): the XCD-aware screen order.  PREVIOUS_FRAME: the tiles that
 * consumed the most list entries in the renderer's previous frame first (each on the XCD of its 8 x 4 block, neighbours' consumption
 * counted in) -- for the host the reference is: an interactive viewer whose camera moves a little per frame (Demo.cpp:371-392).  The
 * raster kernel's time is 36 us of full-chip work plus a tail of long tiles that happened to start late; which tiles are long is only
 * known by blending them, but the previous frame knows: 62.5 -> 48.1 us at 1-degree camera steps on BASELINE.json configs[2]
 * (profiles/r06_tile_order.txt).  One small kernel behind the frame's last raster launch makes the next frame's table, and puts it in
 * force only when the last two frames' per-tile consumption differed by less than 35 % (a wrong prediction is worse than none): a camera
 * that jumps gets the screen order by itself.  The order in which tiles start changes no byte of any frame.  Single-view renderers
 * (gs_renderer_set_views 1), plain and captured frames, depth layers included; returns 0, -1 for an unknown mode. */
#define GS_TILE_START_ORDER_SCREEN 0
#define GS_TILE_START_ORDER_PREVIOUS_FRAME 1
int gs_renderer_set_tile_start_order(GsRenderer* r, int mode);
int gs_renderer_tile_start_order(const GsRenderer* r);
/* Addition: how the renderer STORES the splats of the next gs_renderer_set_scene.  CALLER (default): in the caller's order, like the
 * reference (Demo.cpp:314-331).  MORTON: along a 3-D Morton curve through their positions -- the spatial ordering the reference's
 * author names as the cure for the raster's scattered gathers (README.md:138).  Frames are byte-identical either way, for every scene:
 * the input of the (stable) depth sort is laid out in the caller's order, so splats of equal depth code -- however many -- still blend
 * in the caller's index order (GaussianRender.cu:446-454: cub's sort is stable over emission order).  What changes is speed
 * (profiles/r04_ab_experiments.txt, "Morton storage"; round 3's tie-fix variant: profiles/r03_storage_order.txt) and the index space of everything per splat that the renderer hands out: GsRendererBuffers arrays
 * and the `values` of the pair list are in STORAGE order; gs_renderer_storage_order gives the map.  TWO_LEVEL pipeline only: a scene
 * set while GS_PIPELINE_REFERENCE is selected is stored in the caller's order. */
#define GS_STORAGE_ORDER_CALLER 0
#define GS_STORAGE_ORDER_MORTON 1
void gs_renderer_set_storage_order(GsRenderer* r, int order);
/* storageToSplat[k] = the caller's index of the splat stored in slot k (identity for CALLER); returns the splat count.  The array
 * (splat count entries) may be NULL to query the count. */
int32_t gs_renderer_storage_order(const GsRenderer* r, int32_t* storageToSplat);
/* The Morton order itself (host): storageToSplat for `count` positions (x, y, z, _) -- 21 bits per axis over the bounding box, equal
 * codes in index order. */
void gs_morton_order(const float* position4, int32_t count, int32_t* storageToSplat);
/* Upload a scene from host arrays (Demo.cpp:314-331). sh is [N][shCount] un-aligned (as parsePly
 * returns it) and is realigned here (Demo.cpp:291-296); may be NULL at degree 0. */
int gs_renderer_set_scene(GsRenderer* r, int32_t splatCount, const float* position4, const float* scaleAndRotation4,
                          const float* color4, const float* sh, int shDegree, int shCount);
/* Several renderers of ONE scene (frames in flight on separate streams): `r` uses `owner`'s read-only inputs (position, scale/rotation,
 * SH coefficients) instead of a copy of its own and allocates only its intermediates and lists.  `owner` must have its scene set, must
 * not be a sharer itself, and must outlive `r` (or `r` must get another scene first).  Returns 0, -1 on bad arguments. */
int gs_renderer_share_scene(GsRenderer* r, const GsRenderer* owner);
/* Render one frame into the renderer's device back buffer.
 *   timed == 0: no stage timers (throughput mode);
 *   timed == 1: per-stage timings in *stats, blocking after every stage like the reference's CudaTimer;
 *   timed == 2: one non-blocking event pair per stage on the stream; read the sums later with
 *               gs_renderer_collect_stage_times (the frames are not slowed by host waits).
 * Returns 0; -2 when the frame needs more than 0x7FF00000 (2.146e9) list entries -- the int32 list sizes of the reference's interface end
 * there (its own `tileListCapacity *= 2`, Demo.cpp:360, overflows from 2^30 entries on); the list grows by doubling up to that bound. */
int gs_renderer_render(GsRenderer* r, const GsCameraData* camera, int timed, GsFrameStats* statsOrNull);
/* Captured frames: gs_renderer_render (timed == 0, stats == NULL, TWO_LEVEL pipeline, native sort) replays the whole
 * frame as ONE hipGraph launch; list sizes stay on the device, the camera is refreshed by a memcpy node.  The call returns once
 * the frame's pair total has reached the host (mid-frame, through a host-mapped word); the next call may follow at once (two
 * instances of the graph with a camera slot each, so the stream always holds the next frame).  A frame whose demand exceeds the list
 * capacity is NOT handed out truncated (the reference shows it and doubles the list for the next frame, Demo.cpp:357-366,447):
 * the list is grown, the frame re-captured and the same camera rendered again before the call returns. */
void gs_renderer_set_graph(GsRenderer* r, int enabled);
/* Addition: several cameras per chain of kernels.  A renderer set to V views (1..4, BEFORE its scene is set or shared) renders V
 * frames of its scene with ONE replay of the captured chain: view v's splat i is treated as the splat v*N + i (own clip data,
 * shared inputs) and its tile t as the tile v*T + t, so one depth sort and one tile sort order the pairs of all V frames and every
 * kernel runs once per V frames -- what a frame of a tile-row band (multi-GPU) needs, whose kernels are too small to fill the
 * time their launches take.  Every image is byte-identical to the one gs_renderer_render gives for that camera.  The back buffer
 * holds the V images one after another (an external one must be that large); gs_renderer_back_buffer_view(r, v) is view v's.
 * gs_renderer_render_views takes V cameras; it needs captured frames (gs_renderer_set_graph) and the two-level pipeline and returns
 * -3 otherwise; every SH degree the reference ingests (0..4, PlyParser.cpp:204-243) is rendered -- the colours of degrees 1..4 are
 * evaluated by the raster kernel from the coefficient rows (a process that keeps the SH pass, GS_NO_LAZY_SH, is refused with -3 for
 * those: the pass would need one colour buffer per view); overflow of the pair list is handled as for single frames (all V frames
 * are rendered again).  Returns 0 on success.  After a chain of V > 1 views gs_renderer_buffers describes
 * the chain, not a frame: the pair list holds the pairs of all views (values are virtual splat indices v*N + i, ranges of view v start
 * at entry 2*v*T of tileRange), and the per-splat arrays it unpacks are view 0's. */
int gs_renderer_set_views(GsRenderer* r, int views);
int gs_renderer_views(const GsRenderer* r);
int gs_renderer_render_views(GsRenderer* r, const GsCameraData* cameras);
uint8_t* gs_renderer_back_buffer_view(GsRenderer* r, int view);
int gs_renderer_read_back_buffer_view(GsRenderer* r, int view, uint8_t* hostRgba); /* waits for the renderer's stream; -1: no such view */
/* The same frame in two halves, so that one host thread can keep several renderers (each with its own stream and
 * buffers, gs_renderer_set_stream) busy: begin() enqueues everything up to the pair emission and returns; finish()
 * picks up the pair count -- by then on the host -- and enqueues tile sort, ranges and raster. render() = begin + finish. */
void gs_renderer_set_stream(GsRenderer* r, void* hipStream);
int gs_renderer_begin(GsRenderer* r, const GsCameraData* camera, int timed);
int gs_renderer_finish(GsRenderer* r, GsFrameStats* statsOrNull);
/* Sums of the per-stage times (ms) and of the list sizes over the timed==2 frames since the last call.
 * Returns the number of frames summed. */
int gs_renderer_collect_stage_times(GsRenderer* r, GsFrameStats* sums);
/* Device pointer of the uchar4 back buffer (width*height*4 bytes), valid until destroy. */
uint8_t* gs_renderer_back_buffer(GsRenderer* r);
/* Render into a caller-provided device buffer instead (e.g. a mapped interop buffer, Demo.cpp:396). */
void gs_renderer_set_back_buffer(GsRenderer* r, uint8_t* deviceBackBufferOrNull);
/* Copy the back buffer to host (blocking). */
void gs_renderer_read_back_buffer(GsRenderer* r, uint8_t* hostRgba);
/* Device pointers of the per-splat intermediates, for tests/inspection.  The two-level frames keep centre, depth, conic and
 * ellipse inside two packed 32-byte records per splat; this call unpacks them into the reference's arrays (and brings the colour buffer
 * up to date when the colours were evaluated on demand). */
typedef struct GsRendererBuffers
{
    float *position, *scaleAndRotation, *color, *sphericalHarmonics;
    float *positionClipSpaceXY, *positionClipSpaceZ, *conic, *screenEllipse;
    int32_t* tileRange;
    void* keys;       /* sorted keys: uint64 tile|depth (keyBytes 8) or uint16/uint32 tile index (2/4) */
    int32_t* values;  /* sorted splat indices */
    int32_t splatCount, tileListSize;
    int32_t keyBytes;
} GsRendererBuffers;
void gs_renderer_buffers(GsRenderer* r, GsRendererBuffers* out);

/* ---- multi-GPU: framebuffer gather (no counterpart in the single-GPU reference) --------------------
 * One process per GPU.  Rank r renders tile rows [bands[2r], bands[2r+1]) of the frame (gs_renderer_set_tile_rows) into its own
 * full-size back buffer; gs_gather_bands then moves every band's pixel rows to `root` with ONE grouped ncclSend/ncclRecv set,
 * enqueued on `hipStream` (use the stream the frame was rendered on: the transfer then waits for the raster kernel, and the next
 * frame of that renderer waits for the transfer).  The root's receives land directly in the destination rows of its back buffer.
 * RCCL (librccl.so.1) is loaded on first use; a world of 1 needs no RCCL and every call is a no-op.
 * Bootstrap: rank 0 calls gs_comm_unique_id and hands the 128 bytes to the other ranks by any means (MPI, a file, a socket,
 * torch.distributed); every rank then calls gs_comm_create with its device current (gs_set_device). */
#define GS_COMM_UNIQUE_ID_BYTES 128
typedef struct GsComm GsComm;
int gs_comm_unique_id(char out[GS_COMM_UNIQUE_ID_BYTES]);                       /* 0, or -1 when RCCL is unavailable        */
GsComm* gs_comm_create(int rank, int world, const char uniqueId[GS_COMM_UNIQUE_ID_BYTES]); /* NULL on failure; id ignored at world 1 */
void gs_comm_destroy(GsComm* comm);
int gs_comm_rank(const GsComm* comm);
int gs_comm_world(const GsComm* comm);
/* Pixel rows [rowBegin, rowEnd) of a tile-row band; the last band may end on a partial tile row. */
void gs_band_pixel_rows(int tileRowBegin, int tileRowEnd, int height, int* rowBegin, int* rowEnd);
/* tileRowBands: 2 * world int32 (begin, end per rank).  Returns 0, -1 on bad arguments, -2 on an RCCL error. */
int gs_gather_bands(GsComm* comm, uint8_t* deviceBackBuffer, int width, int height, const int32_t* tileRowBands, int root, void* hipStream);
/* The same for the `views` images of a renderer set to several views per chain (gs_renderer_set_views: the images follow one another
 * in the back buffer): ONE grouped send/receive set for all of them. */
int gs_gather_bands_views(GsComm* comm, uint8_t* backBuffer, int width, int height, int views, const int32_t* tileRowBands, int root,
                          void* hipStream);

const char* gs_version(void);

#ifdef __cplusplus
}
#endif
#endif
