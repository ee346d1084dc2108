// gs_capi.hip -- the C ABI of include/gsplat_hip.h: library state, timers, device buffers, the eight
// staged host functions (GaussianRender.cuh:118-143) and the one-call renderer (Demo.cpp:352-479).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <vector>

#include "gs_internal.h"

namespace gs
{

void setRasterPixelsPerLane(int ppl);
int rasterPixelsPerLane();
void setRasterTrace(unsigned long long* deviceBufferOrNull);

namespace
{
int g_footprint = GS_FOOTPRINT_GAUSSIAN; // process-wide, like the reference's #define
int g_ablation = 0;                      // diagnostics: stages left out of captured frames (gs_debug_set_ablation)
}

Screen makeScreen(int width, int height, int rowBegin, int rowEnd)
{
    if (width <= 0) width = GS_DEFAULT_SCREEN_SIZE;
    if (height <= 0) height = GS_DEFAULT_SCREEN_SIZE;
    Screen s;
    s.width = width;
    s.height = height;
    s.tilesX = (width + kTileSize - 1) / kTileSize;
    s.tilesY = (height + kTileSize - 1) / kTileSize;
    s.totalTiles = s.tilesX * s.tilesY;
    if (rowBegin == 0 && rowEnd == 0) rowEnd = s.tilesY;
    s.rowBegin = std::min(std::max(rowBegin, 0), s.tilesY);
    s.rowEnd = std::min(std::max(rowEnd, s.rowBegin), s.tilesY);
    s.tilesPerScreenX = (float) width / (float) kTileSize;
    s.tilesPerScreenY = (float) height / (float) kTileSize;
    s.tileClipSizeX = ((float) kTileSize / (float) width) * 2.0f;
    s.tileClipSizeY = ((float) kTileSize / (float) height) * 2.0f;
    s.pixelClipSizeX = 2.0f / (float) width;
    s.pixelClipSizeY = 2.0f / (float) height;
    const float pi = 3.14159265359f;
    s.traceBump = (1.0f / pi) * s.pixelClipSizeX * s.pixelClipSizeY;
    s.epanechnikov = g_footprint == GS_FOOTPRINT_EPANECHNIKOV ? 1 : 0;
    s.extentFactor = s.epanechnikov ? sqrtf(7.0f) : 3.0f;
    return s;
}

namespace
{

// The reference keeps these in __constant__ symbols (GaussianRender.cu:42-43); here they are a host
// copy and every launch passes what its kernel needs by value.
struct State
{
    GsGlobalArgs global{};
    GsTileListArgs tileList{};
    hipStream_t stream = nullptr;
    TileListScratch scratch;
    RadixScratch radix;
    int64_t lastDemand = 0;
};
// One state per host thread: like the reference, the staged calls are not re-entrant, but each host thread may drive
// its own renderer / staged sequence on its own stream (bench.py runs one thread per frame in flight).
thread_local State g;

// Process-wide settings.
struct Settings
{
    int sortBackend = GS_SORT_BACKEND_NATIVE;
    bool onesweep = false; // NATIVE with single-kernel (look-back) passes
    bool collectRasterStats = false;
    unsigned long long* consumedDevice = nullptr; // counter of the staged gs_rasterize_tile
};
Settings cfg;

Screen currentScreen() { return makeScreen(g.global.width, g.global.height, g.global.tileRowBegin, g.global.tileRowEnd); }

ClipCamera currentCamera()
{
    ClipCamera c;
    std::memcpy(c.view, g.global.cameraData.view, sizeof(c.view));
    c.fovCotX = g.global.cameraData.fovCotangent[0];
    c.fovCotY = g.global.cameraData.fovCotangent[1];
    c.depthScale = g.global.cameraData.depthScaleBias[0];
    c.depthBias = g.global.cameraData.depthScaleBias[1];
    return c;
}

void reportLaunch(const char* stage)
{
    if (hipGetLastError() != hipSuccess) printf("kernel error %s\n", stage);
}

} // namespace
} // namespace gs

using namespace gs;

// ---- timer (Utilities.h:155-187) ------------------------------------------------------------------
struct GsTimer
{
    hipEvent_t start, stop;
    float elapsedMs;
    int blocking;
};

extern "C" GsTimer* gs_timer_create(void)
{
    GsTimer* t = new GsTimer{};
    GS_CHECK(hipEventCreate(&t->start));
    GS_CHECK(hipEventCreate(&t->stop));
    t->blocking = 1;
    return t;
}
extern "C" void gs_timer_destroy(GsTimer* t)
{
    if (!t) return;
    (void) hipEventDestroy(t->start);
    (void) hipEventDestroy(t->stop);
    delete t;
}
extern "C" void gs_timer_set_blocking(GsTimer* t, int blocking) { t->blocking = blocking; }
extern "C" void gs_timer_start(GsTimer* t)
{
    if (t) GS_CHECK(hipEventRecord(t->start, g.stream));
}
extern "C" void gs_timer_stop(GsTimer* t)
{
    if (!t) return;
    GS_CHECK(hipEventRecord(t->stop, g.stream));
    if (t->blocking)
    {
        GS_CHECK(hipEventSynchronize(t->stop));
        GS_CHECK(hipEventElapsedTime(&t->elapsedMs, t->start, t->stop));
    }
}
extern "C" float gs_timer_elapsed_ms(const GsTimer* t)
{
    if (!t->blocking)
    {
        float ms = 0.0f;
        GS_CHECK(hipEventSynchronize(t->stop));
        GS_CHECK(hipEventElapsedTime(&ms, t->start, t->stop));
        return ms;
    }
    return t->elapsedMs;
}

// ---- device / stream ------------------------------------------------------------------------------
extern "C" void gs_set_device(int device) { GS_CHECK(hipSetDevice(device)); }
extern "C" void gs_set_stream(void* s) { g.stream = (hipStream_t) s; }
extern "C" void* gs_get_stream(void) { return (void*) g.stream; }
extern "C" void gs_synchronize(void) { GS_CHECK(hipStreamSynchronize(g.stream)); }
extern "C" int gs_device_info(char* name, size_t nameLen, int* computeUnits, size_t* totalMemBytes)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    if (name && nameLen) snprintf(name, nameLen, "%s (%s)", prop.name, prop.gcnArchName);
    if (computeUnits) *computeUnits = prop.multiProcessorCount;
    if (totalMemBytes) *totalMemBytes = prop.totalGlobalMem;
    return 0;
}

// ---- DeviceBuffer (DeviceBuffer.cu) -----------------------------------------------------------------
extern "C" void* gs_device_alloc(size_t bytes)
{
    void* p = nullptr;
    GS_CHECK(hipMalloc(&p, bytes ? bytes : 1));
    return p;
}
extern "C" void gs_device_free(void* p)
{
    if (p) GS_CHECK(hipFree(p));
}
extern "C" void gs_device_upload(void* dst, const void* src, size_t bytes)
{
    if (bytes) GS_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
}
extern "C" void gs_device_download(void* dst, const void* src, size_t bytes)
{
    if (bytes)
    {
        GS_CHECK(hipStreamSynchronize(g.stream));
        GS_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    }
}
extern "C" void gs_device_memset(void* p, int value, size_t bytes)
{
    if (bytes) GS_CHECK(hipMemsetAsync(p, value, bytes, g.stream));
}

// ---- staged API -----------------------------------------------------------------------------------
extern "C" void gs_set_global_args(const GsGlobalArgs* a) { g.global = *a; }
extern "C" void gs_set_tile_list_args(const GsTileListArgs* a) { g.tileList = *a; }

extern "C" int gs_sort_end_bit(int width, int height) { return sortEndBit(makeScreen(width, height, 0, 0).totalTiles); }
extern "C" void gs_tile_counts(int width, int height, int* tilesX, int* tilesY)
{
    const Screen s = makeScreen(width, height, 0, 0);
    if (tilesX) *tilesX = s.tilesX;
    if (tilesY) *tilesY = s.tilesY;
}

extern "C" void gs_evaluate_spherical_harmonics(GsTimer* timer, int32_t count)
{
    gs_timer_start(timer);
    const float* cp = g.global.cameraData.position;
    launchSphericalHarmonics(g.stream, count, g.global.sphericalHarmonicsDegree, g.global.sphericalHarmonicsCount,
                             (const float4*) g.global.position, g.global.sphericalHarmonics, make_float3(cp[0], cp[1], cp[2]),
                             (float4*) g.global.color);
    gs_timer_stop(timer);
    reportLaunch("evaluateSphericalHarmonics");
}

extern "C" void gs_evaluate_splat_clip_data(GsTimer* timer, int32_t count)
{
    gs_timer_start(timer);
    launchSplatClipData(g.stream, count, currentScreen(), currentCamera(), (const float4*) g.global.position,
                        (const float4*) g.global.scaleAndRotation, (float2*) g.global.positionClipSpaceXY,
                        g.global.positionClipSpaceZ, (float4*) g.global.screenEllipse, (float4*) g.global.conic, nullptr, nullptr);
    gs_timer_stop(timer);
    reportLaunch("evaluateSplatsClipData");
}

extern "C" int32_t gs_build_tile_list(GsTimer* timer, int32_t /*numBlocks*/, int32_t tileListCapacity)
{
    const Screen scr = currentScreen();
    const int count = g.global.splatCount;
    gs_timer_start(timer);
    launchTileListCount(g.stream, count, scr, (const float2*) g.global.positionClipSpaceXY, (const float4*) g.global.screenEllipse,
                        g.scratch);
    launchTileListEmit(g.stream, count, scr, (const float2*) g.global.positionClipSpaceXY, g.global.positionClipSpaceZ,
                       (const float4*) g.global.screenEllipse, g.scratch, g.tileList.keys, g.tileList.values, tileListCapacity);
    gs_timer_stop(timer);
    // The reference reads the pair counter back here too (GaussianRender.cu:793), a host sync point.
    GS_CHECK(hipStreamSynchronize(g.stream));
    reportLaunch("buildTileList");
    g.lastDemand = *g.scratch.totalHost;
    return (int32_t) std::min<int64_t>(g.lastDemand, (int64_t) tileListCapacity); // GaussianRender.cu:800-801
}

extern "C" int64_t gs_last_tile_list_demand(void) { return g.lastDemand; }

extern "C" void gs_sort_tile_list(GsTimer* timer, int32_t size, void** deviceTempStorage, size_t* tempStorageSizeInBytes,
                                  GsDoubleBufferU64* keys, GsDoubleBufferI32* values)
{
    gs_timer_start(timer);
    if (size > 0)
    {
        const int endBit = sortEndBit(currentScreen().totalTiles);
        const size_t required = sortTempBytes(8, size, 0, endBit);
        if (required > *tempStorageSizeInBytes) // GaussianRender.cu:831-839
        {
            if (*deviceTempStorage != nullptr) GS_CHECK(hipFree(*deviceTempStorage));
            GS_CHECK(hipMalloc(deviceTempStorage, required));
            *tempStorageSizeInBytes = required;
        }
        // Keys and values may start on different selectors only if the caller mixed them up; follow keys.
        const int sel = launchSortPairs(g.stream, *deviceTempStorage, *tempStorageSizeInBytes, 8, keys->buffers[0], keys->buffers[1],
                                        values->buffers[0], values->buffers[1], keys->selector, size, 0, endBit);
        keys->selector = sel;
        values->selector = sel;
    }
    gs_timer_stop(timer);
    if (hipGetLastError() != hipSuccess) printf("error sortTileList\n");
}

extern "C" void gs_evaluate_tile_range(GsTimer* timer, int32_t size)
{
    gs_timer_start(timer);
    launchTileRanges(g.stream, size, 8, g.tileList.keys, g.global.tileRange);
    gs_timer_stop(timer);
    reportLaunch("evaluateTileRanges");
}

extern "C" void gs_set_collect_raster_stats(int enabled)
{
    cfg.collectRasterStats = enabled != 0;
    if (enabled && cfg.consumedDevice == nullptr) GS_CHECK(hipMalloc((void**) &cfg.consumedDevice, sizeof(unsigned long long)));
}

extern "C" int64_t gs_last_raster_consumed_pairs(void)
{
    if (cfg.consumedDevice == nullptr) return 0;
    unsigned long long v = 0;
    GS_CHECK(hipStreamSynchronize(g.stream));
    GS_CHECK(hipMemcpy(&v, cfg.consumedDevice, sizeof(v), hipMemcpyDeviceToHost));
    return (int64_t) v;
}

extern "C" void gs_set_footprint(int footprint)
{
    g_footprint = footprint == GS_FOOTPRINT_EPANECHNIKOV ? GS_FOOTPRINT_EPANECHNIKOV : GS_FOOTPRINT_GAUSSIAN;
}
extern "C" int gs_get_footprint(void) { return g_footprint; }

extern "C" void gs_set_sort_backend(int backend)
{
    cfg.onesweep = backend == GS_SORT_BACKEND_ONESWEEP;
    cfg.sortBackend = (backend == GS_SORT_BACKEND_ROCPRIM) ? GS_SORT_BACKEND_ROCPRIM : GS_SORT_BACKEND_NATIVE;
}
extern "C" int gs_get_sort_backend(void) { return cfg.onesweep ? GS_SORT_BACKEND_ONESWEEP : cfg.sortBackend; }

extern "C" int gs_radix_sort_pairs(int keyBytes, void* keys0, void* keys1, int32_t* values0, int32_t* values1, int32_t count, int beginBit,
                                   int endBit)
{
    if (!radixSortSupported(keyBytes, count)) return -1;
    g.radix.onesweep = cfg.onesweep;
    return launchRadixSortPairs(g.stream, g.radix, keyBytes, keys0, keys1, values0, values1, count, beginBit, endBit);
}

extern "C" void gs_debug_set_raster_trace(uint64_t* deviceBufferOrNull) { setRasterTrace((unsigned long long*) deviceBufferOrNull); }
extern "C" void gs_set_raster_pixels_per_lane(int ppl) { setRasterPixelsPerLane(ppl); }
extern "C" int gs_get_raster_pixels_per_lane(void) { return rasterPixelsPerLane(); }

extern "C" void gs_rasterize_tile(GsTimer* timer)
{
    unsigned long long* consumed = cfg.collectRasterStats ? cfg.consumedDevice : nullptr;
    if (consumed) GS_CHECK(hipMemsetAsync(consumed, 0, sizeof(unsigned long long), g.stream));
    gs_timer_start(timer);
    launchRasterize(g.stream, currentScreen(), g.global.tileRange, g.tileList.values, (const float2*) g.global.positionClipSpaceXY,
                    (const float4*) g.global.conic, (const float4*) g.global.color, (uchar4*) g.global.backBuffer, consumed);
    gs_timer_stop(timer);
    reportLaunch("renderDepthBuffer");
}

// ---- one-call renderer (Demo.cpp:314-331 buffers, :352-479 frame) ---------------------------------
//
// Three pipelines produce the same frame:
//   GS_PIPELINE_REFERENCE  the reference's stage order through the eight staged calls above: pairs keyed by
//                          (tile << 32 | depth), one 64-bit radix sort over all pairs (44-47 key bits);
//   GS_PIPELINE_TWO_LEVEL  (default) sort the SPLATS by the 32-bit depth key first (N elements), emit the
//                          pairs in that order keyed by the tile index only, then one stable sort on the
//                          13-15 tile bits.  The resulting value list is identical (same total order, ties by
//                          splat index) while the radix-sort traffic drops from ~152 B to ~30 B per pair;
//   GS_PIPELINE_BINNED     the same depth sort, then every hit is written straight to its slot of the per-tile
//                          lists through LDS cursors (gs_bins.hip): no pair sort, no range search.
// The last two leave the SH pass out when the raster kernel can evaluate colours on demand (useLazyColours below),
// keep centre + ellipse in one 32-byte record per splat, and reset the range table from inside the clip kernel.
namespace
{
constexpr int kStageTimers = 7; // sh, clip, tile list, pair sort, ranges, raster, depth sort (added to the sort total)
// What begin() leaves for finish().
struct FrameState
{
    Screen scr{};
    GsTimer* stageTimers[kStageTimers] = {};
    bool blocking = false;
    bool listDone = false; // reference pipeline: the sorted list and ranges were produced in begin()
    int orderSel = 0, tileKeyBytes = 2;
    bool lazyColours = false; // colours are evaluated by the raster kernel, entry by entry (no SH pass this frame)
    float cameraPosition[3] = {0.f, 0.f, 0.f};
    bool binned = false; // binned pipeline: lists and ranges are on their way, only the pair total is awaited
    int binBatches = 0;
    const int* binCountDevice = nullptr;
    int32_t size = 0;
    GsFrameStats st{};
};
} // namespace

struct GsRenderer
{
    int width = 0, height = 0, rowBegin = 0, rowEnd = 0;
    int pipeline = GS_PIPELINE_TWO_LEVEL;
    int32_t splatCount = 0, shDegree = 0, shCount = 0;
    float *position = nullptr, *scaleAndRotation = nullptr, *color = nullptr, *sh = nullptr;
    float *clipXY = nullptr, *clipZ = nullptr, *conic = nullptr, *ellipse = nullptr;
    bool colourStale = false;   // the last frame evaluated its colours inside the raster kernel: r->color is not current
    float lastCameraPosition[3] = {0.f, 0.f, 0.f};
    float* tileRecord = nullptr; // 8 floats per splat: clip x, y, cos, sin, extent x, y, 0, 0 (one memory sector)
    float* rasterRecord = nullptr; // 8 floats per splat: clip x, y, conic x, y, z, 0, 0, 0 (what the raster gathers per list entry)
    bool sceneShared = false; // position, scaleAndRotation, sh, shPerSplat belong to another renderer (gs_renderer_share_scene)
    uint32_t* depthKey[2] = {nullptr, nullptr};
    int32_t* order[2] = {nullptr, nullptr};
    // band cull (multi-GPU): per-splat coefficients in the [splat][shCount] layout for gathered SH evaluation,
    // one keep-mask and count per 64 splats, their scan, and the kept-splat count (device scalar)
    float* shPerSplat = nullptr;
    // per-renderer stream and scratch, so that several renderers can have frames in flight at once
    hipStream_t stream = nullptr;
    bool hasStream = false;
    TileListScratch scratch;
    RadixScratch radix;
    BinScratch bins;
    unsigned long long* consumed = nullptr;
    // captured frame (hipGraph): the whole two-level frame as one graph launch, list sizes kept on the device
    bool useGraph = false, graphValid = false, ownsStream = false;
    // TWO instances of the captured frame, each reading the camera from its own pinned slot: a frame can be launched while the
    // previous one of this renderer is still running (its slot is not touched), so the stream never runs dry between frames
    hipGraph_t graph[2] = {nullptr, nullptr};
    hipGraphExec_t graphExec[2] = {nullptr, nullptr};
    int graphSlot = 0; // instance of the NEXT launch
    int views = 1;     // cameras per captured chain (gs_renderer_set_views)
    FrameParams* paramsDevice = nullptr;
    FrameParams* paramsHost = nullptr; // pinned, two slots
    int graphCapacity = 0, graphSplats = 0, graphRowBegin = -1, graphRowEnd = -1, graphCull = -1, graphBinned = -1, graphFootprint = -1, graphLazy = -1, graphAblation = 0, graphSelector = 0, graphKeyBytes = 2;
    bool pending = false; // begin() issued, finish() not yet
    int capturedRetries = 0; // captured frames rendered again because their pair demand exceeded the list capacity
    FrameState frame;
    unsigned long long* bandMask = nullptr;
    int32_t *bandCounts = nullptr, *bandOffsets = nullptr, *bandSlotPrefix = nullptr;
    int64_t* bandTotal = nullptr;
    int bandCull = GS_BAND_CULL_AUTO;
    int32_t* tileRange = nullptr;
    uint8_t* backBufferOwned = nullptr;
    uint8_t* backBuffer = nullptr;
    int32_t tileListCapacity = 0;
    uint64_t* keys[2] = {nullptr, nullptr}; // 64-bit keys, or 16/32-bit tile indices in the same storage
    int32_t* values[2] = {nullptr, nullptr};
    int selector = 0;
    int keyBytes = 8;
    int32_t tileListSize = 0;
    void* sortTemp = nullptr;
    size_t sortTempBytes = 0;
    GsTimer* timer = nullptr;
    hipEvent_t frameStart = nullptr, frameStop = nullptr, countReady = nullptr;
    // timed == 2: non-blocking per-stage event pairs, kStageTimers per frame, read by gs_renderer_collect_stage_times
    std::vector<GsTimer*> asyncTimers;
    size_t asyncUsed = 0;
    int64_t asyncPairs = 0;
    int asyncFrames = 0;
};

namespace
{
// The launchers enqueue on g.stream; a renderer that was given its own stream borrows the slot for the call, so that
// several renderers (frames in flight) can be driven from one host thread on different streams.
struct StreamScope
{
    hipStream_t saved;
    explicit StreamScope(GsRenderer* r) : saved(g.stream)
    {
        if (r->hasStream) g.stream = r->stream;
    }
    ~StreamScope() { g.stream = saved; }
};
} // namespace


namespace
{
// Colours on demand: the two-level / binned frames skip the SH pass and let the raster kernel evaluate the colour of every
// list entry it stages ("Colours on demand").  Needs the [splat][shCount] coefficient copy and a degree whose
// rows are 16-byte multiples; GS_NO_LAZY_SH=1 keeps the reference's stage order.
bool useLazyColours(const GsRenderer* r)
{
    static const bool disabled = getenv("GS_NO_LAZY_SH") != nullptr;
    return !disabled && r->shDegree != 0 && r->shPerSplat != nullptr && r->pipeline != GS_PIPELINE_REFERENCE &&
           rasterCanEvaluateColours(r->shDegree, r->shCount);
}
RasterColourSource colourSource(const GsRenderer* r, const float* cameraPosition, const FrameParams* paramsDevice)
{
    RasterColourSource src;
    src.degree = r->shDegree;
    src.shCount = r->shCount;
    src.position = (const float4*) r->position;
    src.shPerSplat = r->shPerSplat;
    if (cameraPosition) src.cameraPosition = make_float3(cameraPosition[0], cameraPosition[1], cameraPosition[2]);
    src.params = paramsDevice;
    return src;
}

void freeTileList(GsRenderer* r)
{
    for (int i = 0; i != 2; ++i)
    {
        gs_device_free(r->keys[i]);
        gs_device_free(r->values[i]);
        r->keys[i] = nullptr;
        r->values[i] = nullptr;
    }
}
void allocTileList(GsRenderer* r, int32_t capacity)
{
    freeTileList(r);
    for (int i = 0; i != 2; ++i)
    {
        r->keys[i] = (uint64_t*) gs_device_alloc(sizeof(uint64_t) * (size_t) capacity);
        r->values[i] = (int32_t*) gs_device_alloc(sizeof(int32_t) * (size_t) capacity);
    }
    r->tileListCapacity = capacity;
}
void freeScene(GsRenderer* r)
{
    if (r->sceneShared)
    {
        // the read-only inputs are the owner's
        r->position = r->scaleAndRotation = r->sh = r->shPerSplat = nullptr;
        r->sceneShared = false;
    }
    float** bufs[] = {&r->position, &r->scaleAndRotation, &r->color,   &r->sh,         &r->clipXY,
                      &r->clipZ,    &r->conic,            &r->ellipse, &r->tileRecord, &r->rasterRecord};
    for (float** b : bufs)
    {
        gs_device_free(*b);
        *b = nullptr;
    }
    for (int i = 0; i != 2; ++i)
    {
        gs_device_free(r->depthKey[i]);
        gs_device_free(r->order[i]);
        r->depthKey[i] = nullptr;
        r->order[i] = nullptr;
    }
    gs_device_free(r->shPerSplat);
    gs_device_free(r->bandMask);
    gs_device_free(r->bandCounts);
    gs_device_free(r->bandSlotPrefix);
    gs_device_free(r->bandOffsets);
    gs_device_free(r->bandTotal);
    r->shPerSplat = nullptr;
    r->bandMask = nullptr;
    r->bandCounts = r->bandOffsets = r->bandSlotPrefix = nullptr;
    r->bandTotal = nullptr;
}
void ensureSortTemp(GsRenderer* r, size_t required)
{
    if (required > r->sortTempBytes)
    {
        gs_device_free(r->sortTemp);
        r->sortTemp = gs_device_alloc(required);
        r->sortTempBytes = required;
    }
}
bool growTileListIfNeeded(GsRenderer* r, int64_t demand)
{
    if (demand <= r->tileListCapacity) return false;
    int64_t cap = r->tileListCapacity > 0 ? r->tileListCapacity : 1;
    while (cap < demand) cap <<= 1; // Demo.cpp:360
    cap = std::min<int64_t>(cap, 0x7FFFFFFF);
    allocTileList(r, (int32_t) cap);
    return true;
}
} // namespace

extern "C" GsRenderer* gs_renderer_create(int width, int height)
{
    GsRenderer* r = new GsRenderer{};
    const Screen scr = makeScreen(width, height, 0, 0);
    r->width = scr.width;
    r->height = scr.height;
    r->tileRange = (int32_t*) gs_device_alloc(sizeof(int32_t) * 2 * (size_t) scr.totalTiles);
    GS_CHECK(hipMemset(r->tileRange, 0xFF, sizeof(int32_t) * 2 * (size_t) scr.totalTiles)); // binned frames rewrite their band only
    r->backBufferOwned = (uint8_t*) gs_device_alloc((size_t) scr.width * scr.height * 4);
    r->backBuffer = r->backBufferOwned;
    GS_CHECK(hipMemset(r->backBufferOwned, 0, (size_t) scr.width * scr.height * 4)); // rows outside a band stay zero
    r->timer = gs_timer_create();
    GS_CHECK(hipEventCreate(&r->frameStart));
    GS_CHECK(hipEventCreate(&r->frameStop));
    GS_CHECK(hipEventCreateWithFlags(&r->countReady, hipEventDisableTiming));
    return r;
}

extern "C" void gs_renderer_destroy(GsRenderer* r)
{
    if (!r) return;
    {
        StreamScope scope(r);
        GS_CHECK(hipStreamSynchronize(g.stream));
    }
    freeScene(r);
    freeTileList(r);
    gs_device_free(r->tileRange);
    gs_device_free(r->backBufferOwned);
    gs_device_free(r->sortTemp);
    freeTileListScratch(r->scratch);
    freeRadixScratch(r->radix);
    freeBinScratch(r->bins);
    for (int k = 0; k != 2; ++k)
    {
        if (r->graphExec[k]) (void) hipGraphExecDestroy(r->graphExec[k]);
        if (r->graph[k]) (void) hipGraphDestroy(r->graph[k]);
    }
    if (r->ownsStream) (void) hipStreamDestroy(r->stream);
    if (r->paramsDevice) (void) hipFree(r->paramsDevice);
    if (r->paramsHost) (void) hipHostFree(r->paramsHost);
    if (r->consumed) (void) hipFree(r->consumed);
    gs_timer_destroy(r->timer);
    for (GsTimer* t : r->asyncTimers) gs_timer_destroy(t);
    (void) hipEventDestroy(r->frameStart);
    (void) hipEventDestroy(r->frameStop);
    (void) hipEventDestroy(r->countReady);
    delete r;
}

extern "C" void gs_renderer_set_tile_rows(GsRenderer* r, int begin, int end)
{
    if (begin != r->rowBegin || end != r->rowEnd)
    {
        // ranges of tiles that leave the band must not linger (the binned pipeline rewrites its own rows only)
        StreamScope scope(r);
        GS_CHECK(hipStreamSynchronize(g.stream));
        const Screen scr = makeScreen(r->width, r->height, 0, 0);
        GS_CHECK(hipMemset(r->tileRange, 0xFF, sizeof(int32_t) * 2 * (size_t) scr.totalTiles));
    }
    r->rowBegin = begin;
    r->rowEnd = end;
}

extern "C" void gs_renderer_set_pipeline(GsRenderer* r, int pipeline)
{
    r->pipeline = (pipeline == GS_PIPELINE_REFERENCE || pipeline == GS_PIPELINE_BINNED) ? pipeline : GS_PIPELINE_TWO_LEVEL;
    r->graphValid = false;
}

extern "C" int gs_renderer_colours_on_demand(const GsRenderer* r) { return useLazyColours(r) ? 1 : 0; }

extern "C" void gs_renderer_set_band_cull(GsRenderer* r, int mode)
{
    r->bandCull = (mode == GS_BAND_CULL_ON || mode == GS_BAND_CULL_OFF) ? mode : GS_BAND_CULL_AUTO;
}

namespace
{
// Everything a renderer owns per scene besides the read-only inputs: per-splat intermediates and the list buffers.
void allocateSceneIntermediates(GsRenderer* r, int32_t n)
{
    const size_t f4 = sizeof(float) * 4 * (size_t) n;
    r->color = (float*) gs_device_alloc(f4);
    r->conic = (float*) gs_device_alloc(f4);
    r->ellipse = (float*) gs_device_alloc(f4);
    r->clipXY = (float*) gs_device_alloc(sizeof(float) * 2 * (size_t) n);
    // per-view results (clip data of the "virtual splats" view * n + i, their depth keys and order, the band-cull masks): x views
    const size_t nv = (size_t) n * (size_t) r->views;
    r->clipZ = (float*) gs_device_alloc(sizeof(float) * nv);
    r->tileRecord = (float*) gs_device_alloc(sizeof(float) * 8 * nv);
    r->rasterRecord = (float*) gs_device_alloc(sizeof(float) * 8 * nv);
    r->scratch.tileRecord = (const float4*) r->tileRecord;
    r->bins.tileRecord = (const float4*) r->tileRecord;
    for (int i = 0; i != 2; ++i)
    {
        r->depthKey[i] = (uint32_t*) gs_device_alloc(sizeof(uint32_t) * nv);
        r->order[i] = (int32_t*) gs_device_alloc(sizeof(int32_t) * nv);
    }
    const size_t waves = (((size_t) n + kWave - 1) / kWave) * (size_t) r->views;
    r->bandMask = (unsigned long long*) gs_device_alloc(sizeof(unsigned long long) * waves);
    r->bandCounts = (int32_t*) gs_device_alloc(sizeof(int32_t) * waves);
    r->bandSlotPrefix = (int32_t*) gs_device_alloc(sizeof(int32_t) * waves);
    r->bandOffsets = (int32_t*) gs_device_alloc(sizeof(int32_t) * waves);
    r->bandTotal = (int64_t*) gs_device_alloc(sizeof(int64_t));
    // Demo.cpp:325: 8 pairs per splat to start with; grown on demand.
#ifndef GS_INITIAL_LIST_FACTOR
#define GS_INITIAL_LIST_FACTOR 8
#endif
    const int64_t initial = std::min<int64_t>((int64_t) std::max(n, 1) * GS_INITIAL_LIST_FACTOR * r->views, 0x7FFFFFFF);
    if (r->tileListCapacity < initial) allocTileList(r, (int32_t) initial);
}
} // namespace

extern "C" int gs_renderer_set_scene(GsRenderer* r, int32_t n, const float* position4, const float* scaleAndRotation4,
                                     const float* color4, const float* sh, int shDegree, int shCount)
{
    if (n < 0 || (shDegree != 0 && sh == nullptr)) return -1;
    StreamScope scope(r);
    GS_CHECK(hipStreamSynchronize(g.stream));
    freeScene(r);
    r->graphValid = false;
    r->splatCount = n;
    r->shDegree = shDegree;
    r->shCount = shCount;
    const size_t f4 = sizeof(float) * 4 * (size_t) n;
    r->position = (float*) gs_device_alloc(f4);
    r->scaleAndRotation = (float*) gs_device_alloc(f4);
    allocateSceneIntermediates(r, n);
    gs_device_upload(r->position, position4, f4);
    gs_device_upload(r->scaleAndRotation, scaleAndRotation4, f4);
    gs_device_upload(r->color, color4, f4);
    if (shDegree != 0)
    {
        // Demo.cpp:291-296
        const size_t total = gs_realigned_sh_size(GS_SH_GROUP_SIZE, shCount, n);
        std::vector<float> aligned(total);
        gs_realign_spherical_harmonics(sh, aligned.data(), GS_SH_GROUP_SIZE, shCount, n);
        r->sh = (float*) gs_device_alloc(total * sizeof(float));
        gs_device_upload(r->sh, aligned.data(), total * sizeof(float));
        // second copy in the file's [splat][shCount] layout for the band-culled (gathered) SH evaluation
        r->shPerSplat = (float*) gs_device_alloc(sizeof(float) * (size_t) n * shCount);
        gs_device_upload(r->shPerSplat, sh, sizeof(float) * (size_t) n * shCount);
    }
    return 0;
}

// Several renderers of one scene (frames in flight): the read-only inputs -- position, scale/rotation, both coefficient layouts --
// exist once, in the owner; the sharer allocates only its own intermediates and lists (0.3 instead of 0.75 GB at C3) and copies the
// static colours device to device.  The owner must outlive its sharers and keep its scene.
extern "C" int gs_renderer_share_scene(GsRenderer* r, const GsRenderer* owner)
{
    if (r == nullptr || owner == nullptr || r == owner || owner->sceneShared) return -1;
    StreamScope scope(r);
    GS_CHECK(hipStreamSynchronize(g.stream));
    freeScene(r);
    r->graphValid = false;
    const int32_t n = owner->splatCount;
    r->splatCount = n;
    r->shDegree = owner->shDegree;
    r->shCount = owner->shCount;
    r->position = owner->position;
    r->scaleAndRotation = owner->scaleAndRotation;
    r->sh = owner->sh;
    r->shPerSplat = owner->shPerSplat;
    r->sceneShared = true;
    allocateSceneIntermediates(r, n);
    if (n > 0) GS_CHECK(hipMemcpy(r->color, owner->color, sizeof(float) * 4 * (size_t) n, hipMemcpyDeviceToDevice));
    return 0;
}

extern "C" uint8_t* gs_renderer_back_buffer(GsRenderer* r) { return r->backBuffer; }
extern "C" uint8_t* gs_renderer_back_buffer_view(GsRenderer* r, int view)
{
    if (view < 0 || view >= r->views) return nullptr;
    return r->backBuffer + (size_t) view * r->width * r->height * 4;
}
extern "C" int gs_renderer_views(const GsRenderer* r) { return r->views; }

// Several cameras per chain of kernels (header: gs_renderer_set_views).  Before the scene is set: the per-view intermediates are
// sized there.
extern "C" int gs_renderer_set_views(GsRenderer* r, int views)
{
    if (views < 1 || views > kMaxViews || r->position != nullptr) return -1;
    if (views == r->views) return 0;
    StreamScope scope(r);
    GS_CHECK(hipStreamSynchronize(g.stream));
    const Screen scr = makeScreen(r->width, r->height, 0, 0);
    const bool external = r->backBuffer != r->backBufferOwned;
    gs_device_free(r->tileRange);
    gs_device_free(r->backBufferOwned);
    r->views = views;
    r->tileRange = (int32_t*) gs_device_alloc(sizeof(int32_t) * 2 * (size_t) scr.totalTiles * views);
    GS_CHECK(hipMemset(r->tileRange, 0xFF, sizeof(int32_t) * 2 * (size_t) scr.totalTiles * views));
    r->backBufferOwned = (uint8_t*) gs_device_alloc((size_t) scr.width * scr.height * 4 * views);
    GS_CHECK(hipMemset(r->backBufferOwned, 0, (size_t) scr.width * scr.height * 4 * views)); // rows outside a band stay zero
    if (!external) r->backBuffer = r->backBufferOwned;
    r->graphValid = false;
    return 0;
}
extern "C" void gs_renderer_set_back_buffer(GsRenderer* r, uint8_t* p) { r->backBuffer = p ? p : r->backBufferOwned; }

extern "C" void gs_renderer_read_back_buffer(GsRenderer* r, uint8_t* hostRgba)
{
    StreamScope scope(r);
    gs_device_download(hostRgba, r->backBuffer, (size_t) r->width * r->height * 4);
}
extern "C" int gs_renderer_read_back_buffer_view(GsRenderer* r, int view, uint8_t* hostRgba)
{
    if (view < 0 || view >= r->views) return -1;
    StreamScope scope(r);
    gs_device_download(hostRgba, r->backBuffer + (size_t) view * r->width * r->height * 4, (size_t) r->width * r->height * 4);
    return 0;
}

extern "C" void gs_renderer_buffers(GsRenderer* r, GsRendererBuffers* out)
{
    out->position = r->position;
    out->scaleAndRotation = r->scaleAndRotation;
    out->color = r->color;
    out->sphericalHarmonics = r->sh;
    out->positionClipSpaceXY = r->clipXY;
    out->positionClipSpaceZ = r->clipZ;
    out->conic = r->conic;
    out->screenEllipse = r->ellipse;
    if (r->tileRecord != nullptr && r->pipeline != GS_PIPELINE_REFERENCE && r->splatCount > 0)
    {
        // the two-level / binned frames keep ellipses, centres and conics only inside the packed records: unpack them for the caller
        StreamScope scope(r);
        launchUnpackEllipse(g.stream, r->splatCount, (const float4*) r->tileRecord, (float4*) r->ellipse);
        launchUnpackRasterRecord(g.stream, r->splatCount, (const float4*) r->rasterRecord, (float2*) r->clipXY, r->clipZ, (float4*) r->conic);
        GS_CHECK(hipStreamSynchronize(g.stream));
    }
    if (r->colourStale && r->shDegree != 0 && r->splatCount > 0)
    {
        // the last frame evaluated colours inside the raster kernel: run the SH pass once for the caller
        StreamScope scope(r);
        launchSphericalHarmonics(g.stream, r->splatCount, r->shDegree, r->shCount, (const float4*) r->position, r->sh,
                                 make_float3(r->lastCameraPosition[0], r->lastCameraPosition[1], r->lastCameraPosition[2]), (float4*) r->color);
        GS_CHECK(hipStreamSynchronize(g.stream));
        r->colourStale = false;
    }
    out->tileRange = r->tileRange;
    out->keys = r->keys[r->selector];
    out->values = r->values[r->selector];
    if (r->useGraph && r->graphValid && r->scratch.totalHost != nullptr)
    {
        // captured frames keep the pair count on the device; fetch the last frame's for the caller
        StreamScope scope(r);
        GS_CHECK(hipStreamSynchronize(g.stream));
        r->tileListSize = (int32_t) std::min<int64_t>(*r->scratch.totalHost, r->tileListCapacity);
        out->keys = r->keys[r->selector];
        out->values = r->values[r->selector];
    }
    out->splatCount = r->splatCount;
    out->tileListSize = r->tileListSize;
    out->keyBytes = r->keyBytes;
}

extern "C" void gs_renderer_set_stream(GsRenderer* r, void* stream)
{
    r->stream = (hipStream_t) stream;
    r->hasStream = true;
}

// First half of a frame: clears, SH, clip data, (band cull,) depth sort of the splats, pair count + scan, and the
// emit pass launched with the current list capacity. Returns without waiting for anything.
extern "C" int gs_renderer_begin(GsRenderer* r, const GsCameraData* camera, int timed)
{
    if (r->pending) return -3;
    StreamScope scope(r);
    FrameState& f = r->frame;
    f = FrameState{};
    f.scr = makeScreen(r->width, r->height, r->rowBegin, r->rowEnd);
    const Screen& scr = f.scr;
    // timed: 0 = no stage timers; 1 = the reference's blocking timer after every stage; 2 = one non-blocking
    // event pair per stage, accumulated and read later with gs_renderer_collect_stage_times.
    GsTimer** stageTimers = f.stageTimers;
    if (timed == 1)
        for (int i = 0; i != kStageTimers; ++i) stageTimers[i] = r->timer;
    else if (timed == 2)
    {
        while (r->asyncTimers.size() < r->asyncUsed + kStageTimers)
        {
            GsTimer* t = gs_timer_create();
            gs_timer_set_blocking(t, 0);
            r->asyncTimers.push_back(t);
        }
        for (int i = 0; i != kStageTimers; ++i) stageTimers[i] = r->asyncTimers[r->asyncUsed + i];
        // stages that do not run this frame must still hold a valid (zero-length) interval
        for (int i = 0; i != kStageTimers; ++i)
        {
            gs_timer_start(stageTimers[i]);
            gs_timer_stop(stageTimers[i]);
        }
        r->asyncUsed += kStageTimers;
        r->asyncFrames += 1;
    }
    f.blocking = timed == 1;
    const bool blocking = f.blocking;
    GsFrameStats& st = f.st;
    const int n = r->splatCount;
    // The binned pipeline shares the front half (clip data, band cull, depth sort) with the two-level one; it needs the
    // native sort (device-side counts) and a band whose cursor table fits the LDS, else the frame runs two-level.
    const bool binned = r->pipeline == GS_PIPELINE_BINNED && n > 0 && cfg.sortBackend == GS_SORT_BACKEND_NATIVE && binsSupported(scr) &&
                        radixSortSupported(4, n);
    const bool twoLevel = r->pipeline != GS_PIPELINE_REFERENCE;

    GS_CHECK(hipEventRecord(r->frameStart, g.stream));

    // Clears (Demo.cpp:399,403). Only this GPU's band of the back buffer; the whole range table.
    const size_t rowBytes = (size_t) r->width * 4;
    const int pixelRowBegin = scr.rowBegin * kTileSize;
    const int pixelRowEnd = std::min(scr.rowEnd * kTileSize, r->height);
    // (two-level pipeline: the raster kernel zeroes the tiles that have no work itself, no 8 MB memset)
    if (pixelRowEnd > pixelRowBegin && !twoLevel)
        gs_device_memset(r->backBuffer + rowBytes * pixelRowBegin, 0, rowBytes * (size_t) (pixelRowEnd - pixelRowBegin));
    // (two-level pipeline: the clip kernel resets the range table on the side; binned: every range of the band is written)
    if (!twoLevel || n <= 0) gs_device_memset(r->tileRange, 0xFF, sizeof(int32_t) * 2 * (size_t) scr.totalTiles);

    // Demo.cpp:406-430
    GsGlobalArgs args{};
    args.cameraData = *camera;
    args.position = r->position;
    args.scaleAndRotation = r->scaleAndRotation;
    args.color = r->color;
    args.sphericalHarmonicsDegree = r->shDegree;
    args.sphericalHarmonicsCount = r->shCount;
    args.sphericalHarmonics = r->sh;
    args.positionClipSpaceXY = r->clipXY;
    args.positionClipSpaceZ = r->clipZ;
    args.conic = r->conic;
    args.screenEllipse = r->ellipse;
    args.splatCount = n;
    args.backBuffer = r->backBuffer;
    args.tileRange = r->tileRange;
    args.width = r->width;
    args.height = r->height;
    args.tileRowBegin = scr.rowBegin;
    args.tileRowEnd = scr.rowEnd;
    gs_set_global_args(&args);

    // Band cull: with a strict sub-band of tile rows (one rank of a multi-GPU split) only the splats whose tile AABB
    // reaches the band go through SH evaluation, the depth sort and the tile-list passes.
    const bool strictBand = scr.rowBegin > 0 || scr.rowEnd < scr.tilesY;
    const bool cull = twoLevel && n > 0 && cfg.sortBackend == GS_SORT_BACKEND_NATIVE &&
                      (r->bandCull == GS_BAND_CULL_ON || (r->bandCull == GS_BAND_CULL_AUTO && strictBand));

    f.lazyColours = useLazyColours(r);
    for (int k = 0; k != 3; ++k) f.cameraPosition[k] = r->lastCameraPosition[k] = camera->position[k];
    if (r->shDegree != 0 && !cull && !f.lazyColours) // Demo.cpp:432
    {
        gs_evaluate_spherical_harmonics(stageTimers[0], n);
        if (blocking) st.msSphericalHarmonics = gs_timer_elapsed_ms(stageTimers[0]);
    }

    if (!twoLevel)
    {
        // ---- the reference's stage order, through the staged calls (whole frame here; finish() only reports) ----
        gs_evaluate_splat_clip_data(stageTimers[1], n);
        if (blocking) st.msClipData = gs_timer_elapsed_ms(stageTimers[1]);

        // Count first and grow if needed, instead of rendering a truncated frame and doubling for the next
        // one (Demo.cpp:357-366,447).
        gs_timer_start(stageTimers[2]);
        launchTileListCount(g.stream, n, scr, (const float2*) r->clipXY, (const float4*) r->ellipse, r->scratch);
        GS_CHECK(hipStreamSynchronize(g.stream)); // pair count readback, as GaussianRender.cu:793
        const int64_t demand = *r->scratch.totalHost;
        g.lastDemand = demand;
        if (demand > 0x7FFFFFFF) return -2;
        if (growTileListIfNeeded(r, demand)) st.grewTileList = 1;
        const int32_t size = (int32_t) demand;
        launchTileListEmit(g.stream, n, scr, (const float2*) r->clipXY, r->clipZ, (const float4*) r->ellipse, r->scratch, r->keys[0],
                           r->values[0], r->tileListCapacity);
        gs_timer_stop(stageTimers[2]);
        if (blocking) st.msBuildTileList = gs_timer_elapsed_ms(stageTimers[2]);

        GsDoubleBufferU64 keys{{r->keys[0], r->keys[1]}, 0};
        GsDoubleBufferI32 values{{r->values[0], r->values[1]}, 0};
        gs_sort_tile_list(stageTimers[3], size, &r->sortTemp, &r->sortTempBytes, &keys, &values); // Demo.cpp:458
        if (blocking) st.msSortTileList = gs_timer_elapsed_ms(stageTimers[3]);
        r->selector = keys.selector;
        r->keyBytes = 8;

        GsTileListArgs tl{keys.buffers[keys.selector], values.buffers[values.selector], size, r->tileListCapacity};
        gs_set_tile_list_args(&tl); // Demo.cpp:463-466
        if (size != 0)              // Demo.cpp:469
        {
            gs_evaluate_tile_range(stageTimers[4], size);
            if (blocking) st.msTileRanges = gs_timer_elapsed_ms(stageTimers[4]);
        }
        f.size = size;
        f.listDone = true;
    }
    else
    {
        // ---- two-level sort ------------------------------------------------------------------------
        f.tileKeyBytes = scr.totalTiles > 65536 ? 4 : 2;

        gs_timer_start(stageTimers[1]);
        const int* countDevice = nullptr;
        if (!cull)
        {
            launchSplatClipData(g.stream, n, scr, currentCamera(), (const float4*) r->position, (const float4*) r->scaleAndRotation,
                                binned ? (float2*) r->clipXY : nullptr, nullptr, nullptr, nullptr, r->depthKey[0],
                                (uint32_t*) r->order[0], nullptr, nullptr, nullptr, nullptr, binned ? nullptr : r->tileRange, 2 * scr.totalTiles,
                                (float4*) r->tileRecord, (float4*) r->rasterRecord);
        }
        else
        {
            // clip data for every splat + keep-mask, then an order-preserving compaction of (depth key, index)
            launchSplatClipData(g.stream, n, scr, currentCamera(), (const float4*) r->position, (const float4*) r->scaleAndRotation,
                                binned ? (float2*) r->clipXY : nullptr, r->clipZ, nullptr, nullptr, nullptr, nullptr, r->bandMask,
                                r->bandSlotPrefix, r->bandCounts, nullptr, binned ? nullptr : r->tileRange, 2 * scr.totalTiles, (float4*) r->tileRecord,
                                (float4*) r->rasterRecord);
            launchBandCompaction(g.stream, n, 1, r->bandMask, r->bandSlotPrefix, r->bandCounts, r->bandOffsets, r->bandTotal, r->clipZ, r->depthKey[0],
                                 (uint32_t*) r->order[0]);
            countDevice = (const int*) r->bandTotal; // low 32 bits of the little-endian int64 total
        }
        gs_timer_stop(stageTimers[1]);
        if (blocking) st.msClipData = gs_timer_elapsed_ms(stageTimers[1]);
        if (cull && r->shDegree != 0 && !f.lazyColours)
        {
            // SH colour of the kept splats only (list = the compacted, still index-ordered order[0])
            gs_timer_start(stageTimers[0]);
            const float* cp = camera->position;
            launchSphericalHarmonicsGather(g.stream, n, countDevice, r->shDegree, r->shCount, r->order[0], (const float4*) r->position,
                                           r->shPerSplat, make_float3(cp[0], cp[1], cp[2]), (float4*) r->color);
            gs_timer_stop(stageTimers[0]);
            if (blocking) st.msSphericalHarmonics = gs_timer_elapsed_ms(stageTimers[0]);
        }
        r->radix.countDevice = countDevice;
        r->radix.onesweep = cfg.onesweep;
        r->scratch.countDevice = countDevice;

        // (1) splats by depth: 32-bit keys, N elements
        int orderSel = 0;
        gs_timer_start(stageTimers[6]);
        const bool native = cfg.sortBackend == GS_SORT_BACKEND_NATIVE;
        if (n > 0)
        {
            if (native && radixSortSupported(4, n))
                orderSel = launchRadixSortPairs(g.stream, r->radix, 4, r->depthKey[0], r->depthKey[1], r->order[0], r->order[1], n, 0, kDepthKeyBits);
            else
            {
                ensureSortTemp(r, sortTempBytes(4, n, 0, kDepthKeyBits));
                orderSel = launchSortPairs(g.stream, r->sortTemp, r->sortTempBytes, 4, r->depthKey[0], r->depthKey[1], r->order[0],
                                           r->order[1], 0, n, 0, kDepthKeyBits);
            }
        }
        gs_timer_stop(stageTimers[6]);
        f.orderSel = orderSel;
        const uint32_t* order = (const uint32_t*) r->order[orderSel];

        // (2) pairs in depth order, keyed by tile. The emit pass is launched with the current capacity while the
        // pair count travels to the host, so the readback (GaussianRender.cu:793) costs no GPU idle time.
        gs_timer_start(stageTimers[2]);
        if (binned)
        {
            // (2b) straight into the final per-tile lists: count per (row of sorted splats, tile), column scan, emit
            const int bandTiles = (scr.rowEnd - scr.rowBegin) * scr.tilesX;
            f.binned = true;
            f.binBatches = binBatchesFor(bandTiles);
            f.binCountDevice = countDevice;
            ensureTileListScratch(r->scratch, n); // the pair total and its host mirror live there
            ensureBinScratch(r->bins, n, bandTiles, f.binBatches);
            launchBinCountScan(g.stream, n, countDevice, scr, order, (const float2*) r->clipXY, (const float4*) r->ellipse, r->bins,
                               f.binBatches);
            launchBinEmit(g.stream, n, countDevice, scr, order, (const float2*) r->clipXY, (const float4*) r->ellipse, r->bins, f.binBatches,
                          r->values[0], r->tileListCapacity, r->tileRange, r->scratch.total, r->scratch.totalHostMapped);
            if (r->scratch.totalHostMapped == nullptr)
                GS_CHECK(hipMemcpyAsync(r->scratch.totalHost, r->scratch.total, sizeof(int64_t), hipMemcpyDeviceToHost, g.stream));
            GS_CHECK(hipEventRecord(r->countReady, g.stream));
        }
        else
        {
            launchTileListFastCount(g.stream, n, scr, order, (const float2*) r->clipXY, (const float4*) r->ellipse, r->scratch);
            launchTileListFastEmit(g.stream, n, scr, order, (const float2*) r->clipXY, (const float4*) r->ellipse, r->scratch, r->keys[0],
                                   f.tileKeyBytes, r->values[0], r->tileListCapacity);
            GS_CHECK(hipEventRecord(r->countReady, g.stream)); // the emit pass leaves the pair total
        }
        gs_timer_stop(stageTimers[2]);
    }
    r->pending = true;
    return 0;
}

// Second half: wait for the pair count (long since on the host), grow and re-emit in the rare overflow case, tile
// sort, ranges, raster.
extern "C" int gs_renderer_finish(GsRenderer* r, GsFrameStats* stats)
{
    if (!r->pending) return -3;
    StreamScope scope(r);
    FrameState& f = r->frame;
    const Screen& scr = f.scr;
    GsTimer** stageTimers = f.stageTimers;
    const bool blocking = f.blocking;
    GsFrameStats& st = f.st;
    const int n = r->splatCount;
    r->pending = false;

    if (f.binned)
    {
        const uint32_t* order = (const uint32_t*) r->order[f.orderSel];
        GS_CHECK(hipEventSynchronize(r->countReady));
        const int64_t demand = *r->scratch.totalHost;
        g.lastDemand = demand;
        if (demand > 0x7FFFFFFF) return -2;
        if (growTileListIfNeeded(r, demand))
        {
            // the lists did not fit: emit again into the larger buffer (counts and offsets are still valid)
            st.grewTileList = 1;
            launchBinEmit(g.stream, n, f.binCountDevice, scr, order, (const float2*) r->clipXY, (const float4*) r->ellipse, r->bins,
                          f.binBatches, r->values[0], r->tileListCapacity, r->tileRange, r->scratch.total, r->scratch.totalHostMapped);
        }
        if (blocking)
        {
            st.msBuildTileList = gs_timer_elapsed_ms(stageTimers[2]);
            st.msSortTileList = gs_timer_elapsed_ms(stageTimers[6]);
        }
        r->radix.countDevice = nullptr;
        r->scratch.countDevice = nullptr;
        r->selector = 0;
        r->keyBytes = 0; // no keys: the lists are addressed through tileRange only
        f.size = (int32_t) demand;
    }
    else if (!f.listDone)
    {
        const int tBits = tileBits(scr.totalTiles);
        const uint32_t* order = (const uint32_t*) r->order[f.orderSel];
        GS_CHECK(hipEventSynchronize(r->countReady));
        const int64_t demand = *r->scratch.totalHost;
        g.lastDemand = demand;
        if (demand > 0x7FFFFFFF) return -2;
        if (growTileListIfNeeded(r, demand))
        {
            // the list did not fit: re-emit into the larger buffers (rare; the first frames of a scene)
            st.grewTileList = 1;
            launchTileListFastEmit(g.stream, n, scr, order, (const float2*) r->clipXY, (const float4*) r->ellipse, r->scratch,
                                   r->keys[0], f.tileKeyBytes, r->values[0], r->tileListCapacity);
        }
        if (blocking) st.msBuildTileList = gs_timer_elapsed_ms(stageTimers[2]);
        r->radix.countDevice = nullptr;
        r->scratch.countDevice = nullptr;
        const int32_t size = (int32_t) demand;

        // (3) stable sort of the pairs on the tile bits only
        const bool native = cfg.sortBackend == GS_SORT_BACKEND_NATIVE;
        gs_timer_start(stageTimers[3]);
        int sel = 0;
        if (size > 0)
        {
            if (native && radixSortSupported(f.tileKeyBytes, size))
                sel = launchRadixSortPairs(g.stream, r->radix, f.tileKeyBytes, r->keys[0], r->keys[1], r->values[0], r->values[1], size, 0,
                                           tBits);
            else
            {
                ensureSortTemp(r, sortTempBytes(f.tileKeyBytes, size, 0, tBits));
                sel = launchSortPairs(g.stream, r->sortTemp, r->sortTempBytes, f.tileKeyBytes, r->keys[0], r->keys[1], r->values[0],
                                      r->values[1], 0, size, 0, tBits);
            }
        }
        gs_timer_stop(stageTimers[3]);
        if (blocking) st.msSortTileList = gs_timer_elapsed_ms(stageTimers[3]) + gs_timer_elapsed_ms(stageTimers[6]);
        r->selector = sel;
        r->keyBytes = f.tileKeyBytes;

        gs_timer_start(stageTimers[4]);
        launchTileRanges(g.stream, size, f.tileKeyBytes, r->keys[sel], r->tileRange);
        gs_timer_stop(stageTimers[4]);
        if (blocking) st.msTileRanges = gs_timer_elapsed_ms(stageTimers[4]);
        f.size = size;
    }
    const int32_t size = f.size;
    r->tileListSize = size;
    st.tileListSize = size;
    st.tileListCapacity = r->tileListCapacity;

    // raster (direct launch: no global argument state, so renderers can interleave their halves)
    unsigned long long* consumed = nullptr;
    if (cfg.collectRasterStats)
    {
        if (r->consumed == nullptr) GS_CHECK(hipMalloc((void**) &r->consumed, sizeof(unsigned long long)));
        consumed = r->consumed;
        GS_CHECK(hipMemsetAsync(consumed, 0, sizeof(unsigned long long), g.stream));
    }
    gs_timer_start(stageTimers[5]);
    const RasterColourSource lazySource = colourSource(r, f.cameraPosition, nullptr);
    launchRasterize(g.stream, scr, r->tileRange, r->values[r->selector], (const float2*) r->clipXY, (const float4*) r->conic,
                    (const float4*) r->color, (uchar4*) r->backBuffer, consumed, r->pipeline != GS_PIPELINE_REFERENCE,
                    f.lazyColours ? &lazySource : nullptr, r->pipeline != GS_PIPELINE_REFERENCE ? (const float4*) r->rasterRecord : nullptr);
    r->colourStale = f.lazyColours;
    gs_timer_stop(stageTimers[5]);
    if (hipGetLastError() != hipSuccess) printf("kernel error renderDepthBuffer\n");
    if (blocking) st.msRasterize = gs_timer_elapsed_ms(stageTimers[5]);

    GS_CHECK(hipEventRecord(r->frameStop, g.stream));
    r->asyncPairs += (r->asyncFrames > 0) ? size : 0;
    if (stats)
    {
        GS_CHECK(hipEventSynchronize(r->frameStop));
        GS_CHECK(hipEventElapsedTime(&st.msFrame, r->frameStart, r->frameStop));
        if (consumed != nullptr)
        {
            unsigned long long v = 0;
            GS_CHECK(hipMemcpy(&v, consumed, sizeof(v), hipMemcpyDeviceToHost));
            st.consumedPairs = (int64_t) v;
        }
        *stats = st;
    }
    return 0;
}

// ---- captured frame ------------------------------------------------------------------------------------------------
// The two-level frame with every list size left on the device (the pair count feeds the tile sort and the range kernel
// through pointers, grids are sized for the list capacity), so the frame has no host round trip and can be captured once
// and replayed as ONE graph launch per frame.  The camera travels in a small device struct that a memcpy node refreshes.
// The pair total reaches the host through a host-mapped word in the middle of the frame; a frame whose demand exceeds the
// list capacity is rendered again with a larger list before the call returns (never handed out truncated).
namespace
{
// Allocation / legacy-stream work of one host thread must not overlap another thread's stream capture: set-up and capture
// are serialised process-wide (they happen once per renderer; replays take no lock).
std::mutex g_captureMutex;
// host-side cost of captured frames (diagnostic, gs_debug_host_times): time inside hipGraphLaunch, time waiting for the pair total
std::atomic<long long> g_hostLaunchNs{0}, g_hostAwaitNs{0}, g_hostFrames{0};

int enqueueCapturableFrame(GsRenderer* r, const Screen& scr, bool cull, bool binned, bool lazy, int slot)
{
    const RasterColourSource lazySource = colourSource(r, nullptr, r->paramsDevice);
    const RasterColourSource* lazyColours = lazy ? &lazySource : nullptr;
    const int n = r->splatCount;
    // Several views per chain: V grid rows in the clip, compaction and raster kernels; everything in between works on the
    // n * V virtual splats and V * totalTiles virtual tiles as if they were one frame's (gs_internal.h: kMaxViews).
    const int views = r->views;
    const int nv = n * views;
    const int virtualTiles = scr.totalTiles * views;
    const int tileKeyBytes = virtualTiles > 65536 ? 4 : 2;
    const int tBits = tileBits(virtualTiles);
    const float3 unused = make_float3(0.f, 0.f, 0.f);
    ClipCamera unusedCamera{};

    GS_CHECK(hipMemcpyAsync(r->paramsDevice, r->paramsHost + slot * kMaxViews, sizeof(FrameParams) * views, hipMemcpyHostToDevice, g.stream));
    // the range table is reset to -1 by the clip kernel (one node less than a memset; see gs_ranges.hip); binned frames write every range
    int32_t* clearTable = binned ? nullptr : r->tileRange;
    const int clearEntries = 2 * virtualTiles;
#ifdef GS_DEBUG_MEMSET_NODE
    // audit build (tools/micro/graph_memset.hip): the table is cleared by a memset NODE instead
    if (clearTable != nullptr) GS_CHECK(hipMemsetAsync(clearTable, 0xFF, sizeof(int32_t) * (size_t) clearEntries, g.stream));
    clearTable = nullptr;
#endif
    if (r->shDegree != 0 && !cull && !lazy)
        launchSphericalHarmonics(g.stream, n, r->shDegree, r->shCount, (const float4*) r->position, r->sh, unused, (float4*) r->color,
                                 r->paramsDevice);
    const int* countDevice = nullptr;
    if (!cull)
    {
        if (!(g_ablation & 1))
            launchSplatClipData(g.stream, n, scr, unusedCamera, (const float4*) r->position, (const float4*) r->scaleAndRotation,
                                binned ? (float2*) r->clipXY : nullptr, nullptr, nullptr, nullptr, r->depthKey[0],
                                (uint32_t*) r->order[0], nullptr, nullptr, nullptr, r->paramsDevice, clearTable, clearEntries, (float4*) r->tileRecord,
                                (float4*) r->rasterRecord, views);
    }
    else
    {
        launchSplatClipData(g.stream, n, scr, unusedCamera, (const float4*) r->position, (const float4*) r->scaleAndRotation,
                            binned ? (float2*) r->clipXY : nullptr, r->clipZ, nullptr, nullptr, nullptr, nullptr, r->bandMask,
                            r->bandSlotPrefix, r->bandCounts, r->paramsDevice, clearTable, clearEntries, (float4*) r->tileRecord, (float4*) r->rasterRecord,
                            views);
        launchBandCompaction(g.stream, n, views, r->bandMask, r->bandSlotPrefix, r->bandCounts, r->bandOffsets, r->bandTotal, r->clipZ, r->depthKey[0],
                             (uint32_t*) r->order[0]);
        countDevice = (const int*) r->bandTotal;
        if (r->shDegree != 0 && !lazy)
            launchSphericalHarmonicsGather(g.stream, n, countDevice, r->shDegree, r->shCount, r->order[0], (const float4*) r->position,
                                           r->shPerSplat, unused, (float4*) r->color, r->paramsDevice);
    }
    r->radix.onesweep = false; // its ticket base / epoch are host-side launch arguments: not replayable
    r->radix.countDevice = countDevice;
    r->scratch.countDevice = countDevice;
    const int orderSel = (g_ablation & 2) ? 0 : launchRadixSortPairs(g.stream, r->radix, 4, r->depthKey[0], r->depthKey[1], r->order[0], r->order[1], nv, 0, kDepthKeyBits);
    const uint32_t* order = (const uint32_t*) r->order[orderSel];
    if (binned)
    {
        const int bandTiles = (scr.rowEnd - scr.rowBegin) * scr.tilesX;
        const int batches = binBatchesFor(bandTiles);
        launchBinCountScan(g.stream, n, countDevice, scr, order, (const float2*) r->clipXY, (const float4*) r->ellipse, r->bins, batches);
        launchBinEmit(g.stream, n, countDevice, scr, order, (const float2*) r->clipXY, (const float4*) r->ellipse, r->bins, batches,
                      r->values[0], r->tileListCapacity, r->tileRange, r->scratch.total, r->scratch.totalHostMapped);
        r->radix.countDevice = nullptr;
        r->scratch.countDevice = nullptr;
        launchRasterize(g.stream, scr, r->tileRange, r->values[0], (const float2*) r->clipXY, (const float4*) r->conic,
                        (const float4*) r->color, (uchar4*) r->backBuffer, nullptr, true, lazyColours, (const float4*) r->rasterRecord);
        r->graphSelector = 0;
        r->graphKeyBytes = 0;
        return 0;
    }
    if (!(g_ablation & 4)) launchTileListFastCount(g.stream, nv, scr, order, (const float2*) r->clipXY, (const float4*) r->ellipse, r->scratch);
    if (!(g_ablation & 8))
        launchTileListFastEmit(g.stream, nv, scr, order, (const float2*) r->clipXY, (const float4*) r->ellipse, r->scratch, r->keys[0],
                               tileKeyBytes, r->values[0], r->tileListCapacity, views > 1 ? n : 0);
    // tile sort + ranges on the device-side pair count (low word of the int64 total), grids sized for the capacity
    const int* pairCount = (const int*) r->scratch.total;
    r->radix.countDevice = pairCount;
    r->scratch.countDevice = nullptr;
    const int sel = (g_ablation & 16) ? 0
                                      : launchRadixSortPairs(g.stream, r->radix, tileKeyBytes, r->keys[0], r->keys[1], r->values[0], r->values[1],
                                                             r->tileListCapacity, 0, tBits);
    r->radix.countDevice = nullptr;
    if (!(g_ablation & 32)) launchTileRanges(g.stream, r->tileListCapacity, tileKeyBytes, r->keys[sel], r->tileRange, pairCount);
    if (!(g_ablation & 64))
        launchRasterize(g.stream, scr, r->tileRange, r->values[sel], (const float2*) r->clipXY, (const float4*) r->conic,
                        (const float4*) r->color, (uchar4*) r->backBuffer, nullptr, true, lazyColours, (const float4*) r->rasterRecord, views, n);
    r->graphSelector = sel;
    r->graphKeyBytes = tileKeyBytes;
    return 0;
}

// The pair total of a captured frame lands in a host-mapped word (written by the scan kernel, or by the binned emit pass).
// The host arms the word with -1 before the launch and waits for the frame's own value: a few microseconds of polling in
// the middle of the frame, long before its raster kernel ends.
int64_t awaitPairTotal(GsRenderer* r)
{
    volatile int64_t* word = r->scratch.totalHost;
    for (unsigned spins = 0;; ++spins)
    {
        const int64_t v = *word;
        if (v >= 0) return v;
        if ((spins & 63u) == 63u && hipStreamQuery(g.stream) == hipSuccess)
        {
            // the whole frame is done: every write of it is visible by now
            const int64_t w = *word;
            if (w >= 0) return w;
            int64_t fromDevice = 0; // (no host-mapped address: the total was left in device memory)
            GS_CHECK(hipMemcpy(&fromDevice, r->scratch.total, sizeof(int64_t), hipMemcpyDeviceToHost));
            return fromDevice;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

// `cameras`: r->views of them (one frame per view, rendered by one chain).
int renderCaptured(GsRenderer* r, const GsCameraData* cameras)
{
    if (!r->hasStream && g.stream == nullptr)
    {
        // the null stream cannot be captured: a renderer without a stream of its own gets one
        GS_CHECK(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
        r->hasStream = true;
        r->ownsStream = true;
        GS_CHECK(hipDeviceSynchronize()); // uploads and earlier frames issued on the null stream
    }
    StreamScope scope(r);
    const Screen scr = makeScreen(r->width, r->height, r->rowBegin, r->rowEnd);
    const int n = r->splatCount;
    const bool strictBand = scr.rowBegin > 0 || scr.rowEnd < scr.tilesY;
    const bool cull = n > 0 && (r->bandCull == GS_BAND_CULL_ON || (r->bandCull == GS_BAND_CULL_AUTO && strictBand));

    // No wait for the previous frame of this renderer: it may still be sorting or rasterising (render() returned when its pair
    // total was known).  What this call overwrites is the OTHER parameter slot, last read by the frame before the previous one,
    // which is complete -- the previous frame's emit pass ran behind it on the same stream and its total has been seen.
    if (r->paramsDevice == nullptr)
    {
        std::lock_guard<std::mutex> lock(g_captureMutex);
        GS_CHECK(hipMalloc((void**) &r->paramsDevice, sizeof(FrameParams) * kMaxViews));
        GS_CHECK(hipHostMalloc((void**) &r->paramsHost, 2 * kMaxViews * sizeof(FrameParams), hipHostMallocDefault));
    }
    const bool lazy = useLazyColours(r);

    FrameParams fp[kMaxViews] = {};
    for (int v = 0; v != r->views; ++v)
    {
        const GsCameraData* camera = cameras + v;
        std::memcpy(fp[v].camera.view, camera->view, sizeof(fp[v].camera.view));
        fp[v].camera.fovCotX = camera->fovCotangent[0];
        fp[v].camera.fovCotY = camera->fovCotangent[1];
        fp[v].camera.depthScale = camera->depthScaleBias[0];
        fp[v].camera.depthBias = camera->depthScaleBias[1];
        for (int k = 0; k != 3; ++k) fp[v].cameraPosition[k] = r->lastCameraPosition[k] = camera->position[k];
    }
    r->colourStale = lazy;

    // A frame whose pair demand exceeds the list capacity is NOT handed out (the reference would show it truncated and double
    // the list for the next one, Demo.cpp:357-366,447): the list is grown, the frame re-captured and the SAME camera rendered
    // again, so every frame that leaves this function is complete.
    for (;;)
    {
        const bool binned = r->pipeline == GS_PIPELINE_BINNED && binsSupported(scr) && radixSortSupported(4, n) &&
                            r->scratch.totalHostMapped != nullptr && r->views == 1;
        if (r->graphValid && (r->graphCapacity != r->tileListCapacity || r->graphSplats != n || r->graphRowBegin != scr.rowBegin ||
                              r->graphRowEnd != scr.rowEnd || r->graphCull != (int) cull || r->graphBinned != (int) binned ||
                              r->graphFootprint != scr.epanechnikov || r->graphLazy != (int) lazy || r->graphAblation != g_ablation))
            r->graphValid = false;
        if (!r->graphValid)
        {
            GS_CHECK(hipStreamSynchronize(g.stream)); // frames of the old instances drain before these are destroyed
            std::lock_guard<std::mutex> lock(g_captureMutex);
            // allocate everything the frame needs BEFORE capturing (no allocation may happen inside a capture)
            ensureTileListFastScratch(r->scratch, n * r->views);
            ensureRadixScratch(r->radix, n * r->views);
            ensureRadixScratch(r->radix, r->tileListCapacity);
            if (binned) ensureBinScratch(r->bins, n, (scr.rowEnd - scr.rowBegin) * scr.tilesX, binBatchesFor((scr.rowEnd - scr.rowBegin) * scr.tilesX));
            for (int slot = 0; slot != 2; ++slot)
            {
                if (r->graphExec[slot]) (void) hipGraphExecDestroy(r->graphExec[slot]);
                if (r->graph[slot]) (void) hipGraphDestroy(r->graph[slot]);
                r->graphExec[slot] = nullptr;
                r->graph[slot] = nullptr;
                GS_CHECK(hipStreamBeginCapture(g.stream, hipStreamCaptureModeThreadLocal));
                enqueueCapturableFrame(r, scr, cull, binned, lazy, slot);
                GS_CHECK(hipStreamEndCapture(g.stream, &r->graph[slot]));
                GS_CHECK(hipGraphInstantiate(&r->graphExec[slot], r->graph[slot], nullptr, nullptr, 0));
            }
            r->graphValid = true;
            r->graphCapacity = r->tileListCapacity;
            r->graphSplats = n;
            r->graphRowBegin = scr.rowBegin;
            r->graphRowEnd = scr.rowEnd;
            r->graphCull = (int) cull;
            r->graphBinned = (int) binned;
            r->graphFootprint = scr.epanechnikov;
            r->graphLazy = (int) lazy;
            r->graphAblation = g_ablation;
        }
        const int slot = r->graphSlot;
        r->graphSlot ^= 1;
        for (int v = 0; v != r->views; ++v) r->paramsHost[slot * kMaxViews + v] = fp[v];
        *r->scratch.totalHost = -1; // armed; the previous frame has left its total (seen by the host), nothing else writes the word
        const auto hostT0 = std::chrono::steady_clock::now();
        GS_CHECK(hipGraphLaunch(r->graphExec[slot], g.stream));
        const auto hostT1 = std::chrono::steady_clock::now();
        r->selector = r->graphSelector;
        r->keyBytes = r->graphKeyBytes;

        const int64_t demand = awaitPairTotal(r);
        const auto hostT2 = std::chrono::steady_clock::now();
        g_hostLaunchNs.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(hostT1 - hostT0).count(), std::memory_order_relaxed);
        g_hostAwaitNs.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(hostT2 - hostT1).count(), std::memory_order_relaxed);
        g_hostFrames.fetch_add(1, std::memory_order_relaxed);
        g.lastDemand = demand;
        r->tileListSize = (int32_t) std::min<int64_t>(demand, r->tileListCapacity);
        if (demand <= r->tileListCapacity) return 0;
        if (demand > 0x7FFFFFFF) return -2;
        GS_CHECK(hipStreamSynchronize(g.stream)); // the truncated frame drains before its buffers are replaced
        {
            std::lock_guard<std::mutex> lock(g_captureMutex);
            growTileListIfNeeded(r, demand);
        }
        r->graphValid = false;
        r->capturedRetries += 1;
    }
}
} // namespace

extern "C" void gs_renderer_set_graph(GsRenderer* r, int enabled)
{
    r->useGraph = enabled != 0;
    r->graphValid = false;
}

extern "C" int gs_renderer_captured_retries(const GsRenderer* r) { return r->capturedRetries; }
extern "C" void gs_debug_set_ablation(int mask) { g_ablation = mask; }
extern "C" long long gs_debug_host_times(double* launchMicroseconds, double* awaitMicroseconds, int reset)
{
    const long long frames = g_hostFrames.load();
    if (launchMicroseconds) *launchMicroseconds = frames ? 1e-3 * (double) g_hostLaunchNs.load() / (double) frames : 0.0;
    if (awaitMicroseconds) *awaitMicroseconds = frames ? 1e-3 * (double) g_hostAwaitNs.load() / (double) frames : 0.0;
    if (reset) g_hostLaunchNs = 0, g_hostAwaitNs = 0, g_hostFrames = 0;
    return frames;
}

extern "C" int gs_renderer_render(GsRenderer* r, const GsCameraData* camera, int timed, GsFrameStats* stats)
{
    // Captured path: plain frames (no stage timers, no stats) of the two-level pipeline on the native sort.
    if (r->useGraph && timed == 0 && stats == nullptr && r->pipeline != GS_PIPELINE_REFERENCE &&
        cfg.sortBackend == GS_SORT_BACKEND_NATIVE && !cfg.collectRasterStats && r->splatCount > 0 && r->views == 1)
        return renderCaptured(r, camera);
    r->graphValid = false; // the regular path re-sizes scratch as it likes
    const int rc = gs_renderer_begin(r, camera, timed);
    if (rc != 0) return rc;
    return gs_renderer_finish(r, stats);
}

// r->views frames by ONE chain of kernels (header: gs_renderer_set_views); captured frames of the two-level pipeline only.
extern "C" int gs_renderer_render_views(GsRenderer* r, const GsCameraData* cameras)
{
    if (r == nullptr || cameras == nullptr) return -1;
    if (r->views == 1) return gs_renderer_render(r, cameras, 0, nullptr);
    const bool coloursPerView = r->shDegree != 0 && !useLazyColours(r); // an SH pass would need one colour buffer per view
    if (!r->useGraph || r->pipeline != GS_PIPELINE_TWO_LEVEL || cfg.sortBackend != GS_SORT_BACKEND_NATIVE || cfg.collectRasterStats ||
        r->splatCount <= 0 || coloursPerView || !radixSortSupported(4, r->splatCount * r->views))
        return -3;
    return renderCaptured(r, cameras);
}

extern "C" int gs_renderer_collect_stage_times(GsRenderer* r, GsFrameStats* sums)
{
    // Sums over the frames rendered with timed == 2 since the last call; resets the accumulation.
    StreamScope scope(r);
    GS_CHECK(hipStreamSynchronize(g.stream));
    GsFrameStats s{};
    float* slots[kStageTimers] = {&s.msSphericalHarmonics, &s.msClipData, &s.msBuildTileList, &s.msSortTileList, &s.msTileRanges,
                                  &s.msRasterize, &s.msSortTileList};
    for (size_t i = 0; i != r->asyncUsed; ++i) *slots[i % kStageTimers] += gs_timer_elapsed_ms(r->asyncTimers[i]);
    s.tileListSize = r->asyncPairs;
    const int frames = r->asyncFrames;
    r->asyncUsed = 0;
    r->asyncFrames = 0;
    r->asyncPairs = 0;
    if (sums) *sums = s;
    return frames;
}

extern "C" const char* gs_version(void) { return "gsplat_hip 0.2 (gfx950)"; }
