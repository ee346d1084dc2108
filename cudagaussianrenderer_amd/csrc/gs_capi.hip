// gs_capi.hip -- the C ABI of include/gsplat_hip.h, part 1: library state, timers, device buffers and the eight
// staged host functions (GaussianRender.cuh:118-143).  The one-call renderer is gs_renderer.hip, its captured frames gs_capture.hip.
#include <algorithm>
#include <cstring>

#include "gs_state.h"

namespace gs
{

int g_footprint = GS_FOOTPRINT_GAUSSIAN;
int g_ablation = 0;
thread_local State g;
Settings cfg;

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
    s.epanechnikov = effectiveFootprint() == GS_FOOTPRINT_EPANECHNIKOV ? 1 : 0; // (the calling renderer's own, or the process-wide switch)
    s.extentFactor = s.epanechnikov ? sqrtf(7.0f) : 3.0f;
    return s;
}

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


} // namespace gs

using namespace gs;

// ---- timer (Utilities.h:155-187) ------------------------------------------------------------------
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
// A copy on the LEGACY (null) stream depends on every blocking stream of the process, and another renderer's thread may be CAPTURING on
// one (a frame re-captured after a list growth or a change of depth layers; a torch stream handed in through gs_renderer_set_stream is a
// blocking stream): "operation would make the legacy stream depend on a capturing blocking stream".  A thread whose stream is its own
// (StreamScope of a renderer with a stream, gs_set_stream) copies on that stream; a thread on the null stream -- the staged API and
// DeviceBuffer<T> of a host that keeps the reference's conventions -- keeps the reference's semantics (the copy waits for everything
// enqueued before it) and takes the capture mutex instead: captures and legacy-stream copies never overlap.
extern "C" void gs_device_upload(void* dst, const void* src, size_t bytes)
{
    if (!bytes) return;
    if (g.stream != nullptr)
    {
        GS_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g.stream));
        GS_CHECK(hipStreamSynchronize(g.stream)); // (the source may be pageable and is the caller's again when this returns, like hipMemcpy)
        return;
    }
    std::lock_guard<std::mutex> lock(g_captureMutex);
    GS_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
}
extern "C" void gs_device_download(void* dst, const void* src, size_t bytes)
{
    if (!bytes) return;
    if (g.stream != nullptr)
    {
        GS_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, g.stream));
        GS_CHECK(hipStreamSynchronize(g.stream));
        return;
    }
    std::lock_guard<std::mutex> lock(g_captureMutex);
    GS_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
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

namespace gs
{
void freeStagedTileOrder(StagedTileOrder& t)
{
    if (t.consumedPair) (void) hipFree(t.consumedPair);
    if (t.order) (void) hipFree(t.order);
    if (t.state) (void) hipFree(t.state);
    t = StagedTileOrder{};
}
namespace
{
// gs_set_staged_tile_start_order: the staged clip call's part -- buffers for the current screen and band (a change starts the history
// afresh), the host-side look at the camera, and what the clip kernel's first eight workgroups need to make this frame's table.
bool stagedTileOrderArgs(const Screen& scr, TileOrderArgs* args)
{
    StagedTileOrder& t = g.stagedTileOrder;
    t.builtThisFrame = false;
    if (cfg.stagedTileStartOrder != GS_TILE_START_ORDER_PREVIOUS_FRAME) return false;
    const int slots = rasterGridSlots(scr.tilesX, scr.rowEnd - scr.rowBegin);
    if (slots <= 0) return false;
    if (t.order == nullptr || t.slots != slots || t.tiles != scr.totalTiles || t.firstTile != scr.rowBegin * scr.tilesX || t.rows != scr.rowEnd - scr.rowBegin)
    {
        std::lock_guard<std::mutex> lock(g_captureMutex); // (allocation: never while another thread captures)
        GS_CHECK(hipStreamSynchronize(g.stream));
        freeStagedTileOrder(t);
        const size_t tiles = (size_t) scr.totalTiles;
        GS_CHECK(hipMalloc((void**) &t.consumedPair, sizeof(int32_t) * 2 * tiles));
        GS_CHECK(hipMalloc((void**) &t.order, sizeof(int32_t) * (size_t) slots));
        GS_CHECK(hipMalloc((void**) &t.state, sizeof(int32_t) * 16));
        GS_CHECK(hipMemsetAsync(t.consumedPair, 0, sizeof(int32_t) * 2 * tiles, g.stream));
        GS_CHECK(hipMemsetAsync(t.order, 0xFF, sizeof(int32_t) * (size_t) slots, g.stream));
        GS_CHECK(hipMemsetAsync(t.state, 0, sizeof(int32_t) * 16, g.stream));
        t.slots = slots;
        t.tiles = scr.totalTiles;
        t.firstTile = scr.rowBegin * scr.tilesX;
        t.rows = scr.rowEnd - scr.rowBegin;
    }
    // (the renderer's rule, gs_renderer.hip cameraJumped: a rotation entry moved by more than 0.06 -- about 3 degrees -- or no frame before)
    const float* view = g.global.cameraData.view;
    bool jumped = !t.haveLastView;
    for (int c = 0; c != 3 && !jumped; ++c)
        for (int k = 0; k != 3; ++k)
            if (!(fabsf(view[c * 4 + k] - t.lastView[c * 4 + k]) <= 0.06f)) jumped = true;
    std::memcpy(t.lastView, view, sizeof(t.lastView));
    t.haveLastView = true;
    *args = TileOrderArgs{};
    args->now = t.consumedPair + (size_t) ((t.parity & 1) ^ 1) * (size_t) t.tiles; // the frame that just ended
    args->prev = t.consumedPair + (size_t) (t.parity & 1) * (size_t) t.tiles;      // cleared: this frame's rasterizeTile writes it
    args->order = t.order;
    args->state = t.state;
    args->slots = t.slots;
    args->skip = jumped ? 1 : 0;
    t.builtThisFrame = true;
    return true;
}
} // namespace
} // namespace gs

extern "C" void gs_set_staged_tile_start_order(int mode)
{
    cfg.stagedTileStartOrder = mode == GS_TILE_START_ORDER_PREVIOUS_FRAME ? GS_TILE_START_ORDER_PREVIOUS_FRAME : GS_TILE_START_ORDER_SCREEN;
}
extern "C" int gs_get_staged_tile_start_order(void) { return cfg.stagedTileStartOrder; }
extern "C" int gs_debug_staged_tile_order_state(int32_t out[4])
{
    for (int k = 0; k != 4; ++k) out[k] = 0;
    if (g.stagedTileOrder.state == nullptr) return -1;
    gs_device_download(out, g.stagedTileOrder.state, sizeof(int32_t) * 4);
    return 0;
}

extern "C" void gs_evaluate_splat_clip_data(GsTimer* timer, int32_t count)
{
    const Screen scr = currentScreen();
    TileOrderArgs tileOrder;
    const bool ordered = count > 0 && stagedTileOrderArgs(scr, &tileOrder); // (before the timed interval: allocation is not the stage's work)
    if (!ordered) g.stagedTileOrder.builtThisFrame = false;
    gs_timer_start(timer);
    launchSplatClipData(g.stream, count, scr, currentCamera(), (const float4*) g.global.position,
                        (const float4*) g.global.scaleAndRotation, (float2*) g.global.positionClipSpaceXY,
                        g.global.positionClipSpaceZ, (float4*) g.global.screenEllipse, (float4*) g.global.conic, nullptr, nullptr, nullptr, nullptr, nullptr,
                        nullptr, nullptr, 0, nullptr, nullptr, 1, nullptr, nullptr, nullptr, ordered ? &tileOrder : nullptr);
    gs_timer_stop(timer);
    reportLaunch("evaluateSplatsClipData");
}

extern "C" int32_t gs_build_tile_list(GsTimer* timer, int32_t /*numBlocks*/, int32_t tileListCapacity)
{
    const Screen scr = currentScreen();
    const int count = g.global.splatCount;
    // Default: the pairs are emitted in the order the sort would leave them (gs_staged.hip); sortTileList then finds nothing to do.
    // (-1: no device memory for that builder's scratch -- this call emits in the reference's order, which needs none)
    // (a capacity within a megabyte of 2^31 -- the caller's own buffers, the reference's int32 -- is beyond what the sorted builder's
    // grids can be sized for without an int wrapping: such a call emits in the reference's order too, and sortTileList sorts)
    if (cfg.stagedBuilder == GS_STAGED_BUILDER_SORTED && radixSortSupported(4, count) && tileListCapacity <= kMaxTileListCapacity)
    {
        const int32_t emitted = stagedBuildTileListSorted(timer, tileListCapacity);
        if (emitted >= 0) return emitted;
    }
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
    sortTileListImpl(timer, size, deviceTempStorage, tempStorageSizeInBytes, keys, values, cfg.stagedBuilder == GS_STAGED_BUILDER_SORTED);
}

void gs::sortTileListImpl(GsTimer* timer, int32_t size, void** deviceTempStorage, size_t* tempStorageSizeInBytes, GsDoubleBufferU64* keys,
                          GsDoubleBufferI32* values, bool lookFirst)
{
    gs_timer_start(timer);
    // A stable sort of a list that is already in order on the compared bits is the identity: look first (one read of the keys).  The
    // default staged builder emits exactly such a list; any other list goes through the sort as before.
    if (size > 0 && size <= kMaxTileListCapacity && lookFirst && keys->selector == values->selector &&
        stagedListIsSorted(size, keys->buffers[keys->selector], sortEndBit(currentScreen().totalTiles)))
    {
        gs_timer_stop(timer);
        if (hipGetLastError() != hipSuccess) printf("error sortTileList\n");
        return;
    }
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
    cfg.sortBackend = (backend == GS_SORT_BACKEND_ROCPRIM) ? GS_SORT_BACKEND_ROCPRIM : GS_SORT_BACKEND_NATIVE;
}
extern "C" int gs_get_sort_backend(void) { return cfg.sortBackend; }
extern "C" void gs_set_list_fusion(int enabled) { cfg.listFusion = enabled != 0; }
extern "C" int gs_get_list_fusion(void) { return cfg.listFusion ? 1 : 0; }

extern "C" int gs_radix_sort_pairs(int keyBytes, void* keys0, void* keys1, int32_t* values0, int32_t* values1, int32_t count, int beginBit,
                                   int endBit)
{
    if (!radixSortSupported(keyBytes, count)) return -1;
    return launchRadixSortPairs(g.stream, g.radix, keyBytes, keys0, keys1, values0, values1, count, beginBit, endBit);
}

extern "C" void gs_debug_set_raster_trace(uint64_t* deviceBufferOrNull) { setRasterTrace((unsigned long long*) deviceBufferOrNull); }
extern "C" void gs_set_raster_pixels_per_lane(int ppl) { setRasterPixelsPerLane(ppl); }
extern "C" int gs_get_raster_pixels_per_lane(void) { return rasterPixelsPerLane(); }

extern "C" void gs_rasterize_tile(GsTimer* timer)
{
    unsigned long long* consumed = cfg.collectRasterStats ? cfg.consumedDevice : nullptr;
    if (consumed) GS_CHECK(hipMemsetAsync(consumed, 0, sizeof(unsigned long long), g.stream));
    // gs_set_staged_tile_start_order: the table this frame's evaluateSplatClipData made (same screen, same band), and this frame's counters
    const Screen scr = currentScreen();
    StagedTileOrder& order = g.stagedTileOrder;
    RasterSchedule schedule;
    const bool ordered = order.builtThisFrame && cfg.stagedTileStartOrder == GS_TILE_START_ORDER_PREVIOUS_FRAME && order.tiles == scr.totalTiles &&
                         order.slots == rasterGridSlots(scr.tilesX, scr.rowEnd - scr.rowBegin) && order.firstTile == scr.rowBegin * scr.tilesX;
    if (ordered)
    {
        schedule.order = order.order;
        schedule.consumedOut = order.consumedPair + (size_t) (order.parity & 1) * (size_t) order.tiles;
        schedule.enabled = order.state;
    }
    gs_timer_start(timer);
    launchRasterize(g.stream, scr, g.global.tileRange, g.tileList.values, (const float2*) g.global.positionClipSpaceXY,
                    (const float4*) g.global.conic, (const float4*) g.global.color, (uchar4*) g.global.backBuffer, consumed, false, nullptr, nullptr, 1, 0, nullptr,
                    ordered ? &schedule : nullptr);
    gs_timer_stop(timer);
    if (ordered)
    {
        order.parity ^= 1;
        order.builtThisFrame = false; // (one table, one frame: a second rasterizeTile without a clip call in between draws in screen order)
    }
    reportLaunch("renderDepthBuffer");
}

extern "C" const char* gs_version(void) { return "gsplat_hip 0.6 (gfx950)"; }
