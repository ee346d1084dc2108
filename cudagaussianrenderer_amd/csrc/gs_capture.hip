// gs_capture.hip -- the C ABI of include/gsplat_hip.h, part 3: captured frames (hipGraph replay) and frames per chain.
//
// The two-level frame with every list size left on the device (the pair count feeds the tile sort and the range kernel
// through pointers, grids are sized for the list capacity), so the frame has no host round trip and can be captured once
// and replayed as ONE graph launch per frame.  The camera travels in a small device struct that a memcpy node refreshes.
// The pair total reaches the host through a host-mapped word in the middle of the frame; a frame whose demand exceeds the
// list capacity is rendered again with a larger list before the call returns (never handed out truncated).
#include <atomic>
#include <chrono>
#include <cstring>

#include "gs_state.h"

using namespace gs;

namespace gs
{
std::mutex g_captureMutex;
namespace
{
// host-side cost of captured frames (diagnostic, gs_debug_host_times): time inside hipGraphLaunch, time waiting for the pair total
std::atomic<long long> g_hostLaunchNs{0}, g_hostAwaitNs{0}, g_hostFrames{0};

// What gs_renderer_debug_layered_raster_report collects while it enqueues a layered frame with plain launches: an event pair around
// every layer's raster launch and four counters per layer (entries consumed, tiles run, tiles that loaded / stored their state).
struct LayerProbe
{
    hipEvent_t start[kMaxDepthLayers], stop[kMaxDepthLayers];
    unsigned long long* counters = nullptr; // device, 4 per layer, zeroed by the caller
    int layers = 0;
};

int enqueueCapturableFrame(GsRenderer* r, const Screen& scr, bool cull, bool lazy, int slot, LayerProbe* probe = nullptr)
{
#ifndef GS_PARAMS_FROM_HOST
#define GS_PARAMS_FROM_HOST 0 // A/B builds: 1 = the kernels read the cameras from the pinned host slot itself (no copy node in the graph)
#endif
    // The cameras of the frame: a memcpy node at the head of the graph refreshes them in device memory from this instance's pinned slot.
    FrameParams* const params = GS_PARAMS_FROM_HOST ? r->paramsHost + slot * kMaxViews : r->paramsDevice;
    const RasterColourSource lazySource = colourSource(r, nullptr, params);
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

    if (!GS_PARAMS_FROM_HOST)
        GS_CHECK(hipMemcpyAsync(r->paramsDevice, r->paramsHost + slot * kMaxViews, sizeof(FrameParams) * views, hipMemcpyHostToDevice, g.stream));
    // Depth layers (gs_renderer_set_depth_layers): only where the emit pass carries the tile sort's first pass (two-pass sorts of 16-bit tile
    // indices) and with enough splats for every layer to have some.  Under a band cull the list's length lives on the device: every layer's
    // slice, the first one's too, is then cut out by the compaction kernels (gs_tiles_fast.hip LayerSlice), which needs the compaction on
    const int fusedBits = effectiveListFusion() ? tileListFusedSortBits(tBits, tileKeyBytes) : 0;
    float layerFraction[kMaxDepthLayers];
    const int wantLayers = effectiveDepthLayers(r, layerFraction);
    // ... and with the host-mapped word the layers' totals travel through: without one (hipHostGetDevicePointer failed) every layer's
    // emit pass would copy ITS total into the word the host waits on -- the host would take the first layer's for the frame's, never see a
    // later, larger layer overflow the list, and read a stale 0 as the listed sum: such a renderer renders one list per frame
    const int layers = (wantLayers > 1 && (!cull || r->layerCompaction) && fusedBits > 0 && nv >= 4096 && g_ablation == 0 &&
                        r->scratch.totalHostMapped != nullptr) ? wantLayers : 1;
    ushort4* tileBox = (layers > 1 && r->layerCompaction) ? r->layerTileBox : nullptr; // (the later layers' live test, gs_tiles_fast.hip)
    // the range table is reset to -1 by the clip kernel (one node less than a memset; see gs_ranges.hip) -- and the layers' tile status
    // bytes behind it with it (0xFF = untouched)
    int32_t* clearTable = r->tileRange;
    const int clearEntries = layers > 1 ? (int) rangeTableInts(scr, views) : 2 * virtualTiles;
#ifdef GS_DEBUG_MEMSET_NODE
    // audit build (tools/micro/graph_memset.hip): the table is cleared by a memset NODE instead
    if (clearTable != nullptr) GS_CHECK(hipMemsetAsync(clearTable, 0xFF, sizeof(int32_t) * (size_t) clearEntries, g.stream));
    clearTable = nullptr;
#endif
    if (r->shDegree != 0 && !cull && !lazy)
        launchSphericalHarmonics(g.stream, n, r->shDegree, r->shCount, (const float4*) r->position, r->sh, unused, (float4*) r->color,
                                 params);
    // gs_renderer_set_tile_start_order: the raster launches start their tiles by the table the previous frame left; the kernel that makes
    // the next frame's table follows the frame's last raster launch
    RasterSchedule schedule;
    // (its buffers were made before the capture began: prepareTileStartOrder.  The two graph instances are launched alternately, so the
    // instance's slot is the frame's parity: this frame's counters in one buffer, the previous frame's in the other)
    const bool orderActive = tileStartOrderSchedule(r, scr, slot, &schedule);
    const RasterSchedule* scheduleOrNull = &schedule;
    // ... and the table itself is made at the start of the frame from the previous frame's counters: by the clip kernel's first eight
    // workgroups (plain, or riding in the sort's scan launches), or (band cull) by a small kernel of its own right here
    TileOrderArgs orderArgs = orderActive ? tileOrderArgsFor(r, scr, slot) : TileOrderArgs{};
    if (orderActive) orderArgs.skipFlagDevice = &params[0].pad; // (renderCaptured sets it per replay: the camera jumped)
    bool orderBuilt = !orderActive;
    if (!orderBuilt && (cull || (g_ablation & 1)))
    {
        launchTileStartOrder(g.stream, scr, orderArgs);
        orderBuilt = true;
    }
    const int* countDevice = nullptr;
    // Morton storage: the depth sort makes its own keys, so the clip kernel and the sort do not depend on each other -- the clip kernel
    // is handed to the sort and shares the launch of its first column scan (gs_internal.h RadixScanCompanion,
    // gs_debug_set_clip_scan_fusion): 17 kernels per frame; the tile start order's builder rides in the same launch.
    ClipWithScan deferredClip;
    bool clipDeferred = false;
    const auto launchClip = [&]() {
        launchSplatClipData(g.stream, n, scr, unusedCamera, (const float4*) r->position, (const float4*) r->scaleAndRotation, nullptr, nullptr, nullptr,
                            nullptr, r->splatToStorage != nullptr ? nullptr : r->depthKey[0],
                            r->splatToStorage != nullptr ? nullptr : (uint32_t*) r->order[0], nullptr, nullptr, nullptr, params, clearTable,
                            clearEntries, (float4*) r->tileRecord, (float4*) r->rasterRecord, views, nullptr, tileBox, nullptr, orderBuilt ? nullptr : &orderArgs);
        orderBuilt = true;
    };
    if (!cull)
    {
        if (!(g_ablation & 1))
        {
            if (cfg.clipScanFusion && r->splatToStorage != nullptr && !(g_ablation & 2))
            {
                deferredClip.count = n;
                deferredClip.scr = scr;
                deferredClip.params = params;
                deferredClip.position = (const float4*) r->position;
                deferredClip.scaleAndRotation = (const float4*) r->scaleAndRotation;
                deferredClip.rangeTableToClear = clearTable;
                deferredClip.rangeEntries = clearEntries;
                deferredClip.tileRecord = (float4*) r->tileRecord;
                deferredClip.rasterRecord = (float4*) r->rasterRecord;
                deferredClip.views = views;
                deferredClip.tileBox = tileBox;
                deferredClip.parts = kClipScanParts;
                if (!orderBuilt) deferredClip.tileOrder = orderArgs;
                orderBuilt = true;
                clipDeferred = true;
                r->radix.firstScanCompanion = RadixScanCompanion{&clipWithScanCompanion, &deferredClip, kClipScanParts};
            }
            else launchClip();
        }
    }
    else
    {
        // (Morton storage: in the caller's order, from the caller-ordered copies of the inputs)
        // (... unless the cull runs by blocks of Morton neighbours: storage order, only the blocks that can reach the band, gs_clip.hip)
        BandBlockCull blocks;
        const bool byBlocks = useBlockCull(r, &blocks);
        const bool callerOrder = r->splatToStorage != nullptr && !byBlocks;
        launchSplatClipData(g.stream, n, scr, unusedCamera, (const float4*) (callerOrder ? r->positionCaller : r->position),
                            (const float4*) (callerOrder ? r->scaleAndRotationCaller : r->scaleAndRotation), nullptr, byBlocks ? nullptr : r->clipZ, nullptr, nullptr, nullptr,
                            nullptr, r->bandMask, r->bandSlotPrefix, r->bandCounts, params, clearTable, clearEntries, (float4*) r->tileRecord,
                            (float4*) r->rasterRecord, views, callerOrder ? r->splatToStorage : nullptr, tileBox, byBlocks ? &blocks : nullptr);
        launchBandCompaction(g.stream, n, views, r->bandMask, r->bandSlotPrefix, r->bandCounts, r->bandOffsets, r->bandTotal, r->clipZ, r->depthKey[0],
                             (uint32_t*) r->order[0], r->splatToStorage, byBlocks ? (const float4*) r->positionCaller : nullptr, nullptr, params);
        countDevice = (const int*) r->bandTotal;
        if (r->shDegree != 0 && !lazy)
            launchSphericalHarmonicsGather(g.stream, n, countDevice, r->shDegree, r->shRowStride, r->order[0], (const float4*) r->position,
                                           r->shPerSplat, unused, (float4*) r->color, params);
    }
    r->radix.countDevice = countDevice;
    r->scratch.countDevice = countDevice;
    // (Morton storage, no cull: the sort's input is in the caller's order -- its first pass makes the keys from the caller-ordered
    // positions and the cameras on the device; the culled list was compacted in that order already)
    DepthKeySource keySource;
    keySource.positionCaller = (const float4*) r->positionCaller;
    keySource.params = params;
    keySource.splatsPerView = n;
    const bool callerOrderedInput = !cull && r->splatToStorage != nullptr;
    const int orderSel = (g_ablation & 2) ? 0
                                          : launchRadixSortPairs(g.stream, r->radix, 4, r->depthKey[0], r->depthKey[1], r->order[0], r->order[1], nv, 0,
                                                                 kDepthKeyBits, callerOrderedInput ? r->splatToStorage : nullptr,
                                                                 callerOrderedInput ? &keySource : nullptr);
    if (clipDeferred)
    {
        // what the sort did not take along (rows of more than 4096 tiles: nothing of it; fewer scans than slices): by itself, then
        r->radix.firstScanCompanion = RadixScanCompanion{};
        launchSplatClipDataRest(g.stream, deferredClip);
    }
    const uint32_t* order = (const uint32_t*) r->order[orderSel];
    // (gs_set_list_fusion: the emit pass carries the first pass of the tile sort.  Under ablation -- cumulative from the back -- the
    // stages in front of a missing one still run as in the product; every position the emit pass writes is checked against the capacity)
    const int sortBits = fusedBits;
    if (layers > 1)
    {
        // layer l = sorted (virtual) splats [begin, end): count / emit / second tile pass / raster, each later layer without the pairs of the
        // tiles an earlier one stopped.  Every launch below reuses the same scratch and list buffers: the layers run one after another.
        int begin = 0;
        for (int l = 0; l != layers; ++l)
        {
            const int end = l + 1 == layers ? nv : std::min(nv, std::max(begin + 1, (int) ((double) layerFraction[l] * (double) nv)));
            const int cnt = end - begin;
            r->scratch.openBits = l > 0 ? r->tileOpenBits : nullptr;
            r->scratch.openWordsPerRow = openWordsPerRow(scr.tilesX);
            r->scratch.frameTotals = r->frameTotals;
            r->scratch.layerFlags = (l == 0 ? 1 : 0) | (l + 1 == layers ? 2 : 0);
            const uint32_t* layerOrder = order + begin;
            r->scratch.countDevice = nullptr;
            uint32_t* live = (uint32_t*) r->order[orderSel ^ 1]; // (the depth sort's other buffer is free by now)
            if (cull)
            {
                // a band's culled list: its length is on the device, so the compaction kernels cut the layer's slice out of the whole list
                // themselves -- the first layer's too, with every splat of it live
                launchLayerCompaction(g.stream, scr, nv, views > 1 ? n : 0, r->layerTileBox, l > 0 ? r->tileOpenBits : nullptr, openWordsPerRow(scr.tilesX),
                                      r->layerAlive, order, nv, r->layerBlockCounts, r->layerLiveNibbles, live, r->layerLiveCount, r->layerBlockOffsets,
                                      r->layerScanTotal, countDevice, layerFraction, l, layers);
                layerOrder = live;
                r->scratch.countDevice = r->layerLiveCount;
            }
            else if (l > 0 && r->layerCompaction)
            {
                // the later layers walk only the splats that still have an open tile under their box (gs_tiles_fast.hip)
                // (experiment, gs_renderer_debug_set_layer_compaction_reuse: lists in buffers that outlive the frame; mode 2 walks them again
                // without launching the three compaction kernels)
                const bool keep = r->layerReuse != 0 && r->layerLiveDebug != nullptr;
                uint32_t* liveList = keep ? r->layerLiveDebug + (size_t) l * (size_t) nv : live;
                int32_t* liveCount = keep ? r->layerLiveCountDebug + l : r->layerLiveCount;
                if (!(keep && r->layerReuse == 2))
                    launchLayerCompaction(g.stream, scr, nv, views > 1 ? n : 0, r->layerTileBox, r->tileOpenBits, openWordsPerRow(scr.tilesX), r->layerAlive, order + begin, cnt, r->layerBlockCounts, r->layerLiveNibbles, liveList,
                                          liveCount, r->layerBlockOffsets, r->layerScanTotal);
                layerOrder = liveList;
                r->scratch.countDevice = liveCount;
            }
            const int walk = cull ? nv : cnt; // (splats the list passes' grids are sized for)
            launchTileListFastCount(g.stream, walk, scr, layerOrder, (const float2*) r->clipXY, (const float4*) r->ellipse, r->scratch, sortBits,
                                    views > 1 ? n : 0);
            launchTileListFastEmit(g.stream, walk, scr, layerOrder, (const float2*) r->clipXY, (const float4*) r->ellipse, r->scratch, r->keys[0],
                                   tileKeyBytes, r->values[0], r->tileListCapacity, views > 1 ? n : 0, sortBits);
            r->scratch.countDevice = nullptr;
            r->radix.countDevice = (const int*) r->scratch.total;
            r->radix.rangeTableForLastPass = r->tileRange;
            const int sel = launchRadixSortPairs(g.stream, r->radix, tileKeyBytes, r->keys[0], r->keys[1], r->values[0], r->values[1],
                                                 r->tileListCapacity, sortBits, tBits);
            r->radix.countDevice = nullptr;
            r->radix.rangeTableForLastPass = nullptr;
            RasterLayer layer;
            layer.status = r->tileStatus;
            layer.state = r->layerState;
            layer.consumedSoFar = r->layerConsumed;
            layer.rangeToReset = r->tileRange;
            layer.openBits = r->tileOpenBits;
            layer.openWordsPerRow = openWordsPerRow(scr.tilesX);
            layer.isFirst = l == 0;
            layer.isLast = l + 1 == layers;
            if (probe != nullptr) GS_CHECK(hipEventRecord(probe->start[l], g.stream));
            launchRasterize(g.stream, scr, r->tileRange, r->values[sel], (const float2*) r->clipXY, (const float4*) r->conic, (const float4*) r->color,
                            (uchar4*) r->backBuffer, (probe != nullptr && probe->counters != nullptr) ? probe->counters + 4 * l : nullptr, true, lazyColours, (const float4*) r->rasterRecord, views, n, &layer,
                            scheduleOrNull);
            if (probe != nullptr)
            {
                GS_CHECK(hipEventRecord(probe->stop[l], g.stream));
                probe->layers = l + 1;
            }
            r->graphSelector = sel;
            begin = end;
        }
        r->scratch.openBits = nullptr;
        r->scratch.frameTotals = nullptr;
        r->scratch.layerFlags = 0;
        r->scratch.countDevice = nullptr;
        r->graphKeyBytes = tileKeyBytes;
        r->graphLayersActive = true;
        return 0;
    }
    r->graphLayersActive = false;
    if (!(g_ablation & 4))
        launchTileListFastCount(g.stream, nv, scr, order, (const float2*) r->clipXY, (const float4*) r->ellipse, r->scratch, sortBits,
                                views > 1 ? n : 0);
    if (!(g_ablation & 8))
        launchTileListFastEmit(g.stream, nv, scr, order, (const float2*) r->clipXY, (const float4*) r->ellipse, r->scratch, r->keys[0],
                               tileKeyBytes, r->values[0], r->tileListCapacity, views > 1 ? n : 0, sortBits);
    // tile sort + ranges on the device-side pair count (low word of the int64 total), grids sized for the capacity
    const int* pairCount = (const int*) r->scratch.total;
    r->radix.countDevice = pairCount;
    r->scratch.countDevice = nullptr;
    // (the last pass of the tile sort leaves the tile ranges: no range kernel, gs_radix.hip)
    r->radix.rangeTableForLastPass = (g_ablation & 32) ? nullptr : r->tileRange;
    const int sel = (g_ablation & 16) ? 0
                                      : launchRadixSortPairs(g.stream, r->radix, tileKeyBytes, r->keys[0], r->keys[1], r->values[0], r->values[1],
                                                             r->tileListCapacity, sortBits, tBits);
    r->radix.countDevice = nullptr;
    r->radix.rangeTableForLastPass = nullptr;
    if (!(g_ablation & 64))
        launchRasterize(g.stream, scr, r->tileRange, r->values[sel], (const float2*) r->clipXY, (const float4*) r->conic,
                        (const float4*) r->color, (uchar4*) r->backBuffer, nullptr, true, lazyColours, (const float4*) r->rasterRecord, views, n, nullptr, scheduleOrNull);
    r->graphSelector = sel;
    r->graphKeyBytes = tileKeyBytes;
    return 0;
}

// The pair total of a captured frame lands in a host-mapped word (written by the emit pass).
// The host arms the word with -1 before the launch and waits for the frame's own value: a few microseconds of polling in
// the middle of the frame, long before its raster kernel ends.
// The wait is bounded twice over: any stream status other than "not ready" ends it (a faulted queue never writes the word, and
// hipStreamQuery then reports the fault for ever), and so does the wall clock (gs_debug_set_await_timeout, 30 s by default) -- both
// the reference's way (CudaErrorCheck.cpp:2-11): print and exit, so that a fault on one rank of a multi-GPU job ends the job instead
// of leaving a host thread spinning.
double g_awaitTimeoutSeconds = 30.0;

int64_t awaitPairTotal(GsRenderer* r)
{
    volatile int64_t* word = r->scratch.totalHost;
    std::chrono::steady_clock::time_point t0;
    bool clockStarted = false; // (the spin counter wraps: the start of the wait is taken once)
    for (unsigned spins = 0;; ++spins)
    {
        const int64_t v = *word;
        if (v >= 0) return v;
        if ((spins & 63u) == 63u)
        {
            const hipError_t status = hipStreamQuery(g.stream);
            if (status == hipSuccess)
            {
                // the whole frame is done: every write of it is visible by now
                const int64_t w = *word;
                if (w >= 0) return w;
                int64_t fromDevice = 0; // (no host-mapped address: the total was left in device memory)
                // (on the renderer's own stream: a copy on the legacy stream would depend on every blocking stream of the process, and
                // another renderer's thread may be CAPTURING on one -- "operation would make the legacy stream depend on a capturing
                // blocking stream")
                GS_CHECK(hipMemcpyAsync(&fromDevice, r->scratch.total, sizeof(int64_t), hipMemcpyDeviceToHost, g.stream));
                GS_CHECK(hipStreamSynchronize(g.stream));
                return fromDevice;
            }
            if (status != hipErrorNotReady)
            {
                printf("HIP error at %s %d: %s (stream status while waiting for a captured frame's pair total)\n", __FILE__, __LINE__,
                       hipGetErrorString(status));
                exit(EXIT_FAILURE);
            }
            if (!clockStarted)
            {
                t0 = std::chrono::steady_clock::now();
                clockStarted = true;
            }
            else if ((spins & 0xFFFFu) == 0xFFFFu &&
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > g_awaitTimeoutSeconds)
            {
                printf("gsplat_hip: no pair total from a captured frame after %.1f s (stream still busy): giving up\n", g_awaitTimeoutSeconds);
                exit(EXIT_FAILURE);
            }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}
} // namespace

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
    if (r->tileStartOrder == GS_TILE_START_ORDER_PREVIOUS_FRAME && r->views == 1) fp[0].pad = cameraJumped(r, cameras[0].view) ? 1.0f : 0.0f;
    r->colourStale = lazy;

    // A frame whose pair demand exceeds the list capacity is NOT handed out (the reference would show it truncated and double
    // the list for the next one, Demo.cpp:357-366,447): the list is grown, the frame re-captured and the SAME camera rendered
    // again, so every frame that leaves this function is complete.
    for (;;)
    {
        if (r->graphValid && (r->graphCapacity != r->tileListCapacity || r->graphSplats != n || r->graphRowBegin != scr.rowBegin ||
                              r->graphRowEnd != scr.rowEnd || r->graphCull != (int) cull ||
                              r->graphFootprint != scr.epanechnikov || r->graphLazy != (int) lazy || r->graphAblation != g_ablation ||
                              r->graphFusion != ((int) effectiveListFusion() | ((int) cfg.clipScanFusion << 1) | (effectivePixelsPerLane() << 2) | ((int) r->blockCull << 5) | (r->tileStartOrder << 6) | ((int) r->tileOrderManual << 7)) || r->graphLayers != (r->depthLayers > 0 ? r->depthLayers : r->autoLayers)))
            r->graphValid = false;
        if (!r->graphValid)
        {
            GS_CHECK(hipStreamSynchronize(g.stream)); // frames of the old instances drain before these are destroyed
            prepareTileStartOrder(r, scr);            // (allocates under the capture mutex itself: before this block takes it)
            std::lock_guard<std::mutex> lock(g_captureMutex);
            // allocate everything the frame needs BEFORE capturing (no allocation may happen inside a capture)
            ensureTileListFastScratch(r->scratch, n * r->views);
            ensureRadixScratch(r->radix, n * r->views);
            ensureRadixScratch(r->radix, r->tileListCapacity);
            float cutsUnused[kMaxDepthLayers];
            if (effectiveDepthLayers(r, cutsUnused) > 1)
            {
                // what layered frames keep between their layers: 16 bytes per pixel and view, 8.4 bytes per (virtual) splat, 4 per tile --
                // only a renderer that is about to capture a layered frame pays for it
                if (r->layerState == nullptr) r->layerState = (float4*) gs_device_alloc(sizeof(float4) * (size_t) r->width * r->height * r->views);
                if (r->layerConsumed == nullptr) r->layerConsumed = (int32_t*) gs_device_alloc(sizeof(int32_t) * (size_t) scr.totalTiles * r->views);
                if (r->frameTotals == nullptr) r->frameTotals = (int64_t*) gs_device_alloc(sizeof(int64_t) * 2);
                if (r->layerAlive == nullptr)
                {
                    const size_t virtualSplats = (size_t) n * r->views;
                    r->layerAlive = (unsigned long long*) gs_device_alloc(sizeof(unsigned long long) * (virtualSplats / 64 + 2));
                    r->layerBlockCounts = (int32_t*) gs_device_alloc(sizeof(int32_t) * (virtualSplats / 1024 + 20));
                    r->layerBlockOffsets = (int32_t*) gs_device_alloc(sizeof(int32_t) * (virtualSplats / 1024 + 20));
                    r->layerScanTotal = (int64_t*) gs_device_alloc(sizeof(int64_t));
                    r->layerLiveCount = (int32_t*) gs_device_alloc(sizeof(int32_t));
                    r->layerLiveNibbles = (uint8_t*) gs_device_alloc(virtualSplats / 4 + 1024);
                    r->layerTileBox = (ushort4*) gs_device_alloc(sizeof(ushort4) * (virtualSplats + 1));
                }
            }
            for (int slot = 0; slot != 2; ++slot)
            {
                if (r->graphExec[slot]) (void) hipGraphExecDestroy(r->graphExec[slot]);
                if (r->graph[slot]) (void) hipGraphDestroy(r->graph[slot]);
                r->graphExec[slot] = nullptr;
                r->graph[slot] = nullptr;
                GS_CHECK(hipStreamBeginCapture(g.stream, hipStreamCaptureModeThreadLocal));
                enqueueCapturableFrame(r, scr, cull, lazy, slot);
                GS_CHECK(hipStreamEndCapture(g.stream, &r->graph[slot]));
                GS_CHECK(hipGraphInstantiate(&r->graphExec[slot], r->graph[slot], nullptr, nullptr, 0));
            }
            r->graphValid = true;
            r->graphCapacity = r->tileListCapacity;
            r->graphSplats = n;
            r->graphRowBegin = scr.rowBegin;
            r->graphRowEnd = scr.rowEnd;
            r->graphCull = (int) cull;
            r->graphFootprint = scr.epanechnikov;
            r->graphLazy = (int) lazy;
            r->graphAblation = g_ablation;
            r->graphFusion = (int) effectiveListFusion() | ((int) cfg.clipScanFusion << 1) | (effectivePixelsPerLane() << 2) | ((int) r->blockCull << 5) | (r->tileStartOrder << 6) | ((int) r->tileOrderManual << 7);
            r->graphLayers = r->depthLayers > 0 ? r->depthLayers : r->autoLayers;
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
        if (r->depthLayers == 0)
        {
            // The renderer chooses the layer count of the NEXT frame from this one's pairs (profiles/r05_depth_layers.txt).  A frame of one
            // layer knows the scene's pair total P; a layered one knows what it listed (the second host word), which is less.
            //   one layer:     two from P >= 16 M, three from 40 M (a LONE frame: two layers at 10.9 / 12.8 M pairs cost it 3 / 14 %, at 27 M
            //                  they save 16 %) -- and two from 4.5 M when the host has said that it keeps several frames in flight
            //                  (GS_DEPTH_LAYERS_CHOOSE_PIPELINED; BASELINE.json configs[2], 5.45 M: +4 % frames/s with four in flight, -14 %
            //                  for a lone frame: ten more launches on one stream; 10.9 M: +22 %; 3.7 M: break-even; 2.7 M: -13 %);
            //   first layered frame after a one-layer frame, P < 8 M: layers pay when most of the list lies behind the tiles' stop points --
            //                  listed / P was 0.32-0.38 where they paid and 0.66 on a scene of low opacities where they cost 10 % -- so above
            //                  0.5 the renderer goes back to one layer and does not try again for a while (256 frames, doubling);
            //   two layers:    one below 1.5 M listed (pipelined) / 3.5 M (lone): the scene has become small -- unless what is listed, scaled by the
            //                  listed / total this scene showed on its first layered frame, says that the TOTAL still lies above 0.8 of the
            //                  way up: the Epanechnikov run of the headline scene (4.76 M pairs, 1.51 M listed: share 0.32) stepped down at
            //                  1.49 M listed, up again at once on the total, and so on thirteen times per run on two of four renderers
            //                  (each a re-capture: 6 300-7 100 frames/s by luck) until the doubling hold had grown long enough;
            //                  three from 18 M listed;
            //   three:         two below 6 M listed.
            // (a band under its cull -- a rank of a multi-GPU split -- skips the 4.5-8 M zone and its trial: layered bands gain 8 % at 8-11 M
            // pairs per chain, and a trial that does not pay costs two re-captures)
            const int64_t upToTwo = r->autoPipelined ? (cull ? 8000000 : 4500000) : 16000000, downToOne = r->autoPipelined ? (cull ? 2500000 : 1500000) : 3500000;
            const int64_t listed = r->graphLayersActive ? r->scratch.totalHost[1] : demand;
            int next = r->autoLayers;
            bool atOnce = false; // (a failed trial is undone at once, whatever the hold says)
            if (r->autoBanFrames > 0) r->autoBanFrames -= 1;
            // (the first layered frame after a change: what share of the last known total this scene lists in layers)
            if (r->graphLayersActive && r->autoLayers > 1 && r->autoFramesSinceChange == 0 && r->autoLastTotal > 0 && listed > 0)
                r->autoListedShare = (float) std::min(1.0, (double) listed / (double) r->autoLastTotal);
            if (!r->graphLayersActive || r->autoLayers == 1)
            {
                r->autoLastTotal = demand;
                next = demand >= 40000000 ? 3 : (demand >= upToTwo && (demand >= 8000000 || r->autoBanFrames == 0)) ? 2 : 1;
            }
            else if (r->autoFramesSinceChange == 0 && r->autoLastTotal > 0 && r->autoLastTotal < 8000000 && 2 * listed > r->autoLastTotal)
            {
                next = 1;
                atOnce = true;
                r->autoBanFrames = r->autoBanLength;
                r->autoBanLength = std::min(r->autoBanLength * 2, 1 << 16);
            }
            else if (r->autoLayers == 2)
            {
                const bool totalStillUp = r->autoListedShare > 0.f && (double) listed / (double) r->autoListedShare >= 0.8 * (double) upToTwo;
                next = (listed < downToOne && !totalStillUp) ? 1 : listed >= 18000000 ? 3 : 2;
            }
            else next = listed < 6000000 ? 2 : r->autoLayers;
            // (a frame that cannot be layered -- band cull, one-pass tile sort -- keeps measuring totals; the wish costs nothing)
            // A change re-captures the frame (milliseconds), so a camera path whose totals straddle a threshold must not flap: steps up
            // are taken at once, steps down only after autoHoldFrames frames, and a step down that the next frames undo doubles the hold.
            r->autoFramesSinceChange += 1;
            if (next < r->autoLayers && !atOnce && r->autoFramesSinceChange < r->autoHoldFrames) next = r->autoLayers;
            if (next != r->autoLayers)
            {
                if (next > r->autoLayers && r->autoLayerChanges > 0 && r->autoFramesSinceChange <= 4) r->autoHoldFrames = std::min(r->autoHoldFrames * 2, 1 << 16);
                r->autoLayerChanges += 1;
                r->autoFramesSinceChange = 0;
            }
            r->autoLayers = next;
        }
        if (demand <= r->tileListCapacity) return 0;
        if (demand > kMaxTileListCapacity) return -2;
        GS_CHECK(hipStreamSynchronize(g.stream)); // the truncated frame drains before its buffers are replaced
        {
            std::lock_guard<std::mutex> lock(g_captureMutex);
            growTileListIfNeeded(r, demand);
        }
        r->graphValid = false;
        r->capturedRetries += 1;
    }
}
} // namespace gs

extern "C" void gs_renderer_set_graph(GsRenderer* r, int enabled)
{
    r->useGraph = enabled != 0;
    r->graphValid = false;
}

extern "C" int gs_renderer_captured_retries(const GsRenderer* r) { return r->capturedRetries; }
extern "C" int gs_renderer_depth_layer_changes(const GsRenderer* r) { return r->autoLayerChanges; }
// A/B runs: 0 = the later layers walk every splat of their slice of the depth order instead of the compacted live ones (same lists).
extern "C" void gs_renderer_debug_set_layer_compaction(GsRenderer* r, int enabled)
{
    r->layerCompaction = enabled != 0;
    r->graphValid = false;
}
// Experiment (profiles/r06_layer_compaction_bound.txt): what would a live-splat compaction that costs NOTHING buy a layered frame?
// mode 1: the later layers' live lists are written to buffers of their own (same kernels, same frame); mode 2: the three compaction kernels
// are not launched and the lists of the last mode-1 frame are walked again -- valid for the SAME camera only (the lists are that frame's).
extern "C" void gs_renderer_debug_set_layer_compaction_reuse(GsRenderer* r, int mode)
{
    StreamScope scope(r);
    GS_CHECK(hipStreamSynchronize(g.stream));
    r->layerReuse = (mode == 1 || mode == 2) ? mode : 0;
    if (r->layerReuse != 0 && r->layerLiveDebug == nullptr)
    {
        r->layerLiveDebug = (uint32_t*) gs_device_alloc(sizeof(uint32_t) * (size_t) kMaxDepthLayers * (size_t) r->splatCount * (size_t) r->views + 64);
        r->layerLiveCountDebug = (int32_t*) gs_device_alloc(sizeof(int32_t) * kMaxDepthLayers);
    }
    r->graphValid = false;
}
// A/B runs and tests: 0 = the band cull of a Morton-stored scene projects all N splats (round 5's kernel) instead of testing blocks first.
extern "C" void gs_renderer_debug_set_block_cull(GsRenderer* r, int enabled)
{
    r->blockCull = enabled != 0;
    r->graphValid = false;
}
// Blocks of 256 splats the last band-culled frame's block test kept, summed over its views (and the number of blocks x views in *total):
// what the replicated projection of a multi-GPU rank still walks.  Synchronises the renderer's stream.
extern "C" int64_t gs_renderer_debug_block_cull_kept(GsRenderer* r, int64_t* total)
{
    const int64_t blocks = ((int64_t) r->splatCount + 255) / 256 * r->views;
    if (total) *total = blocks;
    if (r->blockFlags == nullptr || blocks <= 0) return -1;
    StreamScope scope(r);
    GS_CHECK(hipStreamSynchronize(g.stream));
    std::vector<uint8_t> flags((size_t) blocks);
    GS_CHECK(hipMemcpyAsync(flags.data(), r->blockFlags, (size_t) blocks, hipMemcpyDeviceToHost, g.stream));
    GS_CHECK(hipStreamSynchronize(g.stream));
    int64_t kept = 0;
    for (uint8_t f : flags) kept += f ? 1 : 0;
    return kept;
}
// Splats the last band-culled frame kept (all views): the length of the culled list its depth sort ran over.  Synchronises the stream.
extern "C" int64_t gs_renderer_debug_band_kept_splats(GsRenderer* r)
{
    if (r->bandTotal == nullptr) return -1;
    StreamScope scope(r);
    GS_CHECK(hipStreamSynchronize(g.stream));
    int64_t kept = 0;
    GS_CHECK(hipMemcpyAsync(&kept, r->bandTotal, sizeof(int64_t), hipMemcpyDeviceToHost, g.stream));
    GS_CHECK(hipStreamSynchronize(g.stream));
    return kept;
}
// Depth layers the last captured frame was made of (1 when it was not layered: band cull, one-pass tile sort, a small scene, or the choice).
extern "C" int gs_renderer_depth_layers_in_use(const GsRenderer* r) { return (r->graphValid && r->graphLayersActive) ? r->graphLayers : 1; }
// Pairs the last captured frame listed, summed over its depth layers (with one layer: the frame's pair total).
extern "C" int64_t gs_renderer_listed_pairs(GsRenderer* r)
{
    StreamScope scope(r);
    GS_CHECK(hipStreamSynchronize(g.stream));
    if (r->graphValid && r->frameTotals != nullptr && r->graphLayersActive)
    {
        int64_t totals[2] = {0, 0};
        GS_CHECK(hipMemcpyAsync(totals, r->frameTotals, sizeof(totals), hipMemcpyDeviceToHost, g.stream)); // (never the legacy stream: see awaitPairTotal)
        GS_CHECK(hipStreamSynchronize(g.stream));
        return totals[1];
    }
    return r->scratch.totalHost != nullptr ? *r->scratch.totalHost : 0;
}
// One frame of `camera` as the captured layered frame's kernels, launched plainly on the renderer's stream with an event pair around
// every layer's raster launch (header).  The renderer must have rendered a layered captured frame before (buffers, capacity).
extern "C" int gs_renderer_debug_layered_raster_report(GsRenderer* r, const GsCameraData* camera, GsLayeredRasterReport* out)
{
    if (r == nullptr || camera == nullptr || out == nullptr) return -1;
    if (!r->useGraph || !r->graphValid || !r->graphLayersActive || r->views != 1 || r->paramsHost == nullptr) return -3;
    StreamScope scope(r);
    GS_CHECK(hipStreamSynchronize(g.stream));
    const Screen scr = makeScreen(r->width, r->height, r->rowBegin, r->rowEnd);
    const bool strictBand = scr.rowBegin > 0 || scr.rowEnd < scr.tilesY;
    const bool cull = r->splatCount > 0 && (r->bandCull == GS_BAND_CULL_ON || (r->bandCull == GS_BAND_CULL_AUTO && strictBand));
    FrameParams fp{};
    std::memcpy(fp.camera.view, camera->view, sizeof(fp.camera.view));
    fp.camera.fovCotX = camera->fovCotangent[0];
    fp.camera.fovCotY = camera->fovCotangent[1];
    fp.camera.depthScale = camera->depthScaleBias[0];
    fp.camera.depthBias = camera->depthScaleBias[1];
    for (int k = 0; k != 3; ++k) fp.cameraPosition[k] = r->lastCameraPosition[k] = camera->position[k];
    const int slot = r->graphSlot;
    r->paramsHost[slot * kMaxViews] = fp;
    LayerProbe probe;
    for (int l = 0; l != kMaxDepthLayers; ++l)
    {
        GS_CHECK(hipEventCreate(&probe.start[l]));
        GS_CHECK(hipEventCreate(&probe.stop[l]));
    }
    GS_CHECK(hipMalloc((void**) &probe.counters, sizeof(unsigned long long) * 4 * kMaxDepthLayers));
    GS_CHECK(hipMemsetAsync(probe.counters, 0, sizeof(unsigned long long) * 4 * kMaxDepthLayers, g.stream));
    // Two frames of the same camera: the first TIMED without counters (four atomics per tile on four addresses would be in the interval:
    // 141 instead of 48 us for the first layer's launch, measured), the second counted without being timed.
    unsigned long long* const counterBlock = probe.counters;
    probe.counters = nullptr;
    *r->scratch.totalHost = -1;
    enqueueCapturableFrame(r, scr, cull, useLazyColours(r), slot, &probe);
    GS_CHECK(hipStreamSynchronize(g.stream));
    *out = GsLayeredRasterReport{};
    out->layers = probe.layers;
    for (int l = 0; l != probe.layers; ++l) GS_CHECK(hipEventElapsedTime(&out->rasterMs[l], probe.start[l], probe.stop[l]));
    LayerProbe counting = probe; // (its events are recorded again: their times were read above)
    counting.counters = counterBlock;
    *r->scratch.totalHost = -1;
    enqueueCapturableFrame(r, scr, cull, useLazyColours(r), slot, &counting);
    GS_CHECK(hipStreamSynchronize(g.stream));
    probe.counters = counterBlock;
    unsigned long long counters[4 * kMaxDepthLayers] = {};
    GS_CHECK(hipMemcpyAsync(counters, probe.counters, sizeof(counters), hipMemcpyDeviceToHost, g.stream));
    GS_CHECK(hipStreamSynchronize(g.stream));
    for (int l = 0; l != probe.layers; ++l)
    {
        out->consumedPairs[l] = (int64_t) counters[4 * l];
        out->tilesRun[l] = (int64_t) counters[4 * l + 1];
        out->stateTileLoads[l] = (int64_t) counters[4 * l + 2];
        out->stateTileStores[l] = (int64_t) counters[4 * l + 3];
    }
    for (int l = 0; l != kMaxDepthLayers; ++l)
    {
        (void) hipEventDestroy(probe.start[l]);
        (void) hipEventDestroy(probe.stop[l]);
    }
    (void) hipFree(probe.counters);
    return probe.layers > 0 ? 0 : -3;
}
extern "C" void gs_debug_set_ablation(int mask) { g_ablation = mask; }
extern "C" void gs_debug_set_clip_scan_fusion(int enabled) { cfg.clipScanFusion = enabled != 0; }
extern "C" int gs_debug_get_clip_scan_fusion(void) { return cfg.clipScanFusion ? 1 : 0; }
extern "C" void gs_debug_set_await_timeout(double seconds) { g_awaitTimeoutSeconds = seconds > 0.0 ? seconds : 30.0; }
extern "C" long long gs_debug_host_times(double* launchMicroseconds, double* awaitMicroseconds, int reset)
{
    const long long frames = g_hostFrames.load();
    if (launchMicroseconds) *launchMicroseconds = frames ? 1e-3 * (double) g_hostLaunchNs.load() / (double) frames : 0.0;
    if (awaitMicroseconds) *awaitMicroseconds = frames ? 1e-3 * (double) g_hostAwaitNs.load() / (double) frames : 0.0;
    if (reset) g_hostLaunchNs = 0, g_hostAwaitNs = 0, g_hostFrames = 0;
    return frames;
}

extern "C" int gs_renderer_render_views(GsRenderer* r, const GsCameraData* cameras)
{
    if (r == nullptr || cameras == nullptr) return -1;
    if (r->views == 1) return gs_renderer_render(r, cameras, 0, nullptr);
    // every degree 1..4 has its colours evaluated by the raster kernel; only a process that keeps the SH pass (GS_NO_LAZY_SH, or a raster
    // shape other than two pixels per lane: diagnostics) would need one colour buffer per view
    const bool coloursPerView = r->shDegree != 0 && !useLazyColours(r);
    if (!r->useGraph || r->pipeline != GS_PIPELINE_TWO_LEVEL || cfg.sortBackend != GS_SORT_BACKEND_NATIVE || cfg.collectRasterStats ||
        r->splatCount <= 0 || coloursPerView || !radixSortSupported(4, r->splatCount * r->views))
        return -3;
    return renderCaptured(r, cameras);
}

