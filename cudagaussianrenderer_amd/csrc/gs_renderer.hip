// gs_renderer.hip -- the C ABI of include/gsplat_hip.h, part 2: the one-call renderer (Demo.cpp:314-331 buffers, :352-479 frame).
//
// Two pipelines produce the same frame:
//   GS_PIPELINE_REFERENCE  the reference's stage order through the eight staged calls above: pairs keyed by
//                          (tile << 32 | depth), one 64-bit radix sort over all pairs (44-47 key bits);
//   GS_PIPELINE_TWO_LEVEL  (default) sort the SPLATS by the 32-bit depth key first (N elements), emit the
//                          pairs in that order keyed by the tile index only, then one stable sort on the
//                          13-15 tile bits.  The resulting value list is identical (same total order, ties by
//                          splat index) while the radix-sort traffic drops from ~152 B to ~30 B per pair;
// The latter leaves the SH pass out when the raster kernel can evaluate colours on demand (useLazyColours below),
// keeps centre + ellipse in one 32-byte record per splat, and resets the range table from inside the clip kernel.
#include <algorithm>
#include <cstring>

#include "gs_state.h"
#include "gs_sh_eval.h"

using namespace gs;

namespace gs
{
// Colours on demand: the two-level frames skip the SH pass and let the raster kernel evaluate the colour of every
// list entry it stages ("Colours on demand").  Needs the [splat][row] coefficient copy (rows padded to 16-byte
// multiples: every degree 1-4); GS_NO_LAZY_SH=1 keeps the reference's stage order.
bool useLazyColours(const GsRenderer* r)
{
    static const bool disabled = getenv("GS_NO_LAZY_SH") != nullptr;
    StreamScope scope(r); // (the renderer's own pixels-per-lane setting decides with)
    return !disabled && r->shDegree != 0 && r->shPerSplat != nullptr && r->pipeline != GS_PIPELINE_REFERENCE &&
           rasterCanEvaluateColours(r->shDegree, r->shCount);
}
RasterColourSource colourSource(const GsRenderer* r, const float* cameraPosition, const FrameParams* paramsDevice)
{
    RasterColourSource src;
    src.degree = r->shDegree;
    src.shCount = r->shCount;
    src.rowStride = r->shRowStride;
    src.position = (const float4*) r->position;
    src.shPerSplat = r->shPerSplat;
    if (cameraPosition) src.cameraPosition = make_float3(cameraPosition[0], cameraPosition[1], cameraPosition[2]);
    src.params = paramsDevice;
    return src;
}

bool useBlockCull(const GsRenderer* r, BandBlockCull* cull)
{
    if (!r->blockCull || r->splatToStorage == nullptr || r->storageToSplat == nullptr || r->blockBounds == nullptr || r->blockFlags == nullptr) return false;
    cull->bounds = r->blockBounds;
    cull->flags = r->blockFlags;
    cull->storageToSplat = r->storageToSplat;
    cull->viewTerms = (BandViewTerms*) (r->blockFlags + r->blockFlagBytes); // (behind the flags, 16-byte aligned)
    return true;
}

namespace
{
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
void freeLayerCompaction(GsRenderer* r)
{
    // (sized by splats x views: allocated again with the next layered frame)
    gs_device_free(r->layerAlive);
    gs_device_free(r->layerBlockCounts);
    gs_device_free(r->layerBlockOffsets);
    gs_device_free(r->layerScanTotal);
    r->layerBlockOffsets = nullptr;
    r->layerScanTotal = nullptr;
    gs_device_free(r->layerLiveCount);
    gs_device_free(r->layerTileBox);
    gs_device_free(r->layerLiveNibbles);
    gs_device_free(r->layerLiveDebug);
    gs_device_free(r->layerLiveCountDebug);
    r->layerLiveDebug = nullptr;
    r->layerLiveCountDebug = nullptr;
    r->layerReuse = 0;
    r->layerLiveNibbles = nullptr;
    r->layerTileBox = nullptr;
    r->layerAlive = nullptr;
    r->layerBlockCounts = nullptr;
    r->layerLiveCount = nullptr;
}

void freeScene(GsRenderer* r)
{
    freeLayerCompaction(r);
    if (r->sceneShared)
    {
        // the read-only inputs are the owner's
        r->position = r->scaleAndRotation = r->sh = r->shPerSplat = nullptr;
        r->splatToStorage = nullptr;
        r->positionCaller = r->scaleAndRotationCaller = nullptr;
        r->blockBounds = nullptr;
        r->storageToSplat = nullptr;
        r->sceneShared = false;
    }
    gs_device_free(r->storageToSplat);
    r->storageToSplat = nullptr;
    gs_device_free(r->blockBounds);
    gs_device_free(r->blockFlags);

    r->blockBounds = nullptr;
    r->blockFlags = nullptr;

    gs_device_free(r->splatToStorage);
    gs_device_free(r->positionCaller);
    gs_device_free(r->scaleAndRotationCaller);
    r->splatToStorage = nullptr;
    r->positionCaller = r->scaleAndRotationCaller = nullptr;
    r->storageToSplatHost.clear();
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
} // namespace
bool growTileListIfNeeded(GsRenderer* r, int64_t demand)
{
    if (demand <= r->tileListCapacity) return false;
    int64_t cap = r->tileListCapacity > 0 ? r->tileListCapacity : 1;
    while (cap < demand) cap <<= 1; // Demo.cpp:360
    cap = std::min<int64_t>(cap, kMaxTileListCapacity);
    allocTileList(r, (int32_t) cap);
    return true;
}
} // namespace gs

namespace gs
{
// The range table of `views` views, with the depth layers' per-tile status bytes right behind it: one allocation, one 0xFF fill.
// (ranges, then one status byte per virtual tile, then one open bit per virtual tile in rows of openWordsPerRow(tilesX) words)
int openWordsPerRow(int tilesX) { return (tilesX + 31) / 32 + 1; } // (+1: a box that straddles a word boundary reads the word behind its first)
size_t rangeTableInts(const Screen& scr, int views)
{
    return 2 * (size_t) scr.totalTiles * views + ((size_t) scr.totalTiles * views + 3) / 4 + (size_t) scr.tilesY * views * openWordsPerRow(scr.tilesX);
}
static void allocRangeTable(GsRenderer* r, const Screen& scr)
{
    const size_t ints = rangeTableInts(scr, r->views);
    r->tileRange = (int32_t*) gs_device_alloc(sizeof(int32_t) * ints);
    GS_CHECK(hipMemset(r->tileRange, 0xFF, sizeof(int32_t) * ints));
    r->tileStatus = (uint8_t*) (r->tileRange + 2 * (size_t) scr.totalTiles * r->views);
    r->tileOpenBits = (uint32_t*) (r->tileRange + 2 * (size_t) scr.totalTiles * r->views + ((size_t) scr.totalTiles * r->views + 3) / 4);
}
} // namespace gs

extern "C" GsRenderer* gs_renderer_create(int width, int height)
{
    if (width > kMaxScreenSize || height > kMaxScreenSize)
    {
        std::fprintf(stderr, "gsplat_hip: gs_renderer_create(%d, %d): at most %d pixels per axis\n", width, height, kMaxScreenSize);
        return nullptr;
    }
    GsRenderer* r = new GsRenderer{};
    const Screen scr = makeScreen(width, height, 0, 0);
    r->width = scr.width;
    r->height = scr.height;
    allocRangeTable(r, scr);
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
    gs_device_free(r->tileOrder);
    gs_device_free(r->tileConsumed);
    gs_device_free(r->tileConsumedPair);
    gs_device_free(r->tileOrderState);
    gs_device_free(r->layerState);
    gs_device_free(r->layerConsumed);
    gs_device_free(r->frameTotals);
    gs_device_free(r->backBufferOwned);
    gs_device_free(r->sortTemp);
    freeTileListScratch(r->scratch);
    freeRadixScratch(r->radix);
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
        // ranges of tiles that leave the band must not linger
        StreamScope scope(r);
        GS_CHECK(hipStreamSynchronize(g.stream));
        const Screen scr = makeScreen(r->width, r->height, 0, 0);
        GS_CHECK(hipMemset(r->tileRange, 0xFF, sizeof(int32_t) * 2 * (size_t) scr.totalTiles));
    }
    r->rowBegin = begin;
    r->rowEnd = end;
}

extern "C" int gs_renderer_set_pipeline(GsRenderer* r, int pipeline)
{
    // The reference-shaped pipeline emits its pairs in storage order and breaks depth ties by it (one stable sort on tile | depth):
    // with a scene stored along the Morton curve its frames would differ from the caller-order frames wherever splats of equal depth
    // overlap.  A scene set while this pipeline is selected is stored in the caller's order whatever the storage option says; switching
    // to it with a Morton-stored scene in place is refused.
    if (pipeline == GS_PIPELINE_REFERENCE && r->splatToStorage != nullptr) return -1;
    r->pipeline = pipeline == GS_PIPELINE_REFERENCE ? pipeline : GS_PIPELINE_TWO_LEVEL;
    r->graphValid = false;
    return 0;
}

// Depth layers (header): 1..kMaxDepthLayers; fractions = ascending upper bounds of the first layers - 1 layers, or NULL for the defaults.
extern "C" int gs_renderer_set_depth_layers(GsRenderer* r, int layers, const float* fractions)
{
    static const float defaults[gs::kMaxDepthLayers + 1][gs::kMaxDepthLayers] = {
        {1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}, {0.15f, 1.f, 1.f, 1.f}, {0.06f, 0.25f, 1.f, 1.f}, {0.05f, 0.15f, 0.4f, 1.f}};
    if (layers < GS_DEPTH_LAYERS_CHOOSE_PIPELINED || layers > gs::kMaxDepthLayers) return -1;
    if (layers <= 0) // the renderer chooses (header): the cuts are the defaults of the count it picks
    {
        r->depthLayers = 0;
        r->autoPipelined = layers == GS_DEPTH_LAYERS_CHOOSE_PIPELINED;
        r->autoLayers = 1;
        r->autoFramesSinceChange = 0;
        r->autoHoldFrames = 16;
        r->autoBanFrames = 0;
        r->autoBanLength = 256;
        r->autoLastTotal = 0;
        r->autoListedShare = 0.f;
        r->graphValid = false;
        return 0;
    }
    float f[gs::kMaxDepthLayers];
    for (int l = 0; l != gs::kMaxDepthLayers; ++l) f[l] = defaults[layers][l];
    if (fractions != nullptr)
        for (int l = 0; l + 1 < layers; ++l)
        {
            if (!(fractions[l] > (l ? fractions[l - 1] : 0.0f)) || !(fractions[l] < 1.0f)) return -1;
            f[l] = fractions[l];
        }
    r->depthLayers = layers;
    for (int l = 0; l != gs::kMaxDepthLayers; ++l) r->layerFraction[l] = l + 1 < layers ? f[l] : 1.0f;
    r->graphValid = false;
    return 0;
}
extern "C" int gs_renderer_depth_layers(const GsRenderer* r) { return (r->depthLayers == 0 && r->autoPipelined) ? GS_DEPTH_LAYERS_CHOOSE_PIPELINED : r->depthLayers; }
namespace gs
{
// The layer count and cuts the next captured frame uses: as set, or -- depthLayers == 0 -- what the previous frames' pair totals suggest.
int effectiveDepthLayers(const GsRenderer* r, float fraction[kMaxDepthLayers])
{
    static const float defaults[kMaxDepthLayers + 1][kMaxDepthLayers] = {
        {1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}, {0.15f, 1.f, 1.f, 1.f}, {0.06f, 0.25f, 1.f, 1.f}, {0.05f, 0.15f, 0.4f, 1.f}};
    const int layers = r->depthLayers > 0 ? r->depthLayers : r->autoLayers;
    for (int l = 0; l != kMaxDepthLayers; ++l) fraction[l] = r->depthLayers > 0 ? r->layerFraction[l] : defaults[layers][l];
    return layers;
}
} // namespace gs

extern "C" int gs_renderer_colours_on_demand(const GsRenderer* r) { return useLazyColours(r) ? 1 : 0; }
extern "C" void gs_renderer_set_footprint(GsRenderer* r, int footprint)
{
    r->footprint = footprint == GS_FOOTPRINT_EPANECHNIKOV ? GS_FOOTPRINT_EPANECHNIKOV : footprint == GS_FOOTPRINT_GAUSSIAN ? GS_FOOTPRINT_GAUSSIAN : -1;
}
extern "C" int gs_renderer_footprint(const GsRenderer* r) { return r->footprint >= 0 ? r->footprint : g_footprint; }
namespace gs
{
static bool tileStartOrderWanted(const GsRenderer* r, int slots)
{
    return !r->tileOrderManual && r->tileStartOrder == GS_TILE_START_ORDER_PREVIOUS_FRAME && r->views == 1 && slots > 0;
}
void prepareTileStartOrder(GsRenderer* r, const Screen& scr)
{
    const int slots = rasterGridSlots(scr.tilesX, scr.rowEnd - scr.rowBegin);
    if (!tileStartOrderWanted(r, slots)) return;
    if (r->tileOrder != nullptr && r->tileOrderSlots == slots && r->tileConsumedPair != nullptr && r->tileOrderState != nullptr) return;
    // (the band changed, or the first frame with the option: everything starts from zero -- the table is not in force until two
    // consecutive frames resembled each other)
    std::lock_guard<std::mutex> lock(g_captureMutex);
    GS_CHECK(hipStreamSynchronize(g.stream));
    gs_device_free(r->tileOrder);
    gs_device_free(r->tileConsumedPair);
    gs_device_free(r->tileOrderState);
    const size_t tiles = (size_t) scr.totalTiles;
    r->tileOrder = (int32_t*) gs_device_alloc(sizeof(int32_t) * (size_t) slots);
    r->tileConsumedPair = (int32_t*) gs_device_alloc(sizeof(int32_t) * 2 * tiles);
    r->tileOrderState = (int32_t*) gs_device_alloc(sizeof(int32_t) * 16);
    GS_CHECK(hipMemsetAsync(r->tileOrder, 0xFF, sizeof(int32_t) * (size_t) slots, g.stream));
    GS_CHECK(hipMemsetAsync(r->tileConsumedPair, 0, sizeof(int32_t) * 2 * tiles, g.stream));
    GS_CHECK(hipMemsetAsync(r->tileOrderState, 0, sizeof(int32_t) * 16, g.stream));
    GS_CHECK(hipStreamSynchronize(g.stream));
    r->tileOrderSlots = slots;
    r->graphValid = false;
}
bool tileStartOrderSchedule(const GsRenderer* r, const Screen& scr, int parity, RasterSchedule* schedule)
{
    *schedule = RasterSchedule{};
    const int slots = rasterGridSlots(scr.tilesX, scr.rowEnd - scr.rowBegin);
    if (r->tileOrderManual)
    {
        // experiments: a host table, no order kernel; the per-tile counters only when the host armed them
        if (r->tileConsumedArmed) schedule->consumedOut = r->tileConsumed;
        if (r->tileOrder != nullptr && r->tileOrderSlots == slots) schedule->order = r->tileOrder;
        return false;
    }
    if (!tileStartOrderWanted(r, slots) || r->tileOrder == nullptr || r->tileOrderSlots != slots || r->tileConsumedPair == nullptr)
    {
        if (r->tileConsumedArmed) schedule->consumedOut = r->tileConsumed; // (experiments: the counters by themselves)
        return false;
    }
    schedule->order = r->tileOrder;
    schedule->consumedOut = r->tileConsumedPair + (size_t) (parity & 1) * (size_t) scr.totalTiles;
    schedule->enabled = r->tileOrderState;
    return true;
}
bool cameraJumped(GsRenderer* r, const float view[16])
{
    // the rotation part of the view matrix: a 1-degree step changes an entry by at most 0.0175, and beyond ~3 degrees the previous
    // frame's consumption predicts poorly (profiles/r06_tile_order.txt: -23 % raster time at 1 degree, -9 % at 5 with one view at +20 %)
    bool jumped = !r->tileOrderHaveLastView;
    for (int c = 0; c != 3 && !jumped; ++c)
        for (int k = 0; k != 3; ++k)
            if (!(fabsf(view[c * 4 + k] - r->tileOrderLastView[c * 4 + k]) <= 0.06f)) jumped = true;
    std::memcpy(r->tileOrderLastView, view, sizeof(r->tileOrderLastView));
    r->tileOrderHaveLastView = true;
    return jumped;
}
TileOrderArgs tileOrderArgsFor(const GsRenderer* r, const Screen& scr, int parity)
{
    const size_t tiles = (size_t) scr.totalTiles;
    TileOrderArgs a;
    a.now = r->tileConsumedPair + (size_t) ((parity & 1) ^ 1) * tiles; // the frame that just ended
    a.prev = r->tileConsumedPair + (size_t) (parity & 1) * tiles;      // the one before it; cleared: this frame's raster writes it
    a.order = r->tileOrder;
    a.state = r->tileOrderState;
    a.slots = r->tileOrderSlots;
    a.maxMismatch = r->tileOrderMaxMismatch;
    return a;
}
} // namespace gs

extern "C" int gs_renderer_set_tile_start_order(GsRenderer* r, int mode)
{
    if (mode != GS_TILE_START_ORDER_SCREEN && mode != GS_TILE_START_ORDER_PREVIOUS_FRAME) return -1;
    if (mode != r->tileStartOrder || r->tileOrderManual)
    {
        StreamScope scope(r);
        GS_CHECK(hipStreamSynchronize(g.stream));
        // (the next frame with the option starts its history afresh)
        gs_device_free(r->tileConsumedPair);
        r->tileConsumedPair = nullptr;
        r->tileOrderManual = false;
        r->graphValid = false;
    }
    r->tileStartOrder = mode;
    return 0;
}
extern "C" int gs_renderer_tile_start_order(const GsRenderer* r) { return r->tileStartOrder; }
extern "C" int gs_renderer_debug_tile_order_state(GsRenderer* r, int32_t out[4])
{
    for (int k = 0; k != 4; ++k) out[k] = 0;
    if (r->tileOrderState == nullptr) return -1;
    StreamScope scope(r);
    GS_CHECK(hipStreamSynchronize(g.stream));
    gs_device_download(out, r->tileOrderState, sizeof(int32_t) * 4);
    return 0;
}
extern "C" void gs_renderer_debug_set_tile_order_max_mismatch(GsRenderer* r, float maxMismatch)
{
    r->tileOrderMaxMismatch = maxMismatch > 0.0f ? maxMismatch : 0.35f;
    r->graphValid = false; // (a kernel argument of the captured frame)
}

// Experiment (tools/tile_order_experiment.py): the start order of the raster's tiles.
extern "C" int gs_renderer_debug_raster_grid_slots(const GsRenderer* r)
{
    const Screen scr = makeScreen(r->width, r->height, r->rowBegin, r->rowEnd);
    return rasterGridSlots(scr.tilesX, scr.rowEnd - scr.rowBegin);
}
extern "C" int gs_renderer_debug_set_tile_order(GsRenderer* r, const int32_t* slotToTileHost, int slots)
{
    StreamScope scope(r);
    GS_CHECK(hipStreamSynchronize(g.stream));
    gs_device_free(r->tileOrder);
    r->tileOrder = nullptr;
    r->tileOrderSlots = 0;
    r->tileOrderManual = slotToTileHost != nullptr && slots > 0; // (a host table switches the order kernel off until gs_renderer_set_tile_start_order is called again)
    r->graphValid = false;
    if (slotToTileHost == nullptr || slots <= 0) return 0;
    if (slots != gs_renderer_debug_raster_grid_slots(r)) return -1;
    r->tileOrder = (int32_t*) gs_device_alloc(sizeof(int32_t) * (size_t) slots);
    gs_device_upload(r->tileOrder, slotToTileHost, sizeof(int32_t) * (size_t) slots);
    r->tileOrderSlots = slots;
    return 0;
}
extern "C" int gs_renderer_debug_tile_consumed(GsRenderer* r, int32_t* perTileHost)
{
    StreamScope scope(r);
    GS_CHECK(hipStreamSynchronize(g.stream));
    const Screen scr = makeScreen(r->width, r->height, 0, 0);
    if (r->tileConsumed == nullptr)
    {
        r->tileConsumedArmed = true; // (the experiment's frames: per-tile counters; with the option on, the order kernel clears them after every frame)
        r->tileConsumed = (int32_t*) gs_device_alloc(sizeof(int32_t) * (size_t) scr.totalTiles);
        gs_device_memset(r->tileConsumed, 0, sizeof(int32_t) * (size_t) scr.totalTiles);
        GS_CHECK(hipStreamSynchronize(g.stream));
        return 1; // armed: the next plain frames leave their tiles' consumed entries
    }
    if (perTileHost != nullptr) gs_device_download(perTileHost, r->tileConsumed, sizeof(int32_t) * (size_t) scr.totalTiles);
    return 0;
}
extern "C" void gs_renderer_debug_set_list_fusion(GsRenderer* r, int enabled) { r->listFusion = enabled < 0 ? -1 : (enabled != 0 ? 1 : 0); }
extern "C" void gs_renderer_debug_set_raster_pixels_per_lane(GsRenderer* r, int ppl) { r->pixelsPerLane = (ppl == 1 || ppl == 2 || ppl == 4) ? ppl : 0; }

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
    if (r->splatToStorage != nullptr)
    {
        // block-level band cull (Morton storage): one flag per block of 256 splats and view, the keep mask in storage order
        r->blockFlagBytes = ((((size_t) n + 255) / 256) * (size_t) r->views + 15) / 16 * 16;
        r->blockFlags = (uint8_t*) gs_device_alloc(r->blockFlagBytes + sizeof(BandViewTerms) * kMaxViews); // (+ the views' terms of the bound)
    }
    // Demo.cpp:325: 8 pairs per splat to start with; grown on demand.
#ifndef GS_INITIAL_LIST_FACTOR
#define GS_INITIAL_LIST_FACTOR 8
#endif
    const int64_t initial = std::min<int64_t>((int64_t) std::max(n, 1) * GS_INITIAL_LIST_FACTOR * r->views, kMaxTileListCapacity);
    if (r->tileListCapacity < initial) allocTileList(r, (int32_t) initial);
}
} // namespace

extern "C" int gs_renderer_set_scene(GsRenderer* r, int32_t n, const float* position4, const float* scaleAndRotation4,
                                     const float* color4, const float* sh, int shDegree, int shCount)
{
    if (n < 0 || (shDegree != 0 && sh == nullptr) || (int64_t) n * r->views > kMaxVirtualSplats) return -1;
    StreamScope scope(r);
    GS_CHECK(hipStreamSynchronize(g.stream));
    freeScene(r);
    r->graphValid = false;
    r->splatCount = n;
    r->shDegree = shDegree;
    r->shCount = shCount;
    const size_t f4 = sizeof(float) * 4 * (size_t) n;
    // Storage order (gs_renderer_set_storage_order): the splats are uploaded in the caller's order, or along a Morton curve through
    // their positions -- neighbours in space become neighbours in memory, which is what the band cull of a multi-GPU rank, the
    // gathers of the tile-list passes and the raster kernel's L2 hit rate live on.  Frames are the same bytes either way: the
    // depth sort takes its input in the CALLER's order (its first pass makes the keys itself from a caller-ordered copy of the
    // positions, gs_radix.hip), so the stable sort breaks ties by the caller's index.
    // The reference-shaped pipeline emits its pairs in storage order and sorts them once on (tile | depth): it keeps the caller's order.
    std::vector<float> ordered;
    const auto inStorageOrder = [&](const float* src, size_t floatsPerSplat) -> const float* {
        if (r->storageToSplatHost.empty() || src == nullptr) return src;
        ordered.resize(floatsPerSplat * (size_t) n);
        for (int32_t k = 0; k != n; ++k)
            std::memcpy(ordered.data() + floatsPerSplat * (size_t) k, src + floatsPerSplat * (size_t) r->storageToSplatHost[(size_t) k],
                        sizeof(float) * floatsPerSplat);
        return ordered.data();
    };
    if (r->storageOrder == GS_STORAGE_ORDER_MORTON && r->pipeline != GS_PIPELINE_REFERENCE && n > 1)
    {
        r->storageToSplatHost.resize((size_t) n);
        gs_morton_order(position4, n, r->storageToSplatHost.data());
        // caller index -> (virtual) slot, once per view of a chain: the depth sort's input order
        std::vector<int32_t> inverse((size_t) n * (size_t) r->views);
        for (int v = 0; v != r->views; ++v)
            for (int32_t k = 0; k != n; ++k) inverse[(size_t) v * n + (size_t) r->storageToSplatHost[(size_t) k]] = v * n + k;
        r->splatToStorage = (int32_t*) gs_device_alloc(sizeof(int32_t) * inverse.size());
        gs_device_upload(r->splatToStorage, inverse.data(), sizeof(int32_t) * inverse.size());
        // ... and the positions in the caller's order, from which its first pass makes the keys (gs_radix.hip, KEYGEN)
        r->positionCaller = (float*) gs_device_alloc(f4);
        gs_device_upload(r->positionCaller, position4, f4);
        // ... and the band-cull kernel of a multi-GPU rank walks the splats in the caller's order (gs_clip.hip)
        r->scaleAndRotationCaller = (float*) gs_device_alloc(f4);
        gs_device_upload(r->scaleAndRotationCaller, scaleAndRotation4, f4);
    }
    r->position = (float*) gs_device_alloc(f4);
    r->scaleAndRotation = (float*) gs_device_alloc(f4);
    allocateSceneIntermediates(r, n);
    gs_device_upload(r->position, inStorageOrder(position4, 4), f4);
    gs_device_upload(r->scaleAndRotation, inStorageOrder(scaleAndRotation4, 4), f4);
    gs_device_upload(r->color, inStorageOrder(color4, 4), f4);
    if (r->splatToStorage != nullptr && n > 0)
    {
        // block-level band cull: centre box and covariance bound of every 256 Morton-neighbouring splats (gs_clip.hip), once per scene
        r->storageToSplat = (int32_t*) gs_device_alloc(sizeof(int32_t) * (size_t) n); // (where a kept splat's bit goes in the caller-ordered keep mask)
        gs_device_upload(r->storageToSplat, r->storageToSplatHost.data(), sizeof(int32_t) * (size_t) n);
        r->blockBounds = (BlockBound*) gs_device_alloc(sizeof(BlockBound) * (((size_t) n + 255) / 256));
        launchBlockBounds(g.stream, n, (const float4*) r->position, (const float4*) r->scaleAndRotation, r->blockBounds);
        GS_CHECK(hipStreamSynchronize(g.stream));
    }
    if (shDegree != 0)
    {
        // Demo.cpp:291-296
        const float* rows = inStorageOrder(sh, (size_t) shCount);
        const size_t total = gs_realigned_sh_size(GS_SH_GROUP_SIZE, shCount, n);
        std::vector<float> aligned(total);
        gs_realign_spherical_harmonics(rows, aligned.data(), GS_SH_GROUP_SIZE, shCount, n);
        r->sh = (float*) gs_device_alloc(total * sizeof(float));
        gs_device_upload(r->sh, aligned.data(), total * sizeof(float));
        // second copy in the file's [splat][row] layout for colours on demand (the raster kernel gathers a consumed entry's row) and the
        // band-culled SH evaluation; rows padded to a 16-byte multiple (degrees 2 and 4: 27 -> 28, 75 -> 76 floats) so that every degree
        // the reference ingests (PlyParser.cpp:204-243) is read as whole float4s
        r->shRowStride = sheval::rowStrideFloats(shCount);
        const size_t stride = (size_t) r->shRowStride;
        r->shPerSplat = (float*) gs_device_alloc(sizeof(float) * (size_t) n * stride);
        if (stride == (size_t) shCount) gs_device_upload(r->shPerSplat, rows, sizeof(float) * (size_t) n * stride);
        else
        {
            std::vector<float> padded((size_t) n * stride, 0.0f);
            for (int32_t k = 0; k != n; ++k) std::memcpy(padded.data() + stride * (size_t) k, rows + (size_t) shCount * (size_t) k, sizeof(float) * (size_t) shCount);
            gs_device_upload(r->shPerSplat, padded.data(), sizeof(float) * padded.size());
        }
    }
    return 0;
}

extern "C" void gs_renderer_set_storage_order(GsRenderer* r, int order)
{
    r->storageOrder = order == GS_STORAGE_ORDER_MORTON ? GS_STORAGE_ORDER_MORTON : GS_STORAGE_ORDER_CALLER;
}

extern "C" int32_t gs_renderer_storage_order(const GsRenderer* r, int32_t* storageToSplat)
{
    if (storageToSplat != nullptr)
        for (int32_t k = 0; k != r->splatCount; ++k) storageToSplat[k] = r->storageToSplatHost.empty() ? k : r->storageToSplatHost[(size_t) k];
    return r->splatCount;
}

// Several renderers of one scene (frames in flight): the read-only inputs -- position, scale/rotation, both coefficient layouts --
// exist once, in the owner; the sharer allocates only its own intermediates and lists (0.3 instead of 0.75 GB at C3) and copies the
// static colours device to device.  The owner must outlive its sharers and keep its scene.
extern "C" int gs_renderer_share_scene(GsRenderer* r, const GsRenderer* owner)
{
    if (r == nullptr || owner == nullptr || r == owner || owner->sceneShared) return -1;
    // Every refusal comes before anything of r is touched: a sharer that fails keeps the scene it had (the order list of a
    // Morton-stored scene is laid out per view, so the two renderers must agree on the views of a chain).
    if (owner->splatToStorage != nullptr && owner->views != r->views) return -1;
    if ((int64_t) owner->splatCount * r->views > kMaxVirtualSplats) return -1;
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
    r->shRowStride = owner->shRowStride;
    r->storageOrder = owner->storageOrder;
    r->splatToStorage = owner->splatToStorage;
    r->positionCaller = owner->positionCaller;
    r->scaleAndRotationCaller = owner->scaleAndRotationCaller;
    r->blockBounds = owner->blockBounds;
    r->storageToSplat = owner->storageToSplat;
    r->storageToSplatHost = owner->storageToSplatHost;
    r->sceneShared = true;
    if (r->splatToStorage != nullptr) r->pipeline = GS_PIPELINE_TWO_LEVEL; // (see gs_renderer_set_pipeline)
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
    allocRangeTable(r, scr);
    gs_device_free(r->layerState); // (sized per view: allocated again with the next layered frame)
    gs_device_free(r->layerConsumed);
    r->layerState = nullptr;
    r->layerConsumed = nullptr;
    freeLayerCompaction(r);
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
        // the two-level frames keep ellipses, centres and conics only inside the packed records: unpack them for the caller
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
        if (r->graphLayersActive)
        {
            // depth layers: the list in the buffers is the LAST layer's (the host word holds the largest layer's total)
            int64_t lastLayer = 0;
            GS_CHECK(hipMemcpyAsync(&lastLayer, r->scratch.total, sizeof(int64_t), hipMemcpyDeviceToHost, g.stream));
            GS_CHECK(hipStreamSynchronize(g.stream));
            r->tileListSize = (int32_t) std::min<int64_t>(lastLayer, r->tileListCapacity);
        }
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
    const bool twoLevel = r->pipeline != GS_PIPELINE_REFERENCE;

    GS_CHECK(hipEventRecord(r->frameStart, g.stream));

    // Clears (Demo.cpp:399,403). Only this GPU's band of the back buffer; the whole range table.
    const size_t rowBytes = (size_t) r->width * 4;
    const int pixelRowBegin = scr.rowBegin * kTileSize;
    const int pixelRowEnd = std::min(scr.rowEnd * kTileSize, r->height);
    // (two-level pipeline: the raster kernel zeroes the tiles that have no work itself, no 8 MB memset)
    if (pixelRowEnd > pixelRowBegin && !twoLevel)
        gs_device_memset(r->backBuffer + rowBytes * pixelRowBegin, 0, rowBytes * (size_t) (pixelRowEnd - pixelRowBegin));
    // (two-level pipeline: the clip kernel resets the range table on the side)
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
        if (demand > kMaxTileListCapacity) return -2;
        if (growTileListIfNeeded(r, demand)) st.grewTileList = 1;
        const int32_t size = (int32_t) demand;
        launchTileListEmit(g.stream, n, scr, (const float2*) r->clipXY, r->clipZ, (const float4*) r->ellipse, r->scratch, r->keys[0],
                           r->values[0], r->tileListCapacity);
        gs_timer_stop(stageTimers[2]);
        if (blocking) st.msBuildTileList = gs_timer_elapsed_ms(stageTimers[2]);

        GsDoubleBufferU64 keys{{r->keys[0], r->keys[1]}, 0};
        GsDoubleBufferI32 values{{r->values[0], r->values[1]}, 0};
        sortTileListImpl(stageTimers[3], size, &r->sortTemp, &r->sortTempBytes, &keys, &values, false); // Demo.cpp:458 (the list above is in emission order: no look first)
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
        ClipWithScan deferredClip;
        // (Morton storage on the native sort: no keys from the clip kernel -- the depth sort's first pass makes them in the caller's order)
        const bool keysFromSort = !cull && r->splatToStorage != nullptr && cfg.sortBackend == GS_SORT_BACKEND_NATIVE && radixSortSupported(4, n);
        // gs_renderer_set_tile_start_order: this frame's tile start order is made at its start from the previous frame's consumption -- by the
        // clip kernel's first eight workgroups (plain, or riding in the sort's scan launches), or (band cull) by a kernel of its own
        prepareTileStartOrder(r, scr);
        RasterSchedule scheduleProbe;
        f.tileOrderParity = r->tileOrderParity;
        f.tileOrderActive = tileStartOrderSchedule(r, scr, f.tileOrderParity, &scheduleProbe);
        TileOrderArgs orderArgs = f.tileOrderActive ? tileOrderArgsFor(r, scr, f.tileOrderParity) : TileOrderArgs{};
        if (f.tileOrderActive) orderArgs.skip = cameraJumped(r, camera->view) ? 1 : 0;
        bool orderBuilt = !f.tileOrderActive;
        const auto launchClip = [&]() { // the clip kernel of a frame without band cull, by itself
            launchSplatClipData(g.stream, n, scr, currentCamera(), (const float4*) r->position, (const float4*) r->scaleAndRotation, nullptr, nullptr, nullptr,
                                nullptr, keysFromSort ? nullptr : r->depthKey[0], r->splatToStorage != nullptr ? nullptr : (uint32_t*) r->order[0], nullptr,
                                nullptr, nullptr, nullptr, r->tileRange, 2 * scr.totalTiles, (float4*) r->tileRecord, (float4*) r->rasterRecord, 1, nullptr, nullptr, nullptr,
                                orderBuilt ? nullptr : &orderArgs);
            orderBuilt = true;
        };
        if (!orderBuilt && (cull || n <= 0))
        {
            launchTileStartOrder(g.stream, scr, orderArgs);
            orderBuilt = true;
        }
        bool clipDeferred = false;
        if (!cull)
        {
            // The sort makes its own keys: the clip kernel shares the launch of the sort's first column scan (gs_internal.h
            // RadixScanCompanion) -- unless the stages are timed one by one (the clip stage then runs, and is timed, by itself).
            if (keysFromSort && cfg.clipScanFusion && !timed && n > 0)
            {
                deferredClip.count = n;
                deferredClip.scr = scr;
                deferredClip.camera = currentCamera();
                deferredClip.position = (const float4*) r->position;
                deferredClip.scaleAndRotation = (const float4*) r->scaleAndRotation;
                deferredClip.rangeTableToClear = r->tileRange;
                deferredClip.rangeEntries = 2 * scr.totalTiles;
                deferredClip.tileRecord = (float4*) r->tileRecord;
                deferredClip.rasterRecord = (float4*) r->rasterRecord;
                deferredClip.parts = kClipScanParts;
                if (!orderBuilt) deferredClip.tileOrder = orderArgs;
                orderBuilt = true;
                clipDeferred = true;
                r->radix.firstScanCompanion = RadixScanCompanion{&clipWithScanCompanion, &deferredClip, kClipScanParts};
            }
            else launchClip();
        }
        else
        {
            // clip data for every splat + keep-mask, then an order-preserving compaction of (depth key, index)
            // (Morton storage: in the caller's order, from the caller-ordered copies of the inputs)
            // (... unless the cull runs by blocks of Morton neighbours: storage order, only the blocks that can reach the band, gs_clip.hip)
            BandBlockCull blocks;
            const bool byBlocks = useBlockCull(r, &blocks);
            const bool callerOrder = r->splatToStorage != nullptr && !byBlocks;
            const ClipCamera bandCamera = currentCamera();
            launchSplatClipData(g.stream, n, scr, bandCamera, (const float4*) (callerOrder ? r->positionCaller : r->position),
                                (const float4*) (callerOrder ? r->scaleAndRotationCaller : r->scaleAndRotation), nullptr, byBlocks ? nullptr : r->clipZ, nullptr, nullptr,
                                nullptr, nullptr, r->bandMask, r->bandSlotPrefix, r->bandCounts, nullptr, r->tileRange, 2 * scr.totalTiles,
                                (float4*) r->tileRecord, (float4*) r->rasterRecord, 1, callerOrder ? r->splatToStorage : nullptr, nullptr,
                                byBlocks ? &blocks : nullptr);
            launchBandCompaction(g.stream, n, 1, r->bandMask, r->bandSlotPrefix, r->bandCounts, r->bandOffsets, r->bandTotal, r->clipZ, r->depthKey[0],
                                 (uint32_t*) r->order[0], r->splatToStorage, byBlocks ? (const float4*) r->positionCaller : nullptr, &bandCamera, nullptr);
            countDevice = (const int*) r->bandTotal; // low 32 bits of the little-endian int64 total
        }
        gs_timer_stop(stageTimers[1]);
        if (blocking) st.msClipData = gs_timer_elapsed_ms(stageTimers[1]);
        if (cull && r->shDegree != 0 && !f.lazyColours)
        {
            // SH colour of the kept splats only (list = the compacted, still index-ordered order[0])
            gs_timer_start(stageTimers[0]);
            const float* cp = camera->position;
            launchSphericalHarmonicsGather(g.stream, n, countDevice, r->shDegree, r->shRowStride, r->order[0], (const float4*) r->position,
                                           r->shPerSplat, make_float3(cp[0], cp[1], cp[2]), (float4*) r->color);
            gs_timer_stop(stageTimers[0]);
            if (blocking) st.msSphericalHarmonics = gs_timer_elapsed_ms(stageTimers[0]);
        }
        r->radix.countDevice = countDevice;
        r->scratch.countDevice = countDevice;

        // (1) splats by depth: 32-bit keys, N elements
        int orderSel = 0;
        gs_timer_start(stageTimers[6]);
        const bool native = cfg.sortBackend == GS_SORT_BACKEND_NATIVE;
        if (n > 0)
        {
            // Morton storage, no cull: the sort's input is in the caller's order (the culled list was compacted in it already)
            const int32_t* inputOrder = cull ? nullptr : r->splatToStorage;
            if (native && radixSortSupported(4, n))
            {
                DepthKeySource keys;
                keys.positionCaller = (const float4*) r->positionCaller;
                keys.camera = currentCamera();
                keys.splatsPerView = n;
                orderSel = launchRadixSortPairs(g.stream, r->radix, 4, r->depthKey[0], r->depthKey[1], r->order[0], r->order[1], n, 0, kDepthKeyBits,
                                                inputOrder, inputOrder != nullptr ? &keys : nullptr);
            }
            else
            {
                int inputSel = 0;
                if (inputOrder != nullptr)
                {
                    launchGatherSortInput(g.stream, n, r->depthKey[0], inputOrder, r->depthKey[1], r->order[1]);
                    inputSel = 1;
                }
                ensureSortTemp(r, sortTempBytes(4, n, 0, kDepthKeyBits));
                orderSel = launchSortPairs(g.stream, r->sortTemp, r->sortTempBytes, 4, r->depthKey[0], r->depthKey[1], r->order[0],
                                           r->order[1], inputSel, n, 0, kDepthKeyBits);
            }
        }
        if (clipDeferred)
        {
            // what the sort did not take along (rows of more than 4096 tiles: nothing of it; fewer scans than slices): by itself, then
            r->radix.firstScanCompanion = RadixScanCompanion{};
            launchSplatClipDataRest(g.stream, deferredClip);
        }
        gs_timer_stop(stageTimers[6]);
        f.orderSel = orderSel;
        const uint32_t* order = (const uint32_t*) r->order[orderSel];

        // (2) pairs in depth order, keyed by tile. The emit pass is launched with the current capacity while the
        // pair count travels to the host, so the readback (GaussianRender.cu:793) costs no GPU idle time.
        gs_timer_start(stageTimers[2]);
        // (gs_set_list_fusion: the emit pass carries the first pass of the tile sort, finish() sorts on the remaining bits)
        f.sortBits = (effectiveListFusion() && native) ? tileListFusedSortBits(tileBits(scr.totalTiles), f.tileKeyBytes) : 0;
        launchTileListFastCount(g.stream, n, scr, order, (const float2*) r->clipXY, (const float4*) r->ellipse, r->scratch, f.sortBits);
        launchTileListFastEmit(g.stream, n, scr, order, (const float2*) r->clipXY, (const float4*) r->ellipse, r->scratch, r->keys[0],
                               f.tileKeyBytes, r->values[0], r->tileListCapacity, 0, f.sortBits);
        GS_CHECK(hipEventRecord(r->countReady, g.stream)); // the emit pass leaves the pair total
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

    if (!f.listDone)
    {
        const int tBits = tileBits(scr.totalTiles);
        const uint32_t* order = (const uint32_t*) r->order[f.orderSel];
        GS_CHECK(hipEventSynchronize(r->countReady));
        const int64_t demand = *r->scratch.totalHost;
        g.lastDemand = demand;
        if (demand > kMaxTileListCapacity) return -2;
        if (growTileListIfNeeded(r, demand))
        {
            // the list did not fit: re-emit into the larger buffers (rare; the first frames of a scene)
            st.grewTileList = 1;
            launchTileListFastEmit(g.stream, n, scr, order, (const float2*) r->clipXY, (const float4*) r->ellipse, r->scratch,
                                   r->keys[0], f.tileKeyBytes, r->values[0], r->tileListCapacity, 0, f.sortBits);
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
            {
                r->radix.rangeTableForLastPass = r->tileRange; // the last pass leaves the tile ranges: no range kernel (gs_radix.hip)
                sel = launchRadixSortPairs(g.stream, r->radix, f.tileKeyBytes, r->keys[0], r->keys[1], r->values[0], r->values[1], size,
                                           f.sortBits, tBits);
            }
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
        if (r->radix.rangeTableForLastPass == nullptr) launchTileRanges(g.stream, size, f.tileKeyBytes, r->keys[sel], r->tileRange); // (rocPRIM backend)
        r->radix.rangeTableForLastPass = nullptr;
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
    RasterSchedule schedule;
    if (f.tileOrderActive) (void) tileStartOrderSchedule(r, scr, f.tileOrderParity, &schedule); // (begin() made the table; this frame's counters: its parity's buffer)
    else (void) tileStartOrderSchedule(r, scr, 0, &schedule);                                    // (experiments' manual table / counters, or nothing)
    launchRasterize(g.stream, scr, r->tileRange, r->values[r->selector], (const float2*) r->clipXY, (const float4*) r->conic,
                    (const float4*) r->color, (uchar4*) r->backBuffer, consumed, r->pipeline != GS_PIPELINE_REFERENCE,
                    f.lazyColours ? &lazySource : nullptr, r->pipeline != GS_PIPELINE_REFERENCE ? (const float4*) r->rasterRecord : nullptr, 1, 0, nullptr,
                    &schedule);
    r->colourStale = f.lazyColours;
    gs_timer_stop(stageTimers[5]);
    if (f.tileOrderActive) r->tileOrderParity ^= 1;
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


// the fused convenience -- scene in, image out.
extern "C" int gs_render(const GsScene* scene, const GsCameraData* camera, int width, int height, uint8_t* rgbaOutHostOrNull)
{
    if (scene == nullptr || camera == nullptr || width <= 0 || height <= 0 || scene->splatCount < 0) return -1;
    GsRenderer* r = gs_renderer_create(width, height);
    if (r == nullptr) return -1; // (more than kMaxScreenSize pixels per axis)
    int rc = gs_renderer_set_scene(r, scene->splatCount, scene->position, scene->scaleAndRotation, scene->color, scene->sphericalHarmonics,
                                   scene->sphericalHarmonicsDegree, scene->sphericalHarmonicsCount);
    if (rc == 0) rc = gs_renderer_render(r, camera, 0, nullptr);
    if (rc == 0 && rgbaOutHostOrNull != nullptr) gs_renderer_read_back_buffer(r, rgbaOutHostOrNull);
    gs_renderer_destroy(r);
    return rc;
}
