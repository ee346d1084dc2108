// gs_state.h -- host-side state shared by the translation units behind the C ABI: gs_capi.hip (library state, timers, device
// buffers, the eight staged calls), gs_renderer.hip (the one-call renderer) and gs_capture.hip (captured frames).  Not installed.
#pragma once

#include <mutex>
#include <vector>

#include "gs_internal.h"

// Utilities.h:155-187
struct GsTimer
{
    hipEvent_t start, stop;
    float elapsedMs;
    int blocking;
};

namespace gs
{

void setRasterPixelsPerLane(int ppl);
int rasterPixelsPerLane();
void setRasterTrace(unsigned long long* deviceBufferOrNull);

extern int g_footprint; // process-wide, like the reference's #define: the staged calls' footprint and every renderer's default
extern int g_ablation;  // diagnostics: stages left out of captured frames (gs_debug_set_ablation)

// The reference keeps these in __constant__ symbols (GaussianRender.cu:42-43); here they are a host
// copy and every launch passes what its kernel needs by value.
// Scratch of the staged buildTileList that emits its pairs already sorted (gs_staged.hip): depth codes and order of the splats, the
// library's halves of the (tile index, value) ping-pong of the tile sort -- the other value half is the caller's list --, the sort's
// tables for the pairs, and the flag of sortTileList's is-it-sorted-already check.
struct StagedScratch
{
    uint32_t* depthKey[2] = {nullptr, nullptr};
    int32_t* order[2] = {nullptr, nullptr};
    int capacitySplats = 0;
    void* tileKeys[2] = {nullptr, nullptr};
    int32_t* values = nullptr;
    uint32_t* depth[2] = {nullptr, nullptr}; // the pairs' quantised depths, carried through the tile sort beside the values
    int64_t capacityPairBytes = 0; // bytes of each tileKeys half; values and depth[] hold capacityPairs entries
    int capacityPairs = 0;
    gs::RadixScratch radixPairs;
    int* unsortedEpoch = nullptr;     // device word: the epoch of the last check that found a descent
    int* unsortedEpochHost = nullptr; // pinned host word the check kernel writes its verdict to ...
    int* unsortedEpochHostMapped = nullptr; // ... through this device address
    int epoch = 0;
};

// gs_set_staged_tile_start_order: what the staged rasterizeTile / evaluateSplatClipData pair of THIS host thread keeps between frames -- the
// one-call renderer's tile start order (gs_renderer_set_tile_start_order) for the literal drop-in host, whose loop is the interactive
// viewer (Demo.cpp:371-479).
struct StagedTileOrder
{
    int32_t* consumedPair = nullptr; // [2][tiles]
    int32_t* order = nullptr;        // [slots]
    int32_t* state = nullptr;        // [16]
    int slots = 0, tiles = 0, firstTile = -1, rows = 0;
    int parity = 0;
    float lastView[16] = {};
    bool haveLastView = false;
    bool builtThisFrame = false; // evaluateSplatClipData made a table for the frame rasterizeTile is about to draw
};

struct State
{
    GsGlobalArgs global{};
    GsTileListArgs tileList{};
    hipStream_t stream = nullptr;
    gs::TileListScratch scratch;
    gs::RadixScratch radix;
    StagedScratch staged;
    StagedTileOrder stagedTileOrder;
    int64_t lastDemand = 0;
    // What the renderer whose call is running on this thread wants instead of the process-wide defaults (set and restored by
    // StreamScope; -1 / 0 = the default): two threads that drive renderers with different footprints share no switch.
    int footprint = -1;     // GS_FOOTPRINT_*, gs_renderer_set_footprint
    int listFusion = -1;    // gs_renderer_debug_set_list_fusion
    int pixelsPerLane = 0;  // gs_renderer_debug_set_raster_pixels_per_lane
};
// One state per host thread: like the reference, the staged calls are not re-entrant, but each host thread may drive
// its own renderer / staged sequence on its own stream (bench.py runs one thread per frame in flight).
extern thread_local State g;

#ifndef GS_CLIP_SCAN_FUSION_DEFAULT
#define GS_CLIP_SCAN_FUSION_DEFAULT 0 // off: +3 % for one frame at a time, -1 % with four frames in flight (profiles/r04_ab_experiments.txt); A/B builds: 1
#endif
// Process-wide settings.
struct Settings
{
    int sortBackend = GS_SORT_BACKEND_NATIVE;
    bool collectRasterStats = false;
    int stagedBuilder = GS_STAGED_BUILDER_SORTED; // gs_set_staged_builder: how the staged buildTileList emits (and whether sortTileList looks before it sorts)
    int stagedTileStartOrder = GS_TILE_START_ORDER_SCREEN; // gs_set_staged_tile_start_order
    bool listFusion = GS_LIST_FUSION_DEFAULT != 0; // the emit pass carries the tile sort's first pass (gs_set_list_fusion)
    bool clipScanFusion = GS_CLIP_SCAN_FUSION_DEFAULT != 0; // Morton storage, no band cull: the clip kernel shares the launch of the depth sort's first column scan (gs_debug_set_clip_scan_fusion)
    unsigned long long* consumedDevice = nullptr; // counter of the staged gs_rasterize_tile
};
extern Settings cfg;
// The settings in force for the call running on this thread: the renderer's own where it has one (StreamScope), else the process-wide default.
inline int effectiveFootprint() { return g.footprint >= 0 ? g.footprint : g_footprint; }
inline bool effectiveListFusion() { return g.listFusion >= 0 ? g.listFusion != 0 : cfg.listFusion; }
inline int effectivePixelsPerLane() { return g.pixelsPerLane != 0 ? g.pixelsPerLane : rasterPixelsPerLane(); }

Screen currentScreen();
ClipCamera currentCamera();
void reportLaunch(const char* stage);
// gs_staged.hip: the staged buildTileList / sortTileList on the renderer's machinery (see there)
int32_t stagedBuildTileListSorted(GsTimer* timer, int32_t tileListCapacity);
void freeStagedTileOrder(StagedTileOrder& t);
bool stagedListIsSorted(int32_t size, const uint64_t* keys, int endBit);
// sortTileList proper; lookFirst: skip the sort when the list is already in order (a stable sort of a sorted list is the identity)
void sortTileListImpl(GsTimer* timer, int32_t size, void** deviceTempStorage, size_t* tempStorageSizeInBytes, GsDoubleBufferU64* keys,
                      GsDoubleBufferI32* values, bool lookFirst);

constexpr int kStageTimers = 7; // sh, clip, tile list, pair sort, ranges, raster, depth sort (added to the sort total)
// What begin() leaves for finish().
struct FrameState
{
    Screen scr{};
    GsTimer* stageTimers[kStageTimers] = {};
    bool blocking = false;
    bool listDone = false; // reference pipeline: the sorted list and ranges were produced in begin()
    int orderSel = 0, tileKeyBytes = 2;
    int sortBits = 0; // low tile bits already sorted by the emit pass (gs_tiles_fast.hip), 0 = none
    bool lazyColours = false; // colours are evaluated by the raster kernel, entry by entry (no SH pass this frame)
    bool tileOrderActive = false; // gs_renderer_set_tile_start_order: begin() made this frame's table; finish()'s raster uses it
    int tileOrderParity = 0;
    float cameraPosition[3] = {0.f, 0.f, 0.f};
    int32_t size = 0;
    GsFrameStats st{};
};

} // namespace gs

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
    bool sceneShared = false; // position, scaleAndRotation, sh, shPerSplat, splatToStorage, positionCaller, scaleAndRotationCaller belong to another renderer (gs_renderer_share_scene)
    // gs_renderer_set_storage_order: the splats stored along a Morton curve instead of in the caller's order.  storageToSplat[k] = the
    // caller's index of the splat in slot k (host copy, for gs_renderer_storage_order); splatToStorage its inverse on the device, once
    // per view (entry v * N + c = v * N + slot of caller c): the INPUT ORDER of the depth sort (gs_radix.hip: a stable sort leaves
    // equal depth codes in input order = the caller's) and the order the band cull compacts in; null = the caller's order.
    // positionCaller / scaleAndRotationCaller: the projection's inputs once more in the caller's order -- the depth sort's first
    // pass makes its keys from the positions, the band-cull kernel walks both.
    int storageOrder = GS_STORAGE_ORDER_CALLER;
    int32_t* splatToStorage = nullptr;
    float* positionCaller = nullptr;
    float* scaleAndRotationCaller = nullptr;
    std::vector<int32_t> storageToSplatHost;
    uint32_t* depthKey[2] = {nullptr, nullptr};
    int32_t* order[2] = {nullptr, nullptr};
    // band cull (multi-GPU): per-splat coefficients in the [splat][shCount] layout for gathered SH evaluation,
    // one keep-mask and count per 64 splats, their scan, and the kept-splat count (device scalar)
    float* shPerSplat = nullptr;
    int shRowStride = 0; // floats between its rows: shCount rounded up to a multiple of four (degrees 2 and 4: one pad float per row)
    // per-renderer stream and scratch, so that several renderers can have frames in flight at once
    hipStream_t stream = nullptr;
    bool hasStream = false;
    gs::TileListScratch scratch;
    gs::RadixScratch radix;
    unsigned long long* consumed = nullptr;
    // captured frame (hipGraph): the whole two-level frame as one graph launch, list sizes kept on the device
    bool useGraph = false, graphValid = false, ownsStream = false;
    // TWO instances of the captured frame, each reading the camera from its own pinned slot: a frame can be launched while the
    // previous one of this renderer is still running (its slot is not touched), so the stream never runs dry between frames
    hipGraph_t graph[2] = {nullptr, nullptr};
    hipGraphExec_t graphExec[2] = {nullptr, nullptr};
    int graphSlot = 0; // instance of the NEXT launch
    int views = 1;     // cameras per captured chain (gs_renderer_set_views)
    // Depth layers (gs_renderer_set_depth_layers): the depth-sorted splats are listed, sorted and blended in `depthLayers` layers
    // (layer l = sorted splats [layerFraction[l-1], layerFraction[l]) of the frame's), later layers without the tiles earlier ones stopped.
    int depthLayers = 1;  // 1..4 = as set (default 1: the reference's one list per frame), 0 = chosen from the previous frame's pair total (autoLayers)
    int autoLayers = 1;
    bool autoPipelined = false;     // GS_DEPTH_LAYERS_CHOOSE_PIPELINED: the host keeps several frames in flight on the GPU -- two layers from 4.5 M pairs
    unsigned long long* layerAlive = nullptr; // depth layers: one bit per (virtual) splat -- an open tile under its box (launchLayerCompaction)
    int32_t* layerBlockCounts = nullptr;
    int32_t* layerBlockOffsets = nullptr;     // (slices of more than 4 M splats: the scanned counts)
    int64_t* layerScanTotal = nullptr;
    int32_t* layerLiveCount = nullptr;
    uint8_t* layerLiveNibbles = nullptr;
    ushort4* layerTileBox = nullptr;          // the list builder's box per (virtual) splat, written by the clip kernel of a layered frame
    bool layerCompaction = true;
    // experiment (gs_renderer_debug_set_layer_compaction_reuse): 1 = the later layers' live lists go to buffers of their own, 2 = the
    // compaction kernels are NOT launched and the lists the last mode-1 frame left are walked again (same camera: the same lists) --
    // the frame a compaction fused into the count pass at zero cost would be
    int layerReuse = 0;
    uint32_t* layerLiveDebug = nullptr; // [kMaxDepthLayers][splats x views]
    int32_t* layerLiveCountDebug = nullptr; // [kMaxDepthLayers]
    int autoLayerChanges = 0;       // times the choice changed (each one re-captures the frame)
    int autoFramesSinceChange = 0;
    int64_t autoLastTotal = 0;      // the pair total of the last one-layer frame
    float autoListedShare = 0.f;    // listed / total of the first layered frame after it (0: not known): what a layered frame's listed pairs say about its total
    int autoBanFrames = 0;          // frames left in which a scene below 8 M pairs is not tried with layers again (its listed / total was > 0.5)
    int autoBanLength = 256;
    int autoHoldFrames = 16;        // frames a choice is kept before it may go DOWN; doubles when a step down is undone at once
    float layerFraction[gs::kMaxDepthLayers] = {1.f, 1.f, 1.f, 1.f}; // upper bounds, ascending; the last used one is 1
    uint8_t* tileStatus = nullptr;      // inside the range table's allocation, right behind the ranges (the same 0xFF fill resets both)
    uint32_t* tileOpenBits = nullptr;   // behind the status bytes: one bit per (virtual) tile, 1 = open (the fill sets them all)
    float4* layerState = nullptr;       // per pixel and view, allocated with the first layered frame
    int32_t* layerConsumed = nullptr;   // per (virtual) tile
    int64_t* frameTotals = nullptr;     // device: {largest, summed} layer pair totals of the frame
    int graphLayers = 0;
    bool graphLayersActive = false; // the captured frame really is layered (not with a band cull, a one-pass tile sort, a tiny scene)
    gs::FrameParams* paramsDevice = nullptr;
    gs::FrameParams* paramsHost = nullptr; // pinned, two slots
    int graphCapacity = 0, graphSplats = 0, graphRowBegin = -1, graphRowEnd = -1, graphCull = -1, graphFootprint = -1, graphLazy = -1, graphAblation = 0, graphSelector = 0, graphKeyBytes = 2, graphFusion = -1;
    bool pending = false; // begin() issued, finish() not yet
    int capturedRetries = 0; // captured frames rendered again because their pair demand exceeded the list capacity
    gs::FrameState frame;
    unsigned long long* bandMask = nullptr;
    int32_t *bandCounts = nullptr, *bandOffsets = nullptr, *bandSlotPrefix = nullptr;
    int64_t* bandTotal = nullptr;
    int bandCull = GS_BAND_CULL_AUTO;
    // renderer state that the reference has as compile-time or process-wide switches (-1 / 0: the process-wide default at frame time)
    int footprint = -1;    // gs_renderer_set_footprint
    int listFusion = -1;   // gs_renderer_debug_set_list_fusion
    int pixelsPerLane = 0; // gs_renderer_debug_set_raster_pixels_per_lane
    // gs_renderer_set_tile_start_order: the raster's tiles started longest-first by the previous frame's consumption (gs_raster.hip
    // tileStartOrderKernel); tileOrderManual: a table the host handed in instead (gs_renderer_debug_set_tile_order, experiments)
    int tileStartOrder = GS_TILE_START_ORDER_SCREEN;
    float tileOrderMaxMismatch = 0.35f;
    bool tileOrderManual = false;
    float tileOrderLastView[16] = {}; // the previous frame's view matrix (cameraJumped)
    bool tileOrderHaveLastView = false;
    bool tileConsumedArmed = false; // gs_renderer_debug_tile_consumed: the raster leaves its per-tile counters also when no order kernel runs
    int32_t* tileOrder = nullptr;
    int tileOrderSlots = 0;
    int32_t* tileConsumed = nullptr;     // per tile: entries consumed in the frame being rendered (experiments: the one buffer)
    int32_t* tileConsumedPair = nullptr; // the option's two buffers [2][tiles]: a frame writes the one of its parity, the other holds the previous frame's
    int tileOrderParity = 0;             // parity of the next PLAIN frame (captured frames: the graph instance's slot)
    int32_t* tileOrderState = nullptr;   // device: [0] table in force, [1] sum |now - previous|, [2] sum now, [3] frames seen
    // block-level band cull (gs_clip.hip; Morton storage only): the per-block bounds made at set_scene (the scene owner's when shared),
    // this renderer's per-frame block flags and storage-ordered keep masks
    gs::BlockBound* blockBounds = nullptr;
    uint8_t* blockFlags = nullptr;
    size_t blockFlagBytes = 0; // (the views' BandViewTerms follow the flags in the same allocation)
    int32_t* storageToSplat = nullptr; // device copy of storageToSplatHost (the scene owner's when shared)
    bool blockCull = true; // gs_renderer_debug_set_block_cull (A/B and tests: 0 = every rank projects all N splats, round 5's band kernel)
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

namespace gs
{
// The launchers enqueue on g.stream; a renderer that was given its own stream borrows the slot for the call, so that
// several renderers (frames in flight) can be driven from one host thread on different streams.
// ... and so it is with the renderer's own footprint, list-fusion and pixels-per-lane settings: in force for the calls made inside the scope.
struct StreamScope
{
    hipStream_t saved;
    int savedFootprint, savedListFusion, savedPixelsPerLane;
    explicit StreamScope(const GsRenderer* r) : saved(g.stream), savedFootprint(g.footprint), savedListFusion(g.listFusion), savedPixelsPerLane(g.pixelsPerLane)
    {
        if (r->hasStream) g.stream = r->stream;
        g.footprint = r->footprint;
        g.listFusion = r->listFusion;
        g.pixelsPerLane = r->pixelsPerLane;
    }
    ~StreamScope()
    {
        g.stream = saved;
        g.footprint = savedFootprint;
        g.listFusion = savedListFusion;
        g.pixelsPerLane = savedPixelsPerLane;
    }
};

// ---- helpers of the renderer shared with the captured path (gs_renderer.hip) ----
bool useLazyColours(const GsRenderer* r);
RasterColourSource colourSource(const GsRenderer* r, const float* cameraPosition, const FrameParams* paramsDevice);
// The longest pair list a renderer makes room for: a megabyte short of 2^31 entries, so that every "count + tile - 1" and every workgroup's
// "first element of my tile" of the list kernels -- grids are sized for the capacity, rounded up to whole tiles and eight XCDs -- stays an int.
// A frame that needs more returns -2 (the reference's `tileListCapacity *= 2`, Demo.cpp:360, overflows its int from 2^30 pairs on).
constexpr int64_t kMaxTileListCapacity = 0x7FF00000;
// What else a renderer refuses instead of letting an int wrap: more than 65 536 pixels per axis (4 096 tiles: the list builder packs a
// box's tile coordinates into 16 bits each, and tilesX * tilesY * views stays far below 2^31), and more than 2^30 splats x views per chain
// (sorted indices, the two-records-per-splat addressing and n * views are ints; 2^30 splats are 64 GB of positions and shapes alone).
constexpr int kMaxScreenSize = 65536;
constexpr int64_t kMaxVirtualSplats = 1ll << 30;
bool growTileListIfNeeded(GsRenderer* r, int64_t demand);
// The band cull of this renderer's next frame runs by blocks of Morton-neighbouring splats (fills `cull` for launchSplatClipData).
bool useBlockCull(const GsRenderer* r, BandBlockCull* cull);
int effectiveDepthLayers(const GsRenderer* r, float fraction[kMaxDepthLayers]);
size_t rangeTableInts(const Screen& scr, int views); // ranges of all views + the depth layers' status bytes and open bits behind them, in ints
int openWordsPerRow(int tilesX);
// gs_renderer_set_tile_start_order: prepareTileStartOrder makes the buffers the option needs (never inside a capture; takes the capture
// mutex itself); tileStartOrderSchedule fills the raster launches' start-order arguments for this renderer's next frame and says whether
// the order kernel follows the frame's last raster launch.
void prepareTileStartOrder(GsRenderer* r, const Screen& scr);
bool tileStartOrderSchedule(const GsRenderer* r, const Screen& scr, int parity, RasterSchedule* schedule);
// (what the builder at the START of the frame of this parity works on: the previous frame's counters -- the other buffer -- as `now`)
TileOrderArgs tileOrderArgsFor(const GsRenderer* r, const Screen& scr, int parity);
// Did the camera turn by more than ~3 degrees since this renderer's previous frame (or is there none)?  Remembers `view` for the next call.
bool cameraJumped(GsRenderer* r, const float view[16]);
// Captured path (gs_capture.hip).  `cameras`: r->views of them (one frame per view, rendered by one chain).
int renderCaptured(GsRenderer* r, const GsCameraData* cameras);
// Allocation / legacy-stream work of one host thread must not overlap another thread's stream capture: set-up and capture
// are serialised process-wide (they happen once per renderer; replays take no lock).
extern std::mutex g_captureMutex;

} // namespace gs
