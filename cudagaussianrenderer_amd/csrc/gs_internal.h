// gs_internal.h -- declarations shared by the HIP translation units of libgsplat_hip.so.
// Not installed; the public surface is include/gsplat_hip.h.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "gsplat_hip.h"
#include "gsplat_hip_debug.h"

// src/CudaErrorCheck.cpp:2-11 -- print and exit on a failed runtime call.
#define GS_CHECK(call)                                                                       \
    do                                                                                       \
    {                                                                                        \
        hipError_t gsErr_ = (call);                                                          \
        if (gsErr_ != hipSuccess)                                                            \
        {                                                                                    \
            printf("HIP error at %s %d: %s\n", __FILE__, __LINE__, hipGetErrorString(gsErr_)); \
            exit(EXIT_FAILURE);                                                              \
        }                                                                                    \
    } while (0)

namespace gs
{

constexpr int kTileSize = GS_TILE_SIZE;
constexpr int kWave = 64;        // CDNA wavefront
constexpr int kRasterChunk = 32; // GaussianRender.cu:439,950,995: termination is voted every 32 list entries
// Views per frame chain (gs_renderer_set_views): V cameras of ONE scene rendered by one chain of kernels.  View v's splat i is the
// "virtual splat" v*N + i (own clip data, shared inputs) and its tile t the "virtual tile" v*totalTiles + t, so ONE depth sort and
// ONE tile sort order the pairs of all views and every kernel of the chain runs once per V frames.
constexpr int kMaxViews = 4;
constexpr int kMaxDepthLayers = 4; // gs_renderer_set_depth_layers

// Everything the reference derives from the compile-time k_ScreenSize (Consts.h:4-8), per frame here.
// Passed to kernels by value (kernarg segment -> SGPRs) instead of a __constant__ symbol.
struct Screen
{
    int width, height;
    int tilesX, tilesY, totalTiles;
    int rowBegin, rowEnd;                   // tile-row band of this GPU
    float tilesPerScreenX, tilesPerScreenY; // un-rounded W/16, H/16 (GaussianRender.cu:537)
    float tileClipSizeX, tileClipSizeY;     // GaussianRender.cu:656-657
    float pixelClipSizeX, pixelClipSizeY;   // GaussianRender.cu:939
    float traceBump;                        // GaussianRender.cu:271-274
    float extentFactor;                     // 3 sigma, or sqrt(7) with the Epanechnikov footprint (GaussianRender.cu:298-302)
    int epanechnikov;                       // GaussianRender.cu:8-9 as a run-time switch
};

Screen makeScreen(int width, int height, int rowBegin, int rowEnd);

// Camera fields the kernels read (the two projection matrices of CameraData are never used on device).
struct ClipCamera
{
    float view[16];
    float fovCotX, fovCotY;
    float depthScale, depthBias;
};

// Per-frame parameters in device memory: with a captured frame (hipGraph) the kernel arguments are frozen, so the
// camera is read through a pointer that a memcpy node refreshes at the head of every replay.
struct FrameParams
{
    ClipCamera camera;
    float cameraPosition[3];
    float pad; // (view 0's: non-zero = the camera jumped since this renderer's previous frame, gs_renderer_set_tile_start_order)
};

// Block-level band cull (gs_clip.hip): per block of 256 splats in storage order, the box of the centres and a bound on the largest
// eigenvalue of the splats' world-space covariances -- what a rank of a multi-GPU split needs to prove that a whole block of
// Morton-neighbouring splats cannot reach its tile rows without projecting any of them.
struct BlockBound
{
    float cmin[3];
    float lamMax;
    float cmax[3];
    float pad;
};
struct BandViewTerms // what the bound needs of a camera besides its matrix (gs_clip.hip makeBandViewTerms), per view and frame
{
    float wn, zvLo, zvHi, sane;
};
struct BandBlockCull
{
    const BlockBound* bounds = nullptr;          // [ceil(N / 256)], made at set_scene (launchBlockBounds)
    uint8_t* flags = nullptr;                    // [views][ceil(N / 256)]: 1 = the block may reach the band this frame
    const int32_t* storageToSplat = nullptr;     // storage slot -> caller's index: a kept splat sets its bit in the CALLER-ordered keep mask
    BandViewTerms* viewTerms = nullptr;          // [kMaxViews], left by the block test for the band kernel's per-splat test
};
void launchBlockBounds(hipStream_t stream, int count, const float4* position, const float4* scaleAndRotation, BlockBound* bounds);

// The start order of a frame's raster tiles from the PREVIOUS frame's consumption (gs_renderer_set_tile_start_order, gs_tile_order.h): made
// at the start of the frame -- by the first eight workgroups of the clip kernel (launchSplatClipData: tileOrderOrNull) or by a small kernel
// of its own.  state: [0] the table is in force for this frame, [1] sum |now - prev| over the tiles, [2] sum now, [3] frames seen, [4..8]
// the workgroups' partial sums and arrival count.
struct TileOrderArgs
{
    const int32_t* now = nullptr; // per tile: entries consumed in the frame that just ended
    int32_t* prev = nullptr;      // ... in the frame before it; cleared here: the frame that starts writes its own counters into it
    int32_t* order = nullptr;     // [slots]: grid slot -> tile, -1 = none
    int32_t* state = nullptr;
    int slots = 0;
    float maxMismatch = 0.35f;
    int skip = 0;                          // the host saw the camera jump: clear the counters, withdraw the table, build nothing
    const float* skipFlagDevice = nullptr; // ... the same from device memory (captured frames: FrameParams::pad of view 0, refreshed per replay)
};
void launchTileStartOrder(hipStream_t stream, const Screen& scr, const TileOrderArgs& args);

// ---- launchers (one per stage); all enqueue on `stream` and return without synchronising --------
void launchSphericalHarmonics(hipStream_t stream, int count, int degree, int shCount, const float4* position, const float* sh,
                              float3 cameraPosition, float4* color, const FrameParams* paramsDeviceOrNull = nullptr);

void launchSplatClipData(hipStream_t stream, int count, const Screen& scr, const ClipCamera& cam, const float4* position,
                         const float4* scaleAndRotation, float2* clipXY, float* clipZ, float4* screenEllipse, float4* conic,
                         uint32_t* depthKeyOrNull, uint32_t* splatOrderOrNull /* identity list beside the depth keys; may be null */,
                         unsigned long long* bandMaskOrNull = nullptr,
                         int32_t* bandSlotPrefixOrNull = nullptr, int32_t* bandBlockCountsOrNull = nullptr,
                         const FrameParams* paramsDeviceOrNull = nullptr,
                         int32_t* rangeTableToClearOrNull = nullptr, int rangeEntries = 0, float4* tileRecordOrNull = nullptr,
                         float4* rasterRecordOrNull = nullptr, int views = 1, const int32_t* bandOutSlotOrNull = nullptr,
                         ushort4* tileBoxOrNull = nullptr /* depth layers: the list builder's box per splat (gs_aabb.h splatTileBox) */,
                         // band cull of a Morton-stored scene by blocks (gs_clip.hip): position / scaleAndRotation are then the STORAGE-ordered
                         // arrays, bandOutSlot is null, clipZ is not written (launchBandCompaction makes the depth keys from positionCaller)
                         const BandBlockCull* blockCullOrNull = nullptr,
                         // gs_renderer_set_tile_start_order: the plain kernel (no band cull) builds this frame's tile start order with its first
                         // eight workgroups; ignored by the band kernels (the caller launches launchTileStartOrder itself)
                         const TileOrderArgs* tileOrderOrNull = nullptr);
// screenEllipse[i] = (cos, sin, extent x, extent y) from the 32-byte tile records the clip kernel can write instead.
void launchUnpackEllipse(hipStream_t stream, int count, const float4* tileRecord, float4* screenEllipse);
// positionClipSpaceXY / Z / conic from the 32-byte raster records (cx, cy, conic.x, conic.y | conic.z, depth, 0, 0) the clip kernel can
// write instead of the three arrays (what the raster gathers per list entry is then ONE memory sector instead of two).
void launchUnpackRasterRecord(hipStream_t stream, int count, const float4* rasterRecord, float2* clipXY, float* clipZ, float4* conic);
// Band cull (multi-GPU): the clip kernel flags the splats whose tile AABB reaches this GPU's rows (per 64 splats one 64-bit mask
// and the kept splats before them in their block of 256, one count per block); this scans the block counts and writes the kept
// splats' depth keys and (virtual) indices, order preserved, and their number to *bandTotal.  Morton storage: the band kernel was given
// the caller-ordered inputs and bandOutSlot = splatToStorage (caller's index -> storage slot); masks and clip depths are then indexed
// by the caller's index and the compaction runs in the CALLER's order (gs_clip.hip).
void launchBandCompaction(hipStream_t stream, int count, int views, const unsigned long long* bandMask, const int32_t* bandSlotPrefix,
                          const int32_t* bandBlockCounts, int32_t* bandBlockOffsets, int64_t* bandTotal, const float* clipZ,
                          uint32_t* depthKey, uint32_t* splatOrder, const int32_t* splatToStorageOrNull = nullptr,
                          // block cull: no clip depths were stored -- the keys are made from the caller-ordered positions and the camera(s)
                          const float4* positionCallerOrNull = nullptr, const ClipCamera* cameraOrNull = nullptr, const FrameParams* paramsDeviceOrNull = nullptr);
// Exclusive scan of n int32 on the device (one workgroup), total as int64.
void launchScanInt32(hipStream_t stream, int n, const int32_t* in, int32_t* out, int64_t* total);
// SH colour for the splats of a device-side index list only; coefficients in the [splat][rowStride] layout (rows of shCount floats,
// padded to a 16-byte multiple by the renderer: gs_sh_eval.h rowStrideFloats).
void launchSphericalHarmonicsGather(hipStream_t stream, int countHost, const int* countDevice, int degree, int rowStride,
                                    const int32_t* indices, const float4* position, const float* shPerSplat, float3 cameraPosition,
                                    float4* color, const FrameParams* paramsDeviceOrNull = nullptr);

// Scratch of the tile-list builder: per-wave pair counts, their exclusive scan, the total.
struct TileListScratch
{
    int32_t* waveCounts = nullptr;  // [ceil(N/64)]
    int32_t* waveOffsets = nullptr; // [ceil(N/64)]
    int64_t* total = nullptr;       // device scalar
    int64_t* totalHost = nullptr;   // pinned host mirror
    int64_t* totalHostMapped = nullptr; // device address of the same word (written directly by the renderer's scan)
    int32_t capacityWaves = 0;
    // renderer's builder: hit masks of the count pass (one byte per lane and round, the first rounds of every group of
    // 256 splats), replayed by the emit pass instead of repeating the overlap tests
    uint8_t* hitMasks = nullptr;
    void* boxRecords = nullptr; // int4 per depth-sorted splat: the tile box the count pass found, for the emit pass
    int32_t capacityMaskGroups = 0;
    // optional: (clip x, y, cos, sin | extent x, y, 0, 0) per splat as one 32-byte record (written by the clip kernel);
    // when set, the renderer's builders read it instead of clipXY + screenEllipse
    const float4* tileRecord = nullptr;
    const int* countDevice = nullptr; // optional device-side splat count (renderer, culled list); nullptr = host count
    // emit pass that carries the first pass of the tile sort (gs_tiles_fast.hip, sortBits > 0): the count pass's digit histogram per
    // group of 256 splats as table[digit][group], its scan over the groups, the digit totals
    int32_t* digitTable = nullptr;
    int32_t* digitScanned = nullptr;
    int32_t* digitTotals = nullptr;
    int digitStride = 0; // groups per digit row (multiple of 4)
    // the staged buildTileList (gs_staged.hip), set per call: the depth-SORTED 27-bit codes of the splats (parallel to `order`) and where
    // the emit pass leaves every pair's 32-bit quantised depth (gs_tiles_fast.hip CARRY); null otherwise
    const uint32_t* depthSorted = nullptr;
    uint32_t* carryOut = nullptr;
    // depth layers (gs_renderer_set_depth_layers), set per layer by the frame driver: the raster's open-tile bits (pairs of tiles whose
    // bit is clear are not listed), the frame's {largest, summed} layer totals on the device, 1 = first layer | 2 = last layer
    const uint32_t* openBits = nullptr; // one bit per (virtual) tile, 1 = still open; rows of openWordsPerRow words (RasterLayer)
    int openWordsPerRow = 0;
    int64_t* frameTotals = nullptr;
    int layerFlags = 0;
};
void ensureTileListScratch(TileListScratch& s, int count);
void freeTileListScratch(TileListScratch& s);
// Pass 1 + scan: leaves the demand (pairs with unlimited capacity) in *s.total (async copy to s.totalHost enqueued).
void launchTileListCount(hipStream_t stream, int count, const Screen& scr, const float2* clipXY, const float4* screenEllipse,
                         TileListScratch& s);
// Pass 2: writes at most `capacity` pairs, in (splat, local tile) order.
void launchTileListEmit(hipStream_t stream, int count, const Screen& scr, const float2* clipXY, const float* clipZ,
                        const float4* screenEllipse, const TileListScratch& s, uint64_t* keys, int32_t* values, int capacity);

// Sizes everything the renderer's builder needs for `count` splats (call before capturing a frame: no allocation may happen
// inside a capture).
void ensureTileListFastScratch(TileListScratch& s, int count);
// Renderer-only builder (gs_tiles_fast.hip): splats visited in `order` (depth-sorted), key = tile index (2 or 4 bytes).
// Count leaves one pair total per group of 256 splats; Emit sums the totals in front of each group itself and leaves the pair
// total of the frame in *s.total (+ the host-mapped word): the demand is known to the host only after Emit.
// sortBits > 0 (tileListFusedSortBits): the emit pass carries the FIRST PASS of the tile sort -- the count pass also counts the low
// `sortBits` bits of its pairs' tile indices per group (table[digit][group]), a column scan turns that into output positions, and the
// emit pass stages its pairs in LDS, ranks them stably by that digit and writes them where the first scatter would have put them
// (no histogram kernel, no scatter kernel, the unsorted list is never written or read).  The caller then sorts on bits
// [sortBits, tileBits) only.  Both passes must be given the same sortBits and viewSplats.
// depth layers: liveOrder[0 .. *liveCount) = the splats of order[0 .. count) that still have an open tile under their box, in order
void launchLayerCompaction(hipStream_t stream, const Screen& scr, int splats, int viewSplats, const ushort4* tileBox, const uint32_t* openBits, int openWordsPerRow, unsigned long long* aliveWords,
                           const uint32_t* order, int count, int32_t* blockCounts, uint8_t* liveNibbles /* [ceil(count / 4)] */, uint32_t* liveOrder,
                           int32_t* liveCount, int32_t* blockOffsets /* [as blockCounts] */, int64_t* scanTotal,
                           // a list whose length only the device knows (a band's culled list): `order` / `count` are the WHOLE list and its
                           // capacity, the slice of layer `layer` of `layers` is cut on the device from *listCountDevice by the cuts (null: the
                           // host cut it); openBits == null: a first layer, every splat of the slice is live
                           const int* listCountDevice = nullptr, const float* cuts = nullptr, int layer = 0, int layers = 1);
void launchTileListFastCount(hipStream_t stream, int count, const Screen& scr, const uint32_t* order, const float2* clipXY,
                             const float4* screenEllipse, TileListScratch& s, int sortBits = 0, int viewSplats = 0);
void launchTileListFastEmit(hipStream_t stream, int count, const Screen& scr, const uint32_t* order, const float2* clipXY,
                            const float4* screenEllipse, const TileListScratch& s, void* tileKeys, int tileKeyBytes, int32_t* values,
                            int capacity, int viewSplats = 0, int sortBits = 0);
// Digit width of that first pass for a tile sort over `tileBitCount` bits of `tileKeyBytes`-byte keys, 0 when the sort is not a
// two-pass sort of 16-bit keys with a 6..8-bit first digit (then the plain emit pass and the full sort run).
int tileListFusedSortBits(int tileBitCount, int tileKeyBytes);

// rocPRIM radix sort of (key, int32 value) pairs on key bits [beginBit, endBit); keyBytes in {2, 4, 8}.
size_t sortTempBytes(int keyBytes, int size, int beginBit, int endBit);
// Returns the selector (0/1) of the buffer holding the sorted data.
int launchSortPairs(hipStream_t stream, void* temp, size_t tempBytes, int keyBytes, void* keys0, void* keys1, int32_t* values0,
                    int32_t* values1, int selector, int size, int beginBit, int endBit);

// A column scan of the renderer's radix sort is a latency-bound kernel of 256-512 small workgroups between a histogram and a scatter.
// A kernel of the caller's that does not depend on the sort may SHARE THE LAUNCH of the sort's first scan (one kernel, one dependent
// launch and one chip-wide kernel boundary less per frame): the caller leaves a companion in RadixScratch, and the next
// launchRadixSortPairs hands it the arguments of its first scan instead of launching the scan kernel; the companion must launch ONE
// kernel of 256-lane workgroups whose first `rows` workgroups (of grid row 0) call radixColumnScanRow<256> (gs_radix_scan.h).  `shots`
// scans in a row are offered to it (round 6 measured the clip kernel in slices, one per scan of the depth sort: kClipScanParts); the sort
// clears the companion with its last shot.  A sort that cannot take it (rows of more than 4096 tiles) or runs out of passes leaves it set, and the
// caller launches what is left of its kernel itself.
struct RadixScanArgs
{
    int rows; // digit rows = workgroups of the scan
    int nHost;
    const int* nDevice;
    int tileElements, stride;
    const int32_t* table;
    int32_t* scanned;
    int32_t* totals;
};
struct RadixScanCompanion
{
    void (*launch)(hipStream_t stream, const RadixScanArgs& scan, void* user) = nullptr;
    void* user = nullptr;
    int shots = 1;
};
// The renderer's own radix sort (gs_radix.hip): 2- or 4-byte keys, int32 values, stable, bits [beginBit, endBit).
struct RadixScratch
{
    int32_t* table = nullptr;       // [256][tiles] digit histogram
    int32_t* scanned = nullptr;     // chunk-local exclusive scan of the table
    int32_t* chunkTotals = nullptr; // the 256 digit totals of the current pass
    int capacityEntries = 0;
    // Optional device-side element count (<= the host-side n the grids are sized for); set per call by the renderer
    // when the splat list was culled on the device, nullptr otherwise.
    const int* countDevice = nullptr;
    // Tile range table (int32[2 * tiles], preset to -1) that the LAST pass of the next sort fills from its scatter -- the key is then the
    // tile index and no range kernel is needed (GaussianRender.cu:857-888); set per call by the renderer, null otherwise.
    int32_t* rangeTableForLastPass = nullptr;
    // The staged buildTileList (gs_staged.hip) wants the reference's 64-bit keys beside the sorted values: the LAST pass of the next
    // sort of tile indices then writes keys64[pos] = tile << 32 | the pair's quantised depth (GaussianRender.cu:446-454) instead of the
    // tile index itself.  Set per call, null otherwise.
    uint64_t* lastPassKeys64 = nullptr;
    // ... and every pass carries a second 32-bit payload along with the values -- the pairs' quantised depths, from the emit pass --
    // between these two buffers (carry[0] = the input of the first pass); the last pass puts it into the key's low word.
    uint32_t* carry[2] = {nullptr, nullptr};
    int carrySel = 0; // which of the two holds the next pass's input (the sort flips it)
    bool lastPassIsNow = false; // (set by the sort around its last pass)
    RadixScanCompanion firstScanCompanion; // see above; set per call by the renderer
};
// The clip stage of a frame whose depth sort makes its own keys (Morton storage, no band cull), deferred: launched by the sort as the
// companion of its column scans (gs_clip.hip splatClipDataScanKernel): slice `done` of `parts` equal runs of its workgroups with every
// scan the sort offers (kClipScanParts below: one slice is what pays), the tile start order's builder (gs_tile_order.h) with the first.
// Writes the two 32-byte records per (virtual) splat and resets the range table, exactly as launchSplatClipData does with the same
// arguments.
struct ClipWithScan
{
    int count = 0;
    Screen scr{};
    ClipCamera camera{};
    const FrameParams* params = nullptr; // captured frames: cameras on the device, one per view
    const float4* position = nullptr;
    const float4* scaleAndRotation = nullptr;
    int32_t* rangeTableToClear = nullptr;
    int rangeEntries = 0;
    float4* tileRecord = nullptr;
    float4* rasterRecord = nullptr;
    int views = 1;
    ushort4* tileBox = nullptr;
    TileOrderArgs tileOrder{}; // slots > 0: the builder's eight workgroups ride in the first slice's launch
    int parts = 1;             // slices the clip kernel's workgroups are dealt into
    int done = 0;              // slices launched so far
};
// Slices in use.  Round 6 measured 1, 2 and 3 (the depth sort has three scans, 5 us of latency each, and a third of the clip kernel's 22 us
// would hide each of them): a slice pays the kernel's ramp-up again (a third takes 10.3 us beside its scan, not 7.3), so the lone
// frame gains 3-5 us with two or three slices against 2-4 us with one, and with the tile start order -- whose builder needs the whole
// kernel's 22 us to hide in -- slices lose what they gain (profiles/r06_clip_scan_slices.txt).  One slice: the whole kernel with the first scan.
#ifndef GS_CLIP_SCAN_PARTS
#define GS_CLIP_SCAN_PARTS 1
#endif
constexpr int kClipScanParts = GS_CLIP_SCAN_PARTS;
// Launches slice clip.done with the scan in front of it (and counts it); ...Rest: the slices no scan took, in one launch of their own.
void launchSplatClipDataWithScan(hipStream_t stream, ClipWithScan& clip, const RadixScanArgs& scan);
void launchSplatClipDataRest(hipStream_t stream, ClipWithScan& clip);
inline void clipWithScanCompanion(hipStream_t stream, const RadixScanArgs& scan, void* user)
{
    launchSplatClipDataWithScan(stream, *static_cast<ClipWithScan*>(user), scan);
}
bool radixSortSupported(int keyBytes, int n);
// Passes launchRadixSortPairs makes over bits [beginBit, endBit) (digits of up to 9 bits); the sorted data ends in buffer pair passes % 2.
int radixSortPasses(int beginBit, int endBit);
void ensureRadixScratch(RadixScratch& s, int n);
void freeRadixScratch(RadixScratch& s);
// First pass of the depth sort of a renderer with Morton storage: where the keys come from when they are made by the sort itself
// (gs_radix.hip, KEYGEN): the caller-ordered positions and the camera(s).
struct DepthKeySource
{
    const float4* positionCaller = nullptr; // position of the splat with caller index c
    const FrameParams* params = nullptr;    // captured frames: the cameras live on the device, one per view
    ClipCamera camera{};                    // otherwise
    int splatsPerView = 0;
};
// Returns the selector (0/1) of the buffer pair holding the sorted data (input is buffer 0).
// inputValues (optional, n entries, only read): the values of the input instead of values0.  keySource (optional; 27-bit depth sort
// only): the input keys are not in keys0 yet -- the first pass computes key i from the caller-ordered positions and leaves it in keys0[i].
// Together: element i = (depth code of the splat with caller index i, its storage slot inputValues[i]) -- the input is in the CALLER's
// order, and a stable sort leaves equal keys in input order: ties blend in the caller's index order however the splats are stored.
int launchRadixSortPairs(hipStream_t stream, RadixScratch& s, int keyBytes, void* keys0, void* keys1, int32_t* values0, int32_t* values1,
                         int n, int beginBit, int endBit, const int32_t* inputValues = nullptr, const DepthKeySource* keySource = nullptr);
// One pass's column scan by itself: table[digit][tile] (rows `stride` apart, ceil(n / tileElements) live tiles) -> exclusive scan of
// every digit's row + its total.  Used by the list builder when its emit pass carries the tile sort's first pass (gs_tiles_fast.hip).
void launchRadixColumnScan(hipStream_t stream, int digits, int nHost, const int* nDevice, int tileElements, int stride, const int32_t* table,
                           int32_t* scanned, int32_t* totals);
// The same input sequence materialised from storage-ordered keys (for the rocPRIM backend, which has no such hook):
// keysOut[i] = keysIn[inputOrder[i]], valuesOut[i] = inputOrder[i].
void launchGatherSortInput(hipStream_t stream, int n, const uint32_t* keysIn, const int32_t* inputOrder, uint32_t* keysOut, int32_t* valuesOut);

// keyBytes 8: tile = key >> 32 (the reference's keys); 2 or 4: the key is the tile index.
// sizeDevice (optional): the list length lives on the device (captured frame); `size` is then the upper bound the grid covers.
void launchTileRanges(hipStream_t stream, int size, int keyBytes, const void* keys, int32_t* tileRange,
                      const int* sizeDevice = nullptr);

// Where the raster kernel gets an entry's colour when it is NOT read from the colour buffer: the splat's coefficient row
// ([splat][shCount] layout) and position, evaluated when the entry is staged (gs_raster.hip, template parameter SH).
struct RasterColourSource
{
    int degree = 0, shCount = 0;
    int rowStride = 0; // floats between the rows of shPerSplat: shCount rounded up to a 16-byte multiple (27 -> 28, 75 -> 76)
    const float4* position = nullptr;
    const float* shPerSplat = nullptr;
    float3 cameraPosition = {0.f, 0.f, 0.f};
    const FrameParams* params = nullptr; // captured frames: the camera position lives on the device
};
// Depth layers (gs_renderer_set_depth_layers, gs_raster.hip LAYERED): what a layer's raster launch needs besides the usual arguments.
struct RasterLayer
{
    uint8_t* status = nullptr;        // per (virtual) tile: 0xFF untouched, 1 open with state, 2 stopped by a vote
    float4* state = nullptr;          // per pixel (of every view): colour sums and transmittance of an open tile
    int32_t* consumedSoFar = nullptr; // per (virtual) tile: list entries consumed by the layers so far
    int32_t* rangeToReset = nullptr;  // the range table (all views): a tile's entries go back to -1 once it has read them
    uint32_t* openBits = nullptr;     // one bit per (virtual) tile, rows of openWordsPerRow words: cleared when a vote stops the tile
    int openWordsPerRow = 0;          //   (what the list builder of the next layers looks at)
    int isFirst = 1, isLast = 1;
};
// Experiment hooks of the raster launch (gs_renderer_debug_set_tile_order): the tile every grid slot renders (-1: none) instead of the
// XCD-aware screen-order map, and where every tile leaves the number of list entries it consumed.
struct RasterSchedule
{
    const int32_t* order = nullptr;   // [grid slots of the launch]
    int32_t* consumedOut = nullptr;   // [(virtual) tiles]: list entries the tile has consumed in this frame (all its layers)
    const int32_t* enabled = nullptr; // device word: the table is used only while it is non-zero (null: always)
};
bool rasterCanEvaluateColours(int degree, int shCount);
void launchRasterize(hipStream_t stream, const Screen& scr, const int32_t* tileRange, const int32_t* values,
                     const float2* clipXY, const float4* conic, const float4* color, uchar4* backBuffer,
                     unsigned long long* consumedPairsOrNull, bool clearEmptyTiles = false,
                     const RasterColourSource* lazyColoursOrNull = nullptr, const float4* rasterRecordOrNull = nullptr, int views = 1,
                     int viewSplats = 0, const RasterLayer* layerOrNull = nullptr, const RasterSchedule* scheduleOrNull = nullptr);
// Grid slots launchRasterize uses for a band of `rows` tile rows (what a RasterSchedule::order table must cover).
int rasterGridSlots(int tilesX, int rows);

// The renderer sorts its splats on a 27-bit re-coding of getKey()'s 32-bit quantised depth (GaussianRender.cu:446-454): the
// quantised value comes from a float, so it has at most 24 significant bits -- (position of the top bit, next 23 bits) says
// everything: same order, same ties, three 9-bit radix passes instead of four 8-bit ones (gs_clip.hip: compactDepthKey).
constexpr int kDepthKeyBits = 27;
int sortEndBit(int totalTiles);
int tileBits(int totalTiles);

} // namespace gs
