/*
 * gsplat_hip_debug.h -- the lab behind libgsplat_hip.so: diagnostics, A/B switches, counters and trace hooks.
 *
 * Nothing in here is part of the drop-in boundary (gsplat_hip.h holds the reference's interface, the documented additions and the
 * one-call renderer); a host of the reference never needs this header.  bench.py, tools/ and the tests use it to measure, to compare
 * variants of one library on one box, and to inspect what a frame did.  Switches that shape a renderer's frames are RENDERER state
 * (gs_renderer_debug_set_*); the process-wide forms that remain are the defaults new frames of every renderer fall back to and what
 * the staged calls use (they have no renderer), and are meant to be set once, before frames are in flight.
 */
#ifndef GSPLAT_HIP_DEBUG_H
#define GSPLAT_HIP_DEBUG_H

#include "gsplat_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- counters ------------------------------------------------------------------------------------------------------------ */
/* Addition: number of list entries the last gs_rasterize_tile loaded before tile-wide termination
 * (P_c of DESIGN.md); collected only when enabled, costs one atomic per tile. */
void gs_set_collect_raster_stats(int enabled);
int64_t gs_last_raster_consumed_pairs(void);
/* Number of captured frames that had to be rendered again because the list was too small. */
int gs_renderer_captured_retries(const GsRenderer* r);
int gs_renderer_depth_layer_changes(const GsRenderer* r); /* layers = 0: times the renderer changed its choice (each one re-captures the frame; steps
                                                           * down wait 16 frames, doubled whenever one is undone within 4 frames) */
/* ---- A/B switches of a renderer ------------------------------------------------------------------------------------------- */
void gs_renderer_debug_set_layer_compaction(GsRenderer* r, int enabled); /* A/B: 0 = later layers walk all their splats, not only those with an open
                                                                           * tile under their box (default 1; same lists and images) */
/* Experiment (profiles/r06_layer_compaction_bound.txt): 1 = the later layers' live-splat lists of a layered captured frame are kept in buffers of
 * their own; 2 = the three compaction kernels per later layer are NOT launched and the lists the last mode-1 frame left are walked again
 * (only meaningful for the SAME camera) -- the frame a compaction fused into the count pass at no cost would be: the upper bound of that
 * fusion; 0 = off.  Set after the scene. */
void gs_renderer_debug_set_layer_compaction_reuse(GsRenderer* r, int mode);
/* A/B and tests: 0 = the band cull of a Morton-stored scene projects all N splats (round 5's kernel) instead of testing blocks of 256
 * Morton neighbours first (default 1; same lists and images: the per-splat keep mask decides either way). */
void gs_renderer_debug_set_block_cull(GsRenderer* r, int enabled);
/* Blocks of 256 splats the last band-culled frame's block test kept, summed over its views (*total: blocks x views); -1 when the
 * renderer has no block bounds (caller's storage order).  Waits for the renderer's stream. */
int64_t gs_renderer_debug_block_cull_kept(GsRenderer* r, int64_t* total);
/* Splats the last band-culled frame kept (all its views): the length of the list its depth sort ran over.  Waits for the stream. */
int64_t gs_renderer_debug_band_kept_splats(GsRenderer* r);
/* Measurement (bench.py roofline.timed_loop_kernel): one frame of `camera` made of the very kernels of this renderer's captured LAYERED
 * frame (gs_renderer_set_depth_layers), launched plainly on its stream, one frame at a time, with a HIP event pair around every layer's
 * raster launch and four counters per layer.  The renderer must have rendered a layered captured frame before (-3 otherwise; single
 * view).  stateTileLoads / stateTileStores: tiles that read / wrote their per-pixel state (16 bytes per pixel) in that layer. */
typedef struct GsLayeredRasterReport
{
    int32_t layers;
    float rasterMs[4];
    int64_t consumedPairs[4];
    int64_t tilesRun[4];
    int64_t stateTileLoads[4];
    int64_t stateTileStores[4];
} GsLayeredRasterReport;
int gs_renderer_debug_layered_raster_report(GsRenderer* r, const GsCameraData* camera, GsLayeredRasterReport* out);
/* gs_renderer_set_tile_start_order's state after the renderer's last frame: out[0] 1 = the table is in force for the next frame, out[1] sum
 * over the tiles of |consumed now - consumed in the frame before|, out[2] sum consumed now, out[3] frames the order kernel has seen; -1 when
 * the option never ran.  Waits for the renderer's stream.  ..._max_mismatch: the largest out[1] / out[2] at which the table is put in force
 * (default 0.35). */
int gs_renderer_debug_tile_order_state(GsRenderer* r, int32_t out[4]);
int gs_debug_staged_tile_order_state(int32_t out[4]); /* the same for the calling thread's staged loop (gs_set_staged_tile_start_order) */
void gs_renderer_debug_set_tile_order_max_mismatch(GsRenderer* r, float maxMismatch);
/* Experiment ("start order of the raster's tiles"; tools/tile_order_experiment.py): plain (un-captured, single-view)
 * frames of this renderer start their raster tiles in the order of a table -- grid slot -> tile index, -1 = no tile; `slots` must be
 * gs_renderer_debug_raster_grid_slots (slot q runs on XCD q % 8); NULL / 0 = the XCD-aware screen order again.  The order in which tiles
 * start changes no byte of the frame.  gs_renderer_debug_tile_consumed: first call arms the per-tile counters (returns 1), later calls
 * copy the list entries every tile consumed in the last plain frame to perTileHost (returns 0). */
int gs_renderer_debug_raster_grid_slots(const GsRenderer* r);
int gs_renderer_debug_set_tile_order(GsRenderer* r, const int32_t* slotToTileHost, int slots);
int gs_renderer_debug_tile_consumed(GsRenderer* r, int32_t* perTileHost);
/* This renderer's frames with (1) / without (0) the emit pass that carries the tile sort's first pass; -1 (default) = the process-wide
 * setting (gs_set_list_fusion). */
void gs_renderer_debug_set_list_fusion(GsRenderer* r, int enabledOrMinus1);
/* Pixels per lane of this renderer's raster launches: 1, 2 or 4; 0 (default) = the process-wide setting. */
void gs_renderer_debug_set_raster_pixels_per_lane(GsRenderer* r, int pixelsPerLaneOr0);

/* ---- process-wide defaults (A/B runs set them once, before any frame) --------------------------------------------------------- */
/* Addition: radix sort used by the one-call renderer for its two sorts (depth keys of the splats; tile index of
 * the pairs). NATIVE = the three-kernel LSD sort of gs_radix.hip (default), ROCPRIM = rocPRIM's device sort. */
#define GS_SORT_BACKEND_ROCPRIM 0
#define GS_SORT_BACKEND_NATIVE 1
void gs_set_sort_backend(int backend);
int gs_get_sort_backend(void);
/* Addition: the renderer's emit pass may carry the FIRST PASS of its tile sort (native backend, two-pass sorts of 16-bit tile
 * indices): the count pass also counts the pairs' low tile digits per group of 256 splats, a column scan turns the table into output
 * positions, and the emit pass ranks its pairs in LDS and writes them where the first scatter would have put them -- one histogram
 * kernel and one scatter kernel less per frame, the unsorted list is never written.  Same sorted list, same image, bit for bit
 * (replaces cub::DeviceRadixSort::SortPairs at GaussianRender.cu:806-855 together with the emission at :684-712).  Process-wide. */
#ifndef GS_LIST_FUSION_DEFAULT
#define GS_LIST_FUSION_DEFAULT 1
#endif
void gs_set_list_fusion(int enabled);
int gs_get_list_fusion(void);
/* The native sort on its own: stable LSD sort of (keyBytes in {2,4} key, int32 value) pairs on key bits
 * [beginBit, endBit); input in buffers 0, returns the index (0/1) of the buffers holding the result, -1 if the
 * size/key type is not supported. Device pointers, enqueued on the current stream. */
int gs_radix_sort_pairs(int keyBytes, void* keys0, void* keys1, int32_t* values0, int32_t* values1, int32_t count, int beginBit,
                        int endBit);
/* Addition (tuning): pixels per lane of the raster kernel: 1 (four waves per tile), 2 (two waves, default) or
 * 4 (one wave per tile). */
void gs_set_raster_pixels_per_lane(int pixelsPerLane);
int gs_get_raster_pixels_per_lane(void);
/* ---- diagnostics ------------------------------------------------------------------------------------------------------- */
/* Diagnostics (tools/raster_trace.py): when set, every raster workgroup writes four uint64 to deviceBuffer[4 * tile]:
 * start and end on the 100 MHz wall clock, the tile's list length, and (dispatch index << 32 | consumed entries).
 * Process-wide; NULL (default) switches it off.  Not for captured frames (the pointer is a frozen kernel argument there). */
void gs_debug_set_raster_trace(uint64_t* deviceBufferOrNull);
/* Diagnostics (bench.py --ablate): leave stages out of the captured frames -- bit 0 clip, 1 depth sort, 2 pair count, 3 pair emit,
 * 4 tile sort, 5 ranges, 6 raster -- to measure what each stage costs the pipelined frame rate.  The images are garbage. */
void gs_debug_set_ablation(int mask);
/* A renderer with Morton storage (no band cull) has the depth sort make its own keys, so its clip kernel and the sort are independent:
 * the clip kernel may share the LAUNCH of the sort's first column scan (one kernel with both workloads: 17 instead of 18 kernels per
 * frame; same results, same routines).  Measured (profiles/r04_ab_experiments.txt): one frame at a time -- the reference's loop,
 * Demo.cpp:394-479 -- gains 3 % (0.275 -> 0.266 ms), four frames in flight lose 1 %; OFF by default, for hosts that render one frame
 * at a time to switch on.  Process-wide.  Frames whose stages are timed one by one (timed != 0) always launch the two separately, so
 * that the clip stage's time is its own.  Round 6: the builder of gs_renderer_set_tile_start_order rides in the shared launch as it does
 * in the plain clip kernel (the two options of an interactive host add up: -2 ... -4 us on top of the start order's -14 us); dealing the
 * clip kernel in slices to all three scans of the sort was measured and is no better (profiles/r06_clip_scan_slices.txt). */
void gs_debug_set_clip_scan_fusion(int enabled);
int gs_debug_get_clip_scan_fusion(void);
/* A captured frame hands its pair total to the host in the middle of the frame; the host waits for it at most this long (default
 * 30 s) and gives up at once when the stream reports anything but "not ready" -- print + exit, the reference's convention for a
 * failed runtime call (CudaErrorCheck.cpp:2-11), so that a faulted GPU ends the process instead of hanging it. */
void gs_debug_set_await_timeout(double seconds);
/* Tests: allocations of the sorted staged builder's hidden scratch (gs_set_staged_builder) larger than this many bytes fail as if the
 * device were out of memory -- the call must then fall back to the reference's emission order instead of ending the process; 0 = off. */
void gs_debug_set_staged_scratch_limit(int64_t maxBytesPerAllocationOr0);
/* Diagnostics (tools/frame_pump): host-side cost of the captured frames of this process since the last reset -- average microseconds
 * per frame inside hipGraphLaunch and waiting for the frame's pair total; returns the number of frames averaged over. */
long long gs_debug_host_times(double* launchMicroseconds, double* awaitMicroseconds, int reset);
/* Diagnostics: a one-rank RCCL communicator sends `bytes` from src to dst (both device memory) through the grouped ncclSend/ncclRecv
 * path gs_gather_bands uses, on one GPU; blocks until done.  0 = ok, -1 RCCL not loadable, -2 RCCL error. */
int gs_debug_rccl_self_exchange(void* dst, const void* src, size_t bytes, void* hipStream);
/* ... with the block cut into `parts` send/receive pairs inside one group (several operations per peer, as gs_gather_bands_views issues). */
int gs_debug_rccl_self_exchange_parts(void* dst, const void* src, size_t bytes, int parts, void* hipStream);
/* ... through `comms` (1..8) communicators, each on a stream of its own, all in flight at once and each used twice: how a host with
 * several frames in flight drives the gather (one communicator per frame in flight, never one communicator on two streams). */
int gs_debug_rccl_self_exchange_streams(void* dst, const void* src, size_t bytes, int comms);

#ifdef __cplusplus
}
#endif
#endif
