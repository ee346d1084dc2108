// gs_raster.hip -- 16x16-tile front-to-back alpha compositing.
//
// Replaces rasterizeTilesKernel (GaussianRender.cu:908-1016).  Same observable semantics:
//   alpha = colour.a * saturate(exp(-0.5 * d^T Q d)), C += rgb * T * alpha, T *= 1 - alpha, every list
//   entry contributes (no alpha cut-off), and a tile stops only when EVERY on-screen pixel has
//   T <= 0.02, voted after each 32 list entries (the reference's chunk, :950,:995).
//
// CDNA4 mapping:
//   * one workgroup per tile; each lane owns PPL horizontally adjacent pixels (PPL = 4 -> the tile is ONE
//     64-lane wave, the termination vote is a single __ballot and there is no workgroup barrier cost;
//     PPL = 1 -> four waves like the reference).  The dy-dependent half of the quadratic form is shared
//     by a lane's pixels.
//   * list entries are gathered 64 at a time (one entry per lane, 44 B: value -> centre, conic, colour),
//     staged in LDS as two float4 + one float per entry with the -0.5*log2(e) factor folded into the
//     conic, and read back as wave-uniform (broadcast) ds_read_b128;
//   * the gather of batch b+1 is issued into registers before batch b is blended, so HBM/L2 latency is
//     covered by the VALU work even with one wave per tile;
//   * density uses the hardware exp2 (v_exp_f32) -- the reference is built with --use_fast_math/__expf;
//   * blockIdx -> tile is XCD-aware: the screen is cut into blocks of 8 x 4 tiles dealt round-robin to the 8 XCDs, so
//     neighbouring tiles in both directions (which share splats) share an L2.
// The kernel is VALU/transcendental-bound, not HBM-bound: per consumed entry it reads 44 B but does
// ~11 VALU ops for each of 256 pixels ("Raster roofline").
#include "gs_state.h"
#include "gs_sh_eval.h"
#include "gs_tile_order.h"

namespace gs
{
namespace
{

constexpr int kBatch = 64; // entries staged per LDS fill; two vote chunks
// Diagnostics build (-DGS_RASTER_PHASE_TRACE): with a trace buffer set, thread 0 of every tile sums the 100 MHz ticks it spends in
// each phase of the loop -- waiting for the gather, staging (SH evaluation + LDS writes), barriers, blending, voting -- and leaves the
// five sums in place of the regular trace record (tools/raster_lone_tile.py --phases).
#ifdef GS_RASTER_PHASE_TRACE
#define GS_PHASE(k)                                       \
    do                                                    \
    {                                                     \
        const unsigned long long now_ = wall_clock64();   \
        phase[k] += now_ - phaseT;                        \
        phaseT = now_;                                    \
    } while (0)
#else
#define GS_PHASE(k)
#endif
#ifndef GS_RASTER_PK_OPSEL
#define GS_RASTER_PK_OPSEL 1 // A/B builds: 0 = the colour accumulation as the compiler writes it
#endif
// blockIdx -> tile mapping: block shape dealt to one XCD (A/B builds may override)
#ifndef GS_RASTER_BLOCK_W
#define GS_RASTER_BLOCK_W 8
#define GS_RASTER_BLOCK_H 4
#endif

// A lane's PPL pixels as one vector value: the per-pixel multiplies and fmas then compile to packed FP32 instructions
// (v_pk_fma_f32 / v_pk_mul_f32: two pixels per issue slot); the results are the same IEEE operations as scalar code.
template <int N>
struct PixelVec
{
    typedef float type __attribute__((ext_vector_type(N)));
    static __device__ __forceinline__ float get(const type& v, int i) { return v[i]; }
    static __device__ __forceinline__ void set(type& v, int i, float x) { v[i] = x; }
};
template <>
struct PixelVec<1>
{
    typedef float type;
    static __device__ __forceinline__ float get(const type& v, int) { return v; }
    static __device__ __forceinline__ void set(type& v, int, float x) { v = x; }
};

template <int THREADS>
__device__ __forceinline__ bool anyInBlock(bool pred)
{
    if constexpr (THREADS == kWave) return __ballot(pred) != 0ull;
    else return __syncthreads_or(pred ? 1 : 0) != 0;
}

__device__ __forceinline__ uint32_t quantize8(float c)
{
    // GaussianRender.cu:1007-1011: float -> uchar conversion (truncate, saturate, NaN -> 0).
    const float v = c * 255.0f;
    return (uint32_t) fminf(fmaxf(v, 0.0f), 255.0f); // fmaxf(NaN, 0) = 0
}

// EPAN: the reference's EPANECHNIKOV_KERNEL option (GaussianRender.cu:977-981): density 1 - q/7 instead of exp(-q/2).
// SH: 0 = colours come from the colour buffer (the reference's stage order: evaluateSphericalHarmonics ran before);
//     1..4 = the colour of a list entry is evaluated HERE from the splat's coefficient row when the entry is staged, with the
//     routine the SH kernels use (same bits).  A frame consumes the lists' front only -- at C3 88 K distinct splats of 1 M --
//     so evaluating on demand replaces the 224 MB pass over all coefficients by ~4 memory sectors per staged entry.
// LAYERED (gs_renderer_set_depth_layers): the frame's depth-sorted splats are listed, sorted and blended in several layers, nearest
// first, and a layer's list holds no pairs of tiles that an earlier layer already finished.  A tile's blend is the same sequence of
// operations as over its whole list: a tile whose list ran out before a vote stopped it leaves its per-pixel state (colour sums and
// transmittance) and the number of entries it has consumed; the next layer that holds entries for it loads them and goes on, with as
// many zero-opacity lead entries as the tile is past its last vote, so that the votes stay at multiples of 32 entries of the WHOLE
// list (GaussianRender.cu:945-998); a vote is only taken on a full chunk unless this is the last layer (the list's real end).
// status[tile]: 0xFF untouched (the per-frame fill of the range table reaches it), 1 open with state, 2 stopped by a vote.
#ifndef GS_RASTER_LAYERED_WAVES
#define GS_RASTER_LAYERED_WAVES 1 // A/B builds: 4 = four waves per SIMD asked for (127 VGPRs + 16 bytes of scratch)
#endif
#ifndef GS_RASTER_LAYERED_MAX_WAVES
#define GS_RASTER_LAYERED_MAX_WAVES 8
#endif
template <int PPL, bool EPAN, int SH, bool LAYERED = false>
// (the layered SH-3 instantiation comes out at 130 VGPRs -- three waves per SIMD where the one-list kernel has four.  Asking for four
// (127 VGPRs, 16 bytes of scratch) was measured: -3.6 % frames/s at BASELINE.json configs[2], -1.3 % at configs[4], order-balanced runs on
// one box, tools/r05_ab_lib.sh -- so it stays as the compiler allocates it)
__global__ __launch_bounds__(256 / PPL) __attribute__((amdgpu_waves_per_eu(LAYERED && SH == 3 ? GS_RASTER_LAYERED_WAVES : 1, LAYERED && SH == 3 ? GS_RASTER_LAYERED_MAX_WAVES : 8))) void rasterizeTilesKernel(Screen scr, int firstTile, int numTiles,
                                                                  const int32_t* __restrict__ tileRange,
                                                                  const int32_t* __restrict__ values,
                                                                  const float2* __restrict__ clipXY,
                                                                  const float4* __restrict__ conic,
                                                                  const float4* __restrict__ color, uchar4* __restrict__ backBuffer,
                                                                  unsigned long long* __restrict__ consumedPairs, int clearEmptyTiles,
                                                                  RasterColourSource lazy, unsigned long long* __restrict__ trace,
                                                                  const float4* __restrict__ rasterRecord, int viewSplats, RasterLayer layer,
                                                                  RasterSchedule schedule)
{
    constexpr int THREADS = 256 / PPL;
    constexpr int GROUPS_X = kTileSize / PPL;
    __shared__ float4 sA[kBatch];   // cx, cy, A', B'
    __shared__ float4 sB[kBatch];   // C', r, g, b
    __shared__ __attribute__((aligned(16))) float sAlpha[kBatch];

    // XCD-aware remap: workgroups b, b+8, b+16, ... share an XCD (round-robin dispatch).
    // The band is cut into blocks of 8 x 4 tiles (BH at most the band's height); block j belongs to XCD j % 8 and an XCD walks its
    // blocks one after another, so a splat that overlaps neighbouring tiles in BOTH directions is gathered through ONE of the
    // eight L2s (they are not coherent with each other: what two XCDs both need is fetched from HBM twice).  Measured at C3
    // against round 1's runs of 8 tiles in a row (same frames/s within 0.5 %): the kernel's FETCH_SIZE drops from 98.7 to 57.3 MB.
    // Larger blocks share more but balance worse: 8 x 8 51.4 MB at -2 % frames/s, 16 x 4 43.4 MB at -16 %; 4 x 4 69 MB.
    // Edge blocks have idle workgroups (they exit).
    int tile;
    if (schedule.order != nullptr && (schedule.enabled == nullptr || *schedule.enabled != 0))
    {
        // Experiment (gs_renderer_debug_set_tile_order): grid slot -> tile from a table, e.g. the tiles that
        // consumed most entries in the previous frame first.  The order in which tiles START changes no tile's arithmetic.
        tile = schedule.order[blockIdx.x];
        if (tile < 0) return;
    }
    else
    {
        constexpr int BW = GS_RASTER_BLOCK_W;
        const int rows = numTiles / scr.tilesX;
        const int bh = rows < GS_RASTER_BLOCK_H ? rows : GS_RASTER_BLOCK_H;
        const int blocksX = (scr.tilesX + BW - 1) / BW;
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        const int perBlock = BW * bh;
        const int j = (q / perBlock) * 8 + xcd, inside = q % perBlock;
        const int bx = j % blocksX, by = j / blocksX;
        const int tx = bx * BW + inside % BW, ty = by * bh + inside / BW;
        if (tx >= scr.tilesX || ty >= rows) return;
        tile = firstTile + ty * scr.tilesX + tx;
    }
    const unsigned long long traceStart = trace ? wall_clock64() : 0ull; // diagnostics only (gs_debug_set_raster_trace)

    // Several views per chain (gs_renderer_set_views): grid row v renders view v -- its ranges are the virtual tiles
    // v*totalTiles + t, its image the v-th of the back buffer, its list entries virtual splats v*N + i (clip data stored per
    // view under that index; position and coefficient row are splat i's).
    const int view = blockIdx.y;
    const int viewFirstSplat = view * viewSplats;
    tileRange += (size_t) view * 2 * scr.totalTiles;
    backBuffer += (size_t) view * scr.width * scr.height;

    // GaussianRender.cu:917-931 (wave-uniform: scalar loads)
    const int vtile = view * scr.totalTiles + tile; // (tileRange already points at this view's table: the range is tileRange[tile * 2])
    unsigned tileStatus = 0xFFu;
    if constexpr (LAYERED)
    {
        tileStatus = layer.status[vtile];
        if (tileStatus == 2u) return; // an earlier layer's vote stopped this tile: its list in this layer is empty anyway
    }
    int first = tileRange[tile * 2];
    int count = tileRange[tile * 2 + 1] - first;
    if (count == 0 && (!clearEmptyTiles || (LAYERED && !layer.isFirst))) return; // (a later layer leaves an idle tile as it is)

    const int t = threadIdx.x;
    const int tileX = tile % scr.tilesX, tileY = tile / scr.tilesX;
    const int px0 = tileX * kTileSize + (t % GROUPS_X) * PPL;
    const int py = tileY * kTileSize + t / GROUPS_X;
    if (count == 0)
    {
        // The reference leaves tiles without work at the caller's memset value (0,0,0,0) (Demo.cpp:399); the renderer
        // skips that 8 MB memset and has the empty tiles write their zeros here instead.
        if (py < scr.height)
        {
#pragma unroll
            for (int p = 0; p != PPL; ++p)
                if (px0 + p < scr.width) backBuffer[(size_t) py * scr.width + px0 + p] = make_uchar4(0, 0, 0, 0);
        }
        return;
    }
    // GaussianRender.cu:939 (no half-pixel offset)
    using PV = PixelVec<PPL>;
    using V = typename PV::type;
    V clipPx;
    bool onScreen[PPL];
#pragma unroll
    for (int i = 0; i != PPL; ++i)
    {
        PV::set(clipPx, i, (float) (px0 + i) * scr.pixelClipSizeX - 1.0f);
        onScreen[i] = (px0 + i) < scr.width && py < scr.height;
    }
    const float clipPy = (float) py * scr.pixelClipSizeY - 1.0f;

    V cr = V(0.0f), cg = V(0.0f), cb = V(0.0f), tr = V(1.0f);
    int lead = 0, consumedBefore = 0; // LAYERED: zero-opacity entries in front of this layer's list; entries consumed by earlier layers
    if constexpr (LAYERED)
    {
        if (tileStatus == 1u)
        {
            consumedBefore = layer.consumedSoFar[vtile];
            lead = consumedBefore & (kRasterChunk - 1);
            const float4* state = layer.state + (size_t) view * scr.width * scr.height;
#pragma unroll
            for (int i = 0; i != PPL; ++i)
                if (onScreen[i])
                {
                    const float4 v = state[(size_t) py * scr.width + px0 + i];
                    PV::set(cr, i, v.x);
                    PV::set(cg, i, v.y);
                    PV::set(cb, i, v.z);
                    PV::set(tr, i, v.w);
                }
        }
        // virtual list: `lead` no-op entries, then this layer's entries
        first -= lead;
        count += lead;
    }

    // Register-staged gather of one list entry per lane.
    float2 gCentre = make_float2(0.f, 0.f);
    float4 gConic = make_float4(0.f, 0.f, 0.f, 0.f), gColor = make_float4(0.f, 0.f, 0.f, 0.f);
    // (rows of the renderer's [splat][row] coefficient copy are padded to 16-byte multiples: degrees 2 and 4 read one pad float)
    constexpr int kRowQuads = SH ? sheval::kRowQuads<(SH ? SH : 1)> : 1;
    float4 gRow[kRowQuads]; // SH: the splat's coefficient row, in flight while the previous batch is blended
    float3 lazyCam = make_float3(0.f, 0.f, 0.f);
    if constexpr (SH != 0)
    {
        lazyCam = lazy.params ? make_float3(lazy.params[view].cameraPosition[0], lazy.params[view].cameraPosition[1],
                                            lazy.params[view].cameraPosition[2])
                              : lazy.cameraPosition;
#pragma unroll
        for (int m = 0; m != kRowQuads; ++m) gRow[m] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // The list entry (a splat index) of the batch AFTER the one being gathered is already on its way: a gather is two dependent
    // loads (index, then the splat's records), and a tile that runs alone on its SIMD -- the kernel's tail -- would otherwise wait
    // for both, one after the other, once per batch (measured: 62-70 ns per entry for a lone tile whatever its arithmetic costs).
    int nextSrc = (t < kBatch && t < count && t >= lead) ? values[first + t] : 0;
    auto gather = [&](int base) {
        const int src = nextSrc;
        if (t < kBatch && base + kBatch + t < count) nextSrc = values[first + base + kBatch + t]; // GaussianRender.cu:957-960 (lead < 32 <= base + kBatch)
        if (t < kBatch && base + t < count && base + t >= lead)
        {
            if (rasterRecord != nullptr)
            {
                // the renderer: centre and conic from ONE 32-byte record (one memory sector); wave-uniform branch
                const float4 ra = rasterRecord[2 * (size_t) src], rb = rasterRecord[2 * (size_t) src + 1];
                gCentre = make_float2(ra.x, ra.y);
                gConic = make_float4(ra.z, ra.w, rb.x, 0.0f);
            }
            else
            {
                gCentre = clipXY[src];
                gConic = conic[src];
            }
            const int splat = src - viewFirstSplat; // the scene's splat behind the (virtual) list entry
            if constexpr (SH == 0) gColor = color[splat];
            else
            {
                gColor = lazy.position[splat]; // xyz + opacity; becomes the colour when the entry is staged
                const float4* row = reinterpret_cast<const float4*>(lazy.shPerSplat + (size_t) splat * lazy.rowStride);
#pragma unroll
                for (int m = 0; m != kRowQuads; ++m) gRow[m] = row[m];
            }
        }
    };
    gather(0);

    // the conic is staged pre-scaled: exp(-q/2) = exp2(q * -0.5 log2 e); Epanechnikov: 1 - q/7 = 1 + q * (-1/7)
    constexpr float kHalfLog2e = EPAN ? (-1.0f / 7.0f) : (-0.5f * 1.4426950408889634f);
    int base = 0, consumed = 0;
#ifdef GS_RASTER_PHASE_TRACE
    unsigned long long phase[6] = {0, 0, 0, 0, 0, 0}, phaseT = wall_clock64();
#endif
    for (;;)
    {
        const int n = min(count - base, kBatch);
        __syncthreads(); // every wave is done reading the previous batch
        GS_PHASE(0);
#ifdef GS_RASTER_PHASE_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GS_PHASE(1);
#endif
        if (t < kBatch)
        {
            // Entries past the end of the list are staged as all-zero records: opacity 0 makes them exact no-ops
            // (w = T*0, C += c*0, T -= T*0), so the blend loop can always run whole groups of four.
            const bool live = t < n && base + t >= lead; // (lead entries: all-zero records like the ones past the end)
            if constexpr (SH != 0)
            {
                if (live) gColor = sheval::colourFromQuads<SH>(gRow, gColor, lazyCam);
            }
            sA[t] = live ? make_float4(gCentre.x, gCentre.y, gConic.x * kHalfLog2e, gConic.y * (2.0f * kHalfLog2e))
                         : make_float4(0.f, 0.f, 0.f, 0.f);
            sB[t] = live ? make_float4(gConic.z * kHalfLog2e, gColor.x, gColor.y, gColor.z) : make_float4(0.f, 0.f, 0.f, 0.f);
            sAlpha[t] = live ? gColor.w : 0.0f;
        }
        GS_PHASE(2);
        __syncthreads();
        GS_PHASE(0);
        if (base + kBatch < count) gather(base + kBatch); // in flight while this batch is blended

        bool stop = false;
        for (int chunk = 0; chunk < n; chunk += kRasterChunk)
        {
            const int m = min(kRasterChunk, n - chunk);
            // Four list entries per step: their LDS reads and their (mutually independent) densities are issued
            // together, then the four order-dependent blends follow.
            for (int i = chunk; i < chunk + m; i += 4)
            {
                float4 a[4], b[4];
                const float4 opacity4 = *reinterpret_cast<const float4*>(&sAlpha[i]);
                const float opacity[4] = {opacity4.x, opacity4.y, opacity4.z, opacity4.w};
#pragma unroll
                for (int e = 0; e != 4; ++e)
                {
                    a[e] = sA[i + e];
                    b[e] = sB[i + e];
                }
                V alpha[4];
#pragma unroll
                for (int e = 0; e != 4; ++e)
                {
                    // GaussianRender.cu:972-982
                    const float dy = clipPy - a[e].y;
                    const float qy = b[e].x * dy * dy; // C' dy^2
                    const float by = a[e].w * dy;      // B' dy
                    const V dx = clipPx - V(a[e].x);
                    const V power = __builtin_elementwise_fma(dx, __builtin_elementwise_fma(V(a[e].z), dx, V(by)), V(qy));
                    V density;
#pragma unroll
                    for (int p = 0; p != PPL; ++p)
                        // saturate = one v_med3_f32 (a NaN density becomes 0, like __saturatef)
                        PV::set(density, p, __builtin_amdgcn_fmed3f(EPAN ? 1.0f + PV::get(power, p) : __builtin_amdgcn_exp2f(PV::get(power, p)), 0.0f, 1.0f));
                    alpha[e] = V(opacity[e]) * density;
                }
#pragma unroll
                for (int e = 0; e != 4; ++e)
                {
                    // GaussianRender.cu:983-984
                    const V w = tr * alpha[e];
                    if constexpr (PPL == 2 && GS_RASTER_PK_OPSEL)
                    {
                        // colour += component * w for both pixels, the component broadcast from where it lies in the staged record
                        // ((C', r) and (g, b) are the two register pairs of the 16-byte LDS read) through the packed instruction's operand
                        // select: the compiler copies b into a pair of its own first (one v_mov per entry and wave)
                        typedef float Pair __attribute__((ext_vector_type(2)));
                        const Pair cR = {b[e].x, b[e].y}, gB = {b[e].z, b[e].w};
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(cr) : "v"(cR), "v"(w));
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(cg) : "v"(gB), "v"(w));
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(cb) : "v"(gB), "v"(w));
                    }
                    else
                    {
                        cr = __builtin_elementwise_fma(V(b[e].y), w, cr);
                        cg = __builtin_elementwise_fma(V(b[e].z), w, cg);
                        cb = __builtin_elementwise_fma(V(b[e].w), w, cb);
                    }
                    tr = __builtin_elementwise_fma(-tr, alpha[e], tr);
                }
            }
            consumed = base + chunk + m;
            GS_PHASE(3);
            // GaussianRender.cu:995: stop when no pixel of the tile has T > 0.02 (off-screen pixels of a
            // partial edge tile do not vote).
            bool alive = false;
#pragma unroll
            for (int p = 0; p != PPL; ++p) alive |= onScreen[p] && PV::get(tr, p) > 0.02f;
            const bool any = anyInBlock<THREADS>(alive);
            GS_PHASE(4);
            // (LAYERED, not the last layer: the list goes on in a later layer -- the reference votes after FULL chunks of 32 only)
            if (!any && (!LAYERED || layer.isLast || m == kRasterChunk))
            {
                stop = true;
                break;
            }
        }
        base += n;
        if (stop || base >= count) break;
    }

    if (consumedPairs != nullptr && t == 0) atomicAdd(consumedPairs, (unsigned long long) (consumed - lead));
    if (schedule.consumedOut != nullptr && t == 0) schedule.consumedOut[vtile] = consumedBefore + consumed - lead; // (what the NEXT frame's start order is made from)
    if constexpr (LAYERED)
    {
        // (diagnostics, gs_renderer_debug_layered_raster_report: tiles that ran in this layer and tiles that loaded their state -- the
        // counter block of a layered launch is four words)
        if (consumedPairs != nullptr && t == 0)
        {
            atomicAdd(consumedPairs + 1, 1ull);
            if (tileStatus == 1u) atomicAdd(consumedPairs + 2, 1ull);
        }
        // `stop` is the loop's exit by a vote (uniform over the workgroup: the vote is block-wide)
        bool stopped = consumed < count; // the loop left entries behind: only a vote does that
        if (consumed == count)
        {
            // the list ran out on this batch: stopped only if the last vote (taken on a full chunk, or at the frame's real end) said so
            bool alive = false;
#pragma unroll
            for (int p = 0; p != PPL; ++p) alive |= onScreen[p] && PV::get(tr, p) > 0.02f;
            const bool any = anyInBlock<THREADS>(alive);
            stopped = !any && (layer.isLast || (count & (kRasterChunk - 1)) == 0);
        }
        if (t == 0)
        {
            layer.status[vtile] = stopped ? 2 : 1;
            if (stopped && !layer.isLast)
                atomicAnd(&layer.openBits[((size_t) view * scr.tilesY + tileY) * layer.openWordsPerRow + (tileX >> 5)], ~(1u << (tileX & 31)));
            layer.consumedSoFar[vtile] = consumedBefore + consumed - lead;
            if (!layer.isLast)
            {
                // the next layer's tile sort fills the range table again (atomic min / max against -1)
                layer.rangeToReset[(size_t) vtile * 2] = -1;
                layer.rangeToReset[(size_t) vtile * 2 + 1] = -1;
            }
        }
        if (!stopped && !layer.isLast)
        {
            if (consumedPairs != nullptr && t == 0) atomicAdd(consumedPairs + 3, 1ull); // (diagnostics: tiles that store their state)
            float4* state = layer.state + (size_t) view * scr.width * scr.height;
#pragma unroll
            for (int i = 0; i != PPL; ++i)
                if (onScreen[i]) state[(size_t) py * scr.width + px0 + i] = make_float4(PV::get(cr, i), PV::get(cg, i), PV::get(cb, i), PV::get(tr, i));
        }
    }
#ifdef GS_RASTER_PHASE_TRACE
    if (trace != nullptr && t == 0)
    {
        // ticks in barriers, waiting for the gather, staging, blending | voting, consumed entries
        trace[4 * (size_t) tile + 0] = (phase[0] << 32) | phase[1];
        trace[4 * (size_t) tile + 1] = (phase[2] << 32) | phase[3];
        trace[4 * (size_t) tile + 2] = phase[4];
        trace[4 * (size_t) tile + 3] = ((unsigned long long) blockIdx.x << 32) | (unsigned long long) consumed;
        trace = nullptr;
    }
#endif
    if (trace != nullptr && t == 0)
    {
        // per tile: start and end on the 100 MHz wall clock, list length, consumed entries
        trace[4 * (size_t) tile + 0] = traceStart;
        trace[4 * (size_t) tile + 1] = wall_clock64();
        trace[4 * (size_t) tile + 2] = (unsigned long long) count;
        trace[4 * (size_t) tile + 3] = ((unsigned long long) blockIdx.x << 32) | (unsigned long long) consumed;
    }

    // GaussianRender.cu:1007-1015: row 0 of the buffer is clip y = -1.
    if (py < scr.height)
    {
        uchar4* row = backBuffer + (size_t) py * scr.width + px0;
        if constexpr (PPL == 4)
        {
            if (px0 + 3 < scr.width && (scr.width & 3) == 0)
            {
                uint4 packed;
                uint32_t* pk = &packed.x;
#pragma unroll
                for (int p = 0; p != 4; ++p)
                    pk[p] = quantize8(PV::get(cr, p)) | (quantize8(PV::get(cg, p)) << 8) | (quantize8(PV::get(cb, p)) << 16) | 0xFF000000u;
                *reinterpret_cast<uint4*>(row) = packed;
                return;
            }
        }
#pragma unroll
        for (int p = 0; p != PPL; ++p)
            if (onScreen[p]) row[p] = make_uchar4((unsigned char) quantize8(PV::get(cr, p)), (unsigned char) quantize8(PV::get(cg, p)),
                                                  (unsigned char) quantize8(PV::get(cb, p)), 255);
    }
}

int g_rasterPixelsPerLane = 2;
unsigned long long* g_rasterTrace = nullptr;

} // namespace

void setRasterTrace(unsigned long long* deviceBufferOrNull) { g_rasterTrace = deviceBufferOrNull; }
void setRasterPixelsPerLane(int ppl) { g_rasterPixelsPerLane = (ppl == 1 || ppl == 2 || ppl == 4) ? ppl : 2; }
int rasterPixelsPerLane() { return g_rasterPixelsPerLane; }

void launchRasterize(hipStream_t stream, const Screen& scr, const int32_t* tileRange, const int32_t* values,
                     const float2* clipXY, const float4* conic, const float4* color, uchar4* backBuffer,
                     unsigned long long* consumedPairsOrNull, bool clearEmptyTiles, const RasterColourSource* lazyColours,
                     const float4* rasterRecord, int views, int viewSplats, const RasterLayer* layerOrNull, const RasterSchedule* scheduleOrNull)
{
    const int firstTile = scr.rowBegin * scr.tilesX;
    const int numTiles = (scr.rowEnd - scr.rowBegin) * scr.tilesX;
    if (numTiles <= 0) return;
    const int rows = scr.rowEnd - scr.rowBegin, bh = rows < GS_RASTER_BLOCK_H ? rows : GS_RASTER_BLOCK_H;
    const int blocks = ((scr.tilesX + GS_RASTER_BLOCK_W - 1) / GS_RASTER_BLOCK_W) * ((rows + bh - 1) / bh);
    const int gridTiles = ((blocks + 7) / 8) * 8 * GS_RASTER_BLOCK_W * bh; // every XCD walks the same number of blocks
    RasterColourSource lazy{};
    if (lazyColours != nullptr) lazy = *lazyColours;
    const RasterLayer noLayer{};
    const RasterSchedule schedule = scheduleOrNull != nullptr ? *scheduleOrNull : RasterSchedule{};
#define GS_LAUNCH_RASTER_L(PPL, EPAN, SH, LAYERED, LAYER)                                                                                 \
    hipLaunchKernelGGL((rasterizeTilesKernel<PPL, EPAN, SH, LAYERED>), dim3(gridTiles, views), dim3(256 / PPL), 0, stream, scr, firstTile, numTiles, \
                       tileRange, values, clipXY, conic, color, backBuffer, consumedPairsOrNull, clearEmptyTiles ? 1 : 0, lazy, g_rasterTrace, rasterRecord, \
                       views > 1 ? viewSplats : 0, LAYER, schedule)
#define GS_LAUNCH_RASTER(PPL, EPAN, SH) GS_LAUNCH_RASTER_L(PPL, EPAN, SH, false, noLayer)
    const bool epan = scr.epanechnikov != 0;
    if (layerOrNull != nullptr)
    {
        // depth layers: the default shape only (two pixels per lane); colours from the buffer or on demand
        const int sh = (lazyColours != nullptr && rasterCanEvaluateColours(lazy.degree, lazy.shCount)) ? lazy.degree : 0;
        const RasterLayer layer = *layerOrNull;
        switch (sh)
        {
        case 4: if (epan) GS_LAUNCH_RASTER_L(2, true, 4, true, layer); else GS_LAUNCH_RASTER_L(2, false, 4, true, layer); break;
        case 3: if (epan) GS_LAUNCH_RASTER_L(2, true, 3, true, layer); else GS_LAUNCH_RASTER_L(2, false, 3, true, layer); break;
        case 2: if (epan) GS_LAUNCH_RASTER_L(2, true, 2, true, layer); else GS_LAUNCH_RASTER_L(2, false, 2, true, layer); break;
        case 1: if (epan) GS_LAUNCH_RASTER_L(2, true, 1, true, layer); else GS_LAUNCH_RASTER_L(2, false, 1, true, layer); break;
        default: if (epan) GS_LAUNCH_RASTER_L(2, true, 0, true, layer); else GS_LAUNCH_RASTER_L(2, false, 0, true, layer); break;
        }
        return;
    }
    // on-demand colours: two pixels per lane only (the default shape), every degree the reference ingests (PlyParser.cpp:204-243)
    const int pixelsPerLane = effectivePixelsPerLane(); // (the calling renderer's own setting, or the process-wide default)
    if (lazyColours != nullptr && pixelsPerLane == 2 && rasterCanEvaluateColours(lazy.degree, lazy.shCount))
    {
        switch (lazy.degree)
        {
        case 1: if (epan) GS_LAUNCH_RASTER(2, true, 1); else GS_LAUNCH_RASTER(2, false, 1); break;
        case 2: if (epan) GS_LAUNCH_RASTER(2, true, 2); else GS_LAUNCH_RASTER(2, false, 2); break;
        case 3: if (epan) GS_LAUNCH_RASTER(2, true, 3); else GS_LAUNCH_RASTER(2, false, 3); break;
        default: if (epan) GS_LAUNCH_RASTER(2, true, 4); else GS_LAUNCH_RASTER(2, false, 4); break;
        }
        return;
    }
    switch (pixelsPerLane)
    {
    case 1:
        if (epan) GS_LAUNCH_RASTER(1, true, 0);
        else GS_LAUNCH_RASTER(1, false, 0);
        break;
    case 2:
        if (epan) GS_LAUNCH_RASTER(2, true, 0);
        else GS_LAUNCH_RASTER(2, false, 0);
        break;
    case 4:
    default:
        if (epan) GS_LAUNCH_RASTER(4, true, 0);
        else GS_LAUNCH_RASTER(4, false, 0);
        break;
    }
#undef GS_LAUNCH_RASTER
#undef GS_LAUNCH_RASTER_L
}

// ---- the start order of the next frame's tiles (gs_renderer_set_tile_start_order) ---------------------------------------------------
// The raster kernel's 60 us at BASELINE.json configs[2] are 36 us of full-chip work and a tail of tiles that consume several hundred
// entries and happen to start late: the grid is dispatched in screen order.  Which tiles are long is only known by blending them --
// but an interactive camera moves a little per frame (Demo.cpp:371-392), and the previous frame's consumption predicts this frame's:
// starting the tiles longest-first by it takes the kernel from 62.5 to 48.1 us at 1-degree steps (46.5 with the frame's own counts;
// tools/tile_order_experiment.py, profiles/r06_tile_order.txt).  This kernel -- one workgroup behind the frame's last raster launch --
// makes the table: every tile keeps the XCD of its 8 x 4 block (its neighbours' splats are in that L2), and within an XCD the tiles
// are started in descending order of the largest consumption in their 3 x 3 neighbourhood (a 1-degree step moves the image by about
// one tile), bucketed by vote chunks of 32.  A wrong prediction is WORSE than none (tiles that turn out long all start last: +20 % at
// 5 degrees on one view), so the table is only put in force when the last two frames' consumption differed by less than
// maxMismatch (sum |now - previous| / sum now): a camera that jumps falls back to the screen order by itself.
// (A first version ran as ONE workgroup: 35 us of latency behind every raster launch, more than the order saves.)
__global__ __launch_bounds__(1024) void tileStartOrderKernel(Screen scr, TileOrderArgs args) { tileStartOrderBody<1024>((int) blockIdx.x, scr, args); }

void launchTileStartOrder(hipStream_t stream, const Screen& scr, const TileOrderArgs& args)
{
    if (scr.rowEnd <= scr.rowBegin || args.slots <= 0) return;
    hipLaunchKernelGGL(tileStartOrderKernel, dim3(kOrderWorkgroups), dim3(1024), 0, stream, scr, args);
}

int rasterGridSlots(int tilesX, int rows)
{
    if (rows <= 0) return 0;
    const int bh = rows < GS_RASTER_BLOCK_H ? rows : GS_RASTER_BLOCK_H;
    const int blocks = ((tilesX + GS_RASTER_BLOCK_W - 1) / GS_RASTER_BLOCK_W) * ((rows + bh - 1) / bh);
    return ((blocks + 7) / 8) * 8 * GS_RASTER_BLOCK_W * bh;
}

bool rasterCanEvaluateColours(int degree, int shCount)
{
    return degree >= 1 && degree <= 4 && shCount == 3 * (degree + 1) * (degree + 1) && effectivePixelsPerLane() == 2;
}

} // namespace gs
