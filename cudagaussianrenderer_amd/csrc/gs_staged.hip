// gs_staged.hip -- the staged buildTileList / sortTileList pair (GaussianRender.cu:776-855) on the renderer's machinery.
//
// The reference's list has set semantics: buildTileListKernel appends its hits through atomics (GaussianRender.cu:674,689), so the
// order in which the pairs reach the list is not defined, and the only consumer of the unsorted list is cub's sort (Demo.cpp:442-458).
// A host that keeps the reference's eight calls therefore cannot tell in which order buildTileList emitted -- and this builder emits
// the pairs ALREADY IN THE ORDER THE SORT WOULD LEAVE THEM:
//   1. the splats' depth codes from the caller's positionClipSpaceZ (compactDepthKey, the lossless 27-bit re-coding of getKey's low
//      word) and a stable three-pass radix sort of the splats by it (gs_radix.hip);
//   2. count / emit over the depth-sorted splats (gs_tiles_fast.hip: the reference's overlap predicate bit for bit, guard band +
//      gs_overlap_exact.h), the emit pass carrying the first pass of the tile sort;
//   3. the remaining pass(es) of the stable tile sort, ending in the CALLER's value buffer,
//   4. every pair's quantised depth (getKey's low word, decoded from its splat's sorted code by the emit pass) carried along, and the
//      last scatter writing tile << 32 | depth beside the values into the caller's key buffer instead of the tile indices
//      (gs_radix.hip CARRY, RadixScratch::lastPassKeys64): no gather of positionClipSpaceZ per pair -- at 5 M splats that gather alone
//      took 0.9 ms (profiles/r05_staged_path.txt).
// That is the renderer's two-level sort: N splats through three passes and P pairs of 6 bytes through two,
// instead of P pairs of 12 bytes through six.  Equal keys (same tile, same depth code) are in splat-index order: the depth sort is
// stable over the index order and so is the tile sort -- the tie-break the oracle uses.
// sortTileList then LOOKS before it sorts: one pass over the keys finds out whether they are already in order on the bits it
// compares; a stable sort of such a list is the identity, so it returns with the selectors untouched.  Any other list (a caller's
// own, or one from the reference-order builder) is sorted by rocPRIM as before -- the check costs it one read of the keys.
// gs_set_staged_builder(GS_STAGED_BUILDER_REFERENCE_ORDER) brings back gs_tiles.hip's (splat, local tile) emission order and the
// unconditional sort.
#include "gs_depth_key.h"
#include "gs_state.h"

namespace gs
{
namespace
{

__global__ __launch_bounds__(256) void stagedDepthKeysKernel(int n, const float* __restrict__ clipZ, uint32_t* __restrict__ keys,
                                                             int32_t* __restrict__ order)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = compactDepthKey(clipZ[i]);
    order[i] = i;
}

// One lane = two consecutive keys (16 bytes) + the key in front of them; a wave that sees a descent on the compared bits stores the
// epoch of this check (plain store: every writer writes the same value).
__global__ __launch_bounds__(256) void stagedSortedCheckKernel(int size, const uint64_t* __restrict__ keys, uint64_t mask, int epoch,
                                                               int* __restrict__ unsortedEpoch, int* __restrict__ unsortedEpochHostMapped)
{
    const int first = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    bool descent = false;
    if (first < size)
    {
        // (8-byte loads: the caller's key buffer need not be 16-byte aligned)
        const uint64_t a = keys[first] & mask;
        const uint64_t b = (first + 1 < size ? keys[first + 1] : keys[first]) & mask;
        const uint64_t before = first > 0 ? (keys[first - 1] & mask) : 0ull;
        descent = before > a || a > b;
    }
    // The verdict goes straight into a pinned, device-mapped host word (no copy behind the kernel).  A list in order writes nothing;
    // of an unordered one only the waves that have not yet seen the device-side word set cross the bus (benign race: all write `epoch`).
    if (__any(descent) && (threadIdx.x & (kWave - 1)) == 0 && __builtin_nontemporal_load(unsortedEpoch) != epoch)
    {
        *unsortedEpoch = epoch;
        *unsortedEpochHostMapped = epoch;
    }
}

// The sorted builder's hidden scratch: 16 bytes per splat and, per entry of the CALLER's list capacity, two tile-index halves (2 or 4
// bytes each), one value half and two depth halves -- 16 to 20 bytes per entry, 0.64-0.8 GB at a capacity of 40 M (the footprint the
// header states).  The reference's builder needs none of it, so running out of device memory here is no reason to end the process
// (ADVICE r05): false = could not be had -- everything this call allocated is given back and the caller emits in the reference's order.
int64_t g_stagedScratchLimit = 0; // gs_debug_set_staged_scratch_limit: allocations beyond it fail as if the device were out of memory (tests)
bool ensureStagedScratch(StagedScratch& s, int splats, int capacityPairs, int tileKeyBytes)
{
    const auto release = [](auto*& p) {
        if (p) (void) hipFree(p);
        p = nullptr;
    };
    const auto giveUp = [&]() {
        for (int i = 0; i != 2; ++i)
        {
            release(s.depthKey[i]);
            release(s.order[i]);
            release(s.tileKeys[i]);
            release(s.depth[i]);
        }
        release(s.values);
        s.capacitySplats = 0;
        s.capacityPairBytes = 0;
        s.capacityPairs = 0;
        (void) hipGetLastError(); // (the failed allocation's sticky error)
        return false;
    };
    const auto alloc = [](auto*& p, size_t bytes) {
        p = nullptr;
        if (g_stagedScratchLimit > 0 && (int64_t) bytes > g_stagedScratchLimit) return false;
        return hipMalloc((void**) &p, std::max<size_t>(bytes, 16)) == hipSuccess;
    };
    if (splats > s.capacitySplats)
    {
        for (int i = 0; i != 2; ++i)
        {
            release(s.depthKey[i]);
            release(s.order[i]);
            if (!alloc(s.depthKey[i], sizeof(uint32_t) * (size_t) splats) || !alloc(s.order[i], sizeof(int32_t) * (size_t) splats)) return giveUp();
        }
        s.capacitySplats = splats;
    }
    const int64_t keyBytes = (int64_t) std::max(capacityPairs, 1) * tileKeyBytes;
    if (keyBytes > s.capacityPairBytes)
    {
        for (int i = 0; i != 2; ++i)
        {
            release(s.tileKeys[i]);
            if (!alloc(s.tileKeys[i], (size_t) keyBytes)) return giveUp();
        }
        s.capacityPairBytes = keyBytes;
    }
    if (capacityPairs > s.capacityPairs)
    {
        release(s.values);
        if (!alloc(s.values, sizeof(int32_t) * (size_t) capacityPairs)) return giveUp();
        for (int i = 0; i != 2; ++i)
        {
            release(s.depth[i]);
            if (!alloc(s.depth[i], sizeof(uint32_t) * (size_t) capacityPairs)) return giveUp();
        }
        s.capacityPairs = capacityPairs;
    }
    return true;
}

void ensureSortedCheck(StagedScratch& s)
{
    if (s.unsortedEpoch != nullptr) return;
    GS_CHECK(hipMalloc((void**) &s.unsortedEpoch, 2 * sizeof(int)));
    GS_CHECK(hipMemset(s.unsortedEpoch, 0, 2 * sizeof(int)));
    GS_CHECK(hipHostMalloc((void**) &s.unsortedEpochHost, sizeof(int), hipHostMallocMapped));
    *s.unsortedEpochHost = 0;
    if (hipHostGetDevicePointer((void**) &s.unsortedEpochHostMapped, s.unsortedEpochHost, 0) != hipSuccess) s.unsortedEpochHostMapped = nullptr;
}

} // namespace

int32_t stagedBuildTileListSorted(GsTimer* timer, int32_t tileListCapacity)
{
    const Screen scr = currentScreen();
    const int n = g.global.splatCount;
    const int capacity = std::max(tileListCapacity, 0);
    StagedScratch& st = g.staged;
    const int tBits = tileBits(scr.totalTiles);
    const int tileKeyBytes = scr.totalTiles > 65536 ? 4 : 2;
    // (before the timed interval: allocation is not the stage's work.  No memory for the scratch: -1, the caller emits in the reference's order)
    if (n > 0 && !ensureStagedScratch(st, n, capacity, tileKeyBytes))
    {
        static bool told = false;
        if (!told) printf("gsplat_hip: no device memory for the sorted tile-list builder's scratch (%d splats, capacity %d): emitting in the reference's order\n", n, capacity);
        told = true;
        return -1;
    }
    gs_timer_start(timer);
    if (n <= 0)
    {
        ensureTileListScratch(g.scratch, 0);
        *g.scratch.totalHost = 0;
    }
    else
    {
        const float* clipZ = g.global.positionClipSpaceZ;
        const float2* clipXY = (const float2*) g.global.positionClipSpaceXY;
        const float4* ellipse = (const float4*) g.global.screenEllipse;
        // (1) splats by depth
        hipLaunchKernelGGL(stagedDepthKeysKernel, dim3((n + 255) / 256), dim3(256), 0, g.stream, n, clipZ, st.depthKey[0], st.order[0]);
        g.radix.countDevice = nullptr;
        g.radix.rangeTableForLastPass = nullptr;
        const int orderSel = launchRadixSortPairs(g.stream, g.radix, 4, st.depthKey[0], st.depthKey[1], st.order[0], st.order[1], n, 0, kDepthKeyBits);
        const uint32_t* order = (const uint32_t*) st.order[orderSel];
        // (2) count / emit in depth order; the emit pass carries the tile sort's first pass when that sort is a two-pass one
        const int sortBits = effectiveListFusion() ? tileListFusedSortBits(tBits, tileKeyBytes) : 0;
        g.scratch.tileRecord = nullptr;
        g.scratch.countDevice = nullptr;
        g.scratch.depthSorted = st.depthKey[orderSel]; // the emit pass decodes every pair's quantised depth from its splat's sorted code ...
        g.scratch.carryOut = st.depth[0];              // ... and leaves it beside the pair (no gather of clipZ anywhere)
        launchTileListFastCount(g.stream, n, scr, order, clipXY, ellipse, g.scratch, sortBits);
        // The passes that are left flip between two buffer pairs; the value half they END in is the caller's list.
        const int passes = radixSortPasses(sortBits, tBits);
        int32_t* caller = g.tileList.values;
        int32_t* v0 = (passes % 2 == 0) ? caller : st.values;
        int32_t* v1 = (passes % 2 == 0) ? st.values : caller;
        launchTileListFastEmit(g.stream, n, scr, order, clipXY, ellipse, g.scratch, st.tileKeys[0], tileKeyBytes, v0, capacity, 0, sortBits);
        g.scratch.depthSorted = nullptr;
        g.scratch.carryOut = nullptr;
        if (capacity > 0)
        {
            // (3) the rest of the stable tile sort; the list length stays on the device (the pair total the emit pass left, low word of
            // the little-endian int64), bounded by the capacity
            const int* sizeDevice = (const int*) g.scratch.total;
            // (4) ... whose last scatter writes the reference's 64-bit keys (tile | quantised depth of the entry's splat) into the caller's
            // key buffer instead of the tile indices
            st.radixPairs.countDevice = sizeDevice;
            st.radixPairs.rangeTableForLastPass = nullptr;
            st.radixPairs.lastPassKeys64 = g.tileList.keys;
            st.radixPairs.carry[0] = st.depth[0];
            st.radixPairs.carry[1] = st.depth[1];
            st.radixPairs.carrySel = 0;
            launchRadixSortPairs(g.stream, st.radixPairs, tileKeyBytes, st.tileKeys[0], st.tileKeys[1], v0, v1, capacity, sortBits, tBits);
            st.radixPairs.countDevice = nullptr;
            st.radixPairs.lastPassKeys64 = nullptr;
            st.radixPairs.carry[0] = st.radixPairs.carry[1] = nullptr;
        }
    }
    gs_timer_stop(timer);
    // The reference reads the pair counter back here too (GaussianRender.cu:793), a host sync point.
    GS_CHECK(hipStreamSynchronize(g.stream));
    reportLaunch("buildTileList");
    g.lastDemand = *g.scratch.totalHost;
    return (int32_t) std::min<int64_t>(g.lastDemand, (int64_t) capacity); // GaussianRender.cu:800-801
}

bool stagedListIsSorted(int32_t size, const uint64_t* keys, int endBit)
{
    if (size <= 1) return true;
    StagedScratch& st = g.staged;
    ensureSortedCheck(st);
    st.epoch = st.epoch == 0x7FFFFFFF ? 1 : st.epoch + 1;
    const uint64_t mask = endBit >= 64 ? ~0ull : ((1ull << endBit) - 1ull);
    const int lanes = (size + 1) / 2;
    // (no mapped word: the verdict lands in a device-side scratch word next to the epoch word and is copied behind the kernel)
    int* verdict = st.unsortedEpochHostMapped != nullptr ? st.unsortedEpochHostMapped : st.unsortedEpoch + 1;
    hipLaunchKernelGGL(stagedSortedCheckKernel, dim3((lanes + 255) / 256), dim3(256), 0, g.stream, size, keys, mask, st.epoch, st.unsortedEpoch, verdict);
    if (st.unsortedEpochHostMapped == nullptr) GS_CHECK(hipMemcpyAsync(st.unsortedEpochHost, verdict, sizeof(int), hipMemcpyDeviceToHost, g.stream));
    GS_CHECK(hipStreamSynchronize(g.stream));
    return *st.unsortedEpochHost != st.epoch;
}

} // namespace gs

using namespace gs;

// Everything the staged calls of THIS host thread allocated behind the scenes (builder scratch, sort tables, the sorted check's
// words): freed; the next staged call allocates again.  For hosts that run the staged loop on short-lived threads.
extern "C" void gs_release_thread_scratch(void)
{
    GS_CHECK(hipStreamSynchronize(g.stream));
    freeTileListScratch(g.scratch);
    freeRadixScratch(g.radix);
    StagedScratch& st = g.staged;
    freeRadixScratch(st.radixPairs);
    for (int i = 0; i != 2; ++i)
    {
        if (st.depthKey[i]) (void) hipFree(st.depthKey[i]);
        if (st.order[i]) (void) hipFree(st.order[i]);
        if (st.tileKeys[i]) (void) hipFree(st.tileKeys[i]);
        if (st.depth[i]) (void) hipFree(st.depth[i]);
    }
    if (st.values) (void) hipFree(st.values);
    if (st.unsortedEpoch) (void) hipFree(st.unsortedEpoch);
    if (st.unsortedEpochHost) (void) hipHostFree(st.unsortedEpochHost);
    st = StagedScratch{};
    freeStagedTileOrder(g.stagedTileOrder);
}

extern "C" void gs_debug_set_staged_scratch_limit(int64_t maxBytesPerAllocationOr0) { g_stagedScratchLimit = maxBytesPerAllocationOr0 > 0 ? maxBytesPerAllocationOr0 : 0; }

extern "C" void gs_set_staged_builder(int mode)
{
    cfg.stagedBuilder = mode == GS_STAGED_BUILDER_REFERENCE_ORDER ? GS_STAGED_BUILDER_REFERENCE_ORDER : GS_STAGED_BUILDER_SORTED;
}
extern "C" int gs_get_staged_builder(void) { return cfg.stagedBuilder; }
