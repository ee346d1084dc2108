// atomic_rate.hip -- how fast can 5.5 M (splat, tile) hits take a slot from 8 160 per-tile cursors with global atomics?
// (decides whether tile lists can be filled by unordered fetch-adds instead of a radix sort; diagnostics, not product code)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/atomic_rate.hip -o tools/micro/atomic_rate && tools/micro/atomic_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                             \
    do                                                                       \
    {                                                                        \
        hipError_t e_ = (x);                                                 \
        if (e_ != hipSuccess)                                                \
        {                                                                    \
            printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); \
            return 1;                                                        \
        }                                                                    \
    } while (0)

__device__ __forceinline__ unsigned hash(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// every "splat" touches a 2x3 block of tiles around a centre tile; centres are skewed (half of them fall on 1/16 of the tiles)
// MODE 0: atomicAdd without using the result (count pass); 1: fetch-add, the slot decides where 8 bytes are written (scatter pass);
// 2: like 1 but lanes of a wave that hit the same tile take ONE fetch-add together (match by ballot loop)
template <int MODE>
__global__ __launch_bounds__(256) void hitKernel(int splats, int tilesX, int tilesY, unsigned* __restrict__ cursor,
                                                  const unsigned* __restrict__ base, uint2* __restrict__ out, int localOrder)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= splats) return;
    // localOrder: consecutive splats are spatial neighbours (Morton-ordered scene); else random
    const unsigned h = hash(localOrder ? (unsigned) (i >> 5) : (unsigned) i);
    const unsigned h2 = hash(h ^ 0x9e3779b9u ^ (localOrder ? (unsigned) i : 0u));
    int cx = (h & 0xffff) % tilesX, cy = ((h >> 16) & 0xffff) % tilesY;
    if (h2 & 1)
    {
        cx = tilesX / 2 - tilesX / 8 + (int) ((h & 0xffff) % (tilesX / 4));
        cy = tilesY / 2 - tilesY / 8 + (int) (((h >> 16) & 0xffff) % (tilesY / 4));
    }
    if (localOrder)
    {
        cx = min(tilesX - 1, cx + (int) (h2 >> 8 & 1));
        cy = min(tilesY - 1, cy + (int) (h2 >> 9 & 1));
    }
#pragma unroll
    for (int dy = 0; dy != 2; ++dy)
#pragma unroll
        for (int dx = 0; dx != 3; ++dx)
        {
            const int tx = min(cx + dx, tilesX - 1), ty = min(cy + dy, tilesY - 1);
            const int tile = ty * tilesX + tx;
            if (MODE == 0) atomicAdd(&cursor[tile], 1u);
            else if (MODE == 1)
            {
                const unsigned slot = atomicAdd(&cursor[tile], 1u);
                out[base[tile] + slot] = make_uint2((unsigned) i, h);
            }
            else
            {
                // wave-aggregated: peel off one distinct tile per iteration
                unsigned long long todo = __ballot(1);
                unsigned slot = 0;
                while (todo)
                {
                    const int leader = __ffsll((long long) todo) - 1;
                    const int leaderTile = __shfl(tile, leader);
                    const unsigned long long same = __ballot(tile == leaderTile) & todo;
                    unsigned first = 0;
                    if ((int) (threadIdx.x & 63) == leader) first = atomicAdd(&cursor[leaderTile], (unsigned) __popcll(same));
                    first = __shfl(first, leader);
                    if (tile == leaderTile && ((same >> (threadIdx.x & 63)) & 1ull))
                        slot = first + (unsigned) __popcll(same & ((1ull << (threadIdx.x & 63)) - 1ull));
                    todo &= ~same;
                }
                out[base[tile] + slot] = make_uint2((unsigned) i, h);
            }
        }
}

int main()
{
    const int splats = 910000, tilesX = 120, tilesY = 68, tiles = tilesX * tilesY; // x6 hits = 5.46 M
    unsigned *cursor = nullptr, *base = nullptr;
    uint2* out = nullptr;
    CHECK(hipMalloc((void**) &cursor, sizeof(unsigned) * tiles));
    CHECK(hipMalloc((void**) &base, sizeof(unsigned) * tiles));
    CHECK(hipMalloc((void**) &out, sizeof(uint2) * (size_t) splats * 6));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<unsigned> counts(tiles), bases(tiles);
    for (int local = 0; local != 2; ++local)
    {
        // count pass (also gives the segment bases)
        float ms = 0;
        for (int rep = 0; rep != 3; ++rep)
        {
            CHECK(hipMemset(cursor, 0, sizeof(unsigned) * tiles));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(hitKernel<0>, dim3((splats + 255) / 256), dim3(256), 0, 0, splats, tilesX, tilesY, cursor, base, out, local);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        CHECK(hipMemcpy(counts.data(), cursor, sizeof(unsigned) * tiles, hipMemcpyDeviceToHost));
        unsigned run = 0, mx = 0;
        for (int t = 0; t != tiles; ++t)
        {
            bases[t] = run;
            run += counts[t];
            mx = counts[t] > mx ? counts[t] : mx;
        }
        CHECK(hipMemcpy(base, bases.data(), sizeof(unsigned) * tiles, hipMemcpyHostToDevice));
        printf("%s order: %u hits, longest list %u\n  count (atomicAdd, no return)        %7.1f us\n", local ? "neighbour" : "random", run, mx, ms * 1e3);
        for (int mode = 1; mode <= 2; ++mode)
        {
            for (int rep = 0; rep != 3; ++rep)
            {
                CHECK(hipMemset(cursor, 0, sizeof(unsigned) * tiles));
                CHECK(hipEventRecord(e0));
                if (mode == 1) hipLaunchKernelGGL(hitKernel<1>, dim3((splats + 255) / 256), dim3(256), 0, 0, splats, tilesX, tilesY, cursor, base, out, local);
                else hipLaunchKernelGGL(hitKernel<2>, dim3((splats + 255) / 256), dim3(256), 0, 0, splats, tilesX, tilesY, cursor, base, out, local);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("  scatter (%s) %7.1f us\n", mode == 1 ? "fetch-add per hit, 8-B store    " : "wave-aggregated fetch-add, store", ms * 1e3);
        }
    }
    return 0;
}
