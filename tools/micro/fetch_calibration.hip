// fetch_calibration.hip -- what does rocprofv3's FETCH_SIZE tally for the access patterns of the raster kernel?
// MI355X_MICROARCH.md ("HBM / rocprofv3"): FETCH_SIZE reports exactly half of a wide coalesced streaming read on gfx950 and "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern before trusting an absolute".  The raster
// kernel gathers, per consumed list entry, a 4-byte list index (streamed), a 32-byte record, a 16-byte position and a 192-byte
// coefficient row at random splat indices -- none of them a streaming read.  This program issues a KNOWN number of each, every access
// to a line nobody touched before (a bijection over a table far larger than L2 and the 256 MiB Infinity Cache), one kernel per
// pattern; run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` the per-kernel FETCH_SIZE divided by the bytes the lanes asked for is
// the pattern's factor (tools/fetch_calibration.sh prints the table).  Diagnostics, not product code.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/fetch_calibration.hip -o /tmp/fetch_calibration
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

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

// record r of the pattern -> slot: a bijection on [0, slots) (slots a power of two, multiplier odd), so no record is read twice
__device__ __forceinline__ unsigned long long slotOf(unsigned long long r, unsigned long long slots) { return (r * 0x9E3779B97F4A7C15ull + 12345ull) & (slots - 1ull); }

// streaming: lane i reads the i-th float4 (what the guide calibrated: FETCH_SIZE = 1/2 of the bytes)
__global__ __launch_bounds__(256) void streamKernel(const float4* __restrict__ table, unsigned long long count, float* __restrict__ sink)
{
    const unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float4 v = table[i];
    if (v.x + v.y + v.z + v.w == 12345.678f) sink[0] = v.x; // (never true: keeps the load)
}

// gather: lane i reads QUADS consecutive float4 (16 * QUADS bytes) at a slot of STRIDE bytes picked by the bijection
// (PACKED only tells the two launches of a record size apart in the profiler's kernel names)
template <int QUADS, bool PACKED>
__global__ __launch_bounds__(256) void gatherKernel(const float4* __restrict__ table, unsigned long long count, unsigned long long slots, int strideQuads,
                                                     float* __restrict__ sink)
{
    const unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float4* p = table + slotOf(i, slots) * (unsigned long long) strideQuads;
    float acc = 0.0f;
#pragma unroll
    for (int q = 0; q != QUADS; ++q)
    {
        const float4 v = p[q];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// 4-byte gather (a list index read at random: not what the raster does -- its indices stream -- but the other end of the scale)
__global__ __launch_bounds__(256) void gatherWordKernel(const float* __restrict__ table, unsigned long long count, unsigned long long slots, int strideWords,
                                                         float* __restrict__ sink)
{
    const unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float v = table[slotOf(i, slots) * (unsigned long long) strideWords];
    if (v == 12345.678f) sink[0] = v;
}

int main()
{
    const unsigned long long tableBytes = 4ull << 30; // 4 GiB: 16x the Infinity Cache
    float4* table = nullptr;
    float* sink = nullptr;
    CHECK(hipMalloc((void**) &table, tableBytes));
    CHECK(hipMalloc((void**) &sink, 256));
    CHECK(hipMemset(table, 0, tableBytes));
    CHECK(hipDeviceSynchronize());
    const unsigned long long count = 4ull << 20; // accesses per pattern
    const dim3 block(256), grid((unsigned) ((count + 255) / 256));
    // every pattern reads each of its slots once; slots are spaced so that no two records share a 128-byte line unless the record
    // itself is what is being measured (the "packed" variants put records side by side, as the scene's arrays do)
    // 1. streaming 16 B per lane over 64 MiB
    hipLaunchKernelGGL(streamKernel, grid, block, 0, 0, table, count, sink);
    // 2. 16-byte gathers (the position): packed (16-byte slots) and alone in a 256-byte slot
    hipLaunchKernelGGL((gatherKernel<1, true>), grid, block, 0, 0, table, count, tableBytes / 16, 1, sink);
    hipLaunchKernelGGL((gatherKernel<1, false>), grid, block, 0, 0, table, count, tableBytes / 256, 16, sink);
    // 3. 32-byte records (centre + conic): packed and alone
    hipLaunchKernelGGL((gatherKernel<2, true>), grid, block, 0, 0, table, count, tableBytes / 32, 2, sink);
    hipLaunchKernelGGL((gatherKernel<2, false>), grid, block, 0, 0, table, count, tableBytes / 256, 16, sink);
    // 4. 192-byte coefficient rows: packed (192-byte slots: rows straddle 128-byte lines as the [N][48] array does) and in 256-byte slots
    hipLaunchKernelGGL((gatherKernel<12, true>), grid, block, 0, 0, table, count, 1ull << 24, 12, sink); // 2^24 slots of 192 bytes = 3 GiB
    hipLaunchKernelGGL((gatherKernel<12, false>), grid, block, 0, 0, table, count, tableBytes / 256, 16, sink);
    // 5. 64- and 128-byte records alone in 256-byte slots (the line sizes the counter may think in)
    hipLaunchKernelGGL((gatherKernel<4, false>), grid, block, 0, 0, table, count, tableBytes / 256, 16, sink);
    hipLaunchKernelGGL((gatherKernel<8, false>), grid, block, 0, 0, table, count, tableBytes / 256, 16, sink);
    // 6. 4-byte gathers alone in 256-byte slots
    hipLaunchKernelGGL(gatherWordKernel, grid, block, 0, 0, (const float*) table, count, tableBytes / 256, 64, sink);
    CHECK(hipDeviceSynchronize());
    printf("accesses per pattern: %llu\n", count);
    printf("launch order: stream16, gather16 packed, gather16 alone, gather32 packed, gather32 alone, gather192 packed, gather192 alone, gather64 alone, "
           "gather128 alone, gather4 alone\n");
    CHECK(hipFree(table));
    CHECK(hipFree(sink));
    return 0;
}
