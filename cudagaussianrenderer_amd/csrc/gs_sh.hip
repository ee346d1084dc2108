// gs_sh.hip -- view-dependent colour from spherical harmonics.
//
// Replaces evaluateSphericalHarmonicsKernel + sphericalHarmonics() (GaussianRender.cu:62-182).  Same observable
// contract: colour = clamp(sum_k Y_k(normalize(camera - splat)) * c_k + 0.5, 0, 1), alpha = position.w.
//
// Two kernels share one evaluation routine (identical arithmetic, so their colours are bit-identical):
//   * sphericalHarmonicsKernel  -- all splats, coefficients in the blocked layout written by
//     realignSphericalHarmonics (Demo.cpp:160-194): group g of 256 splats, coefficient j, lane i at
//     sh[g*256*S + j*256 + i].  Pure HBM streaming (16 + 4*S read, 16 written per splat; 224 B at degree 3).
//     One 64-lane wave owns one 256-splat group and each lane owns FOUR consecutive splats of it, so every
//     coefficient row is one 16-byte load per lane (1 KiB per wave instruction); a level's loads are all issued
//     before their first use.
//   * sphericalHarmonicsGatherKernel -- only the splats of a device-side index list (the band-culled list of a
//     multi-GPU rank), coefficients in the plain [splat][S] layout parsePly produces, 192 contiguous bytes per splat
//     at degree 3, so a gathered splat costs three 64-byte lines instead of 48 scattered 4-byte accesses.
#include "gs_sh_eval.h"

// Both kernels must produce the same bits for the same splat: no compiler-chosen contraction, only the explicit fmaf.
#pragma clang fp contract(off)

namespace gs
{
namespace
{

constexpr int kGroup = GS_SH_GROUP_SIZE;  // 256 splats per coefficient block
constexpr int kSplatsPerLane = 4;         // 256 / 64
constexpr int kWavesPerBlock = 4;         // 256 threads

using namespace sheval;

__device__ __forceinline__ float4 finalColour(const Rgb4& rgb, int i, float alpha)
{
    const float rr[4] = {rgb.r.x, rgb.r.y, rgb.r.z, rgb.r.w};
    const float gg[4] = {rgb.g.x, rgb.g.y, rgb.g.z, rgb.g.w};
    const float bb[4] = {rgb.b.x, rgb.b.y, rgb.b.z, rgb.b.w};
    // clamp(result + 0.5, 0, 1) (GaussianRender.cu:154); alpha = position.w (:168,180)
    return make_float4(__saturatef(rr[i] + 0.5f), __saturatef(gg[i] + 0.5f), __saturatef(bb[i] + 0.5f), alpha);
}

template <int L>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void sphericalHarmonicsKernel(int count, int shCount,
                                                                                   const float4* __restrict__ position,
                                                                                   const float* __restrict__ sh,
                                                                                   float3 camArg, float4* __restrict__ color,
                                                                                   const FrameParams* __restrict__ params)
{
    const float3 cam = params ? make_float3(params->cameraPosition[0], params->cameraPosition[1], params->cameraPosition[2]) : camArg;
    const int lane = threadIdx.x & (kWave - 1);
    const int group = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int first = group * kGroup + lane * kSplatsPerLane; // first of this lane's four splats
    if (first >= count) return;

    // Positions of the four owned splats (tail lanes clamp the index; their results are not stored).
    float4 p[4];
#pragma unroll
    for (int i = 0; i != 4; ++i) p[i] = position[min(first + i, count - 1)];

    // Row j of channel c is one float4 (this lane's four splats) at rows[(3j+c)*64].
    const float4* rows = reinterpret_cast<const float4*>(sh + (size_t) group * kGroup * shCount) + lane;
    const auto load = [rows](int j, int c) { return rows[(3 * j + c) * (kGroup / 4)]; };
    const Rgb4 rgb = evaluateFour<L>(p, cam, load);
#pragma unroll
    for (int i = 0; i != 4; ++i)
        if (first + i < count) color[first + i] = finalColour(rgb, i, p[i].w);
}

// One list entry per lane: the splat's coefficient row starts at shPerSplat + idx * rowStride (the renderer pads its rows of
// S = 3, 12, 27, 48, 75 floats to 4, 12, 28, 48, 76 -- gs_sh_eval.h rowStrideFloats -- so every row is read as float4; an unpadded
// stride falls back to scalar loads), so a degree-3 splat is twelve 16-byte loads, all issued before the first use.
template <int L>
__global__ __launch_bounds__(256) void sphericalHarmonicsGatherKernel(int countHost, const int* __restrict__ countDevice, int rowStride,
                                                                       const int32_t* __restrict__ indices,
                                                                       const float4* __restrict__ position,
                                                                       const float* __restrict__ shPerSplat, float3 camArg,
                                                                       float4* __restrict__ color, const FrameParams* __restrict__ params)
{
    const float3 cam = params ? make_float3(params->cameraPosition[0], params->cameraPosition[1], params->cameraPosition[2]) : camArg;
    const int count = countDevice ? min(*countDevice, countHost) : countHost;
    const int entry = blockIdx.x * blockDim.x + threadIdx.x;
    if (entry >= count) return;
    const int idx = indices[entry];
    const float* row = shPerSplat + (size_t) idx * rowStride;
    color[idx] = colourFromRow<L>(row, rowStride, position[idx], cam);
}

} // namespace

void launchSphericalHarmonics(hipStream_t stream, int count, int degree, int shCount, const float4* position, const float* sh,
                              float3 cameraPosition, float4* color, const FrameParams* paramsDevice)
{
    if (count <= 0) return;
    const int groups = (count + kGroup - 1) / kGroup;
    const dim3 grid((groups + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
    switch (degree)
    {
    case 0: hipLaunchKernelGGL(sphericalHarmonicsKernel<0>, grid, block, 0, stream, count, shCount, position, sh, cameraPosition, color, paramsDevice); break;
    case 1: hipLaunchKernelGGL(sphericalHarmonicsKernel<1>, grid, block, 0, stream, count, shCount, position, sh, cameraPosition, color, paramsDevice); break;
    case 2: hipLaunchKernelGGL(sphericalHarmonicsKernel<2>, grid, block, 0, stream, count, shCount, position, sh, cameraPosition, color, paramsDevice); break;
    case 3: hipLaunchKernelGGL(sphericalHarmonicsKernel<3>, grid, block, 0, stream, count, shCount, position, sh, cameraPosition, color, paramsDevice); break;
    default: hipLaunchKernelGGL(sphericalHarmonicsKernel<4>, grid, block, 0, stream, count, shCount, position, sh, cameraPosition, color, paramsDevice); break;
    }
}

void launchSphericalHarmonicsGather(hipStream_t stream, int countHost, const int* countDevice, int degree, int rowStride,
                                    const int32_t* indices, const float4* position, const float* shPerSplat, float3 cameraPosition,
                                    float4* color, const FrameParams* paramsDevice)
{
    if (countHost <= 0) return;
    const dim3 grid((countHost + 255) / 256), block(256);
#define GS_LAUNCH_GATHER(L)                                                                                                          \
    hipLaunchKernelGGL(sphericalHarmonicsGatherKernel<L>, grid, block, 0, stream, countHost, countDevice, rowStride, indices, position, \
                       shPerSplat, cameraPosition, color, paramsDevice)
    switch (degree)
    {
    case 0: GS_LAUNCH_GATHER(0); break;
    case 1: GS_LAUNCH_GATHER(1); break;
    case 2: GS_LAUNCH_GATHER(2); break;
    case 3: GS_LAUNCH_GATHER(3); break;
    default: GS_LAUNCH_GATHER(4); break;
    }
#undef GS_LAUNCH_GATHER
}

} // namespace gs
