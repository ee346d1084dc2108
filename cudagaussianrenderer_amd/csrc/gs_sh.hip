// gs_sh.hip -- view-dependent colour from spherical harmonics.
//
// Replaces evaluateSphericalHarmonicsKernel + sphericalHarmonics() (GaussianRender.cu:62-182).
// Same observable contract: colour = clamp(sum_k Y_k(normalize(camera - splat)) * c_k + 0.5, 0, 1),
// alpha = position.w, coefficients in the blocked layout written by realignSphericalHarmonics
// (Demo.cpp:160-194): group g of 256 splats, coefficient j, lane i at sh[g*256*S + j*256 + i].
//
// CDNA4 mapping: pure HBM streaming (16 + 4*S read, 16 written per splat; 224 B at degree 3).
// One 64-lane wave owns one 256-splat group and each lane owns FOUR consecutive splats of it, so
// every coefficient row is read with one 16-byte load per lane (1 KiB per wave instruction) instead
// of four 4-byte ones.  All loads of a level are issued before the first use.
#include "gs_internal.h"

namespace gs
{
namespace
{

constexpr int kGroup = GS_SH_GROUP_SIZE;  // 256 splats per coefficient block
constexpr int kSplatsPerLane = 4;         // 256 / 64
constexpr int kWavesPerBlock = 4;         // 256 threads

struct Rgb4
{
    float4 r, g, b; // one float per owned splat
};

__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c)
{
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 splat4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 scale4(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 affine4(float m, float4 a, float c) // m*a + c
{
    return make_float4(fmaf(m, a.x, c), fmaf(m, a.y, c), fmaf(m, a.z, c), fmaf(m, a.w, c));
}

// Accumulate the rows [first, first+count) of the basis table into rgb. `rows` points at coefficient
// 0 of this lane's four splats; row j of channel c is a float4 at rows[(3j+c)*kGroup/4].
template <int FIRST, int COUNT>
__device__ __forceinline__ void accumulateLevel(const float4* __restrict__ rows, const float4 (&basis)[COUNT], Rgb4& rgb)
{
    float4 cr[COUNT], cg[COUNT], cb[COUNT];
#pragma unroll
    for (int j = 0; j != COUNT; ++j)
    {
        cr[j] = rows[(3 * (FIRST + j) + 0) * (kGroup / 4)];
        cg[j] = rows[(3 * (FIRST + j) + 1) * (kGroup / 4)];
        cb[j] = rows[(3 * (FIRST + j) + 2) * (kGroup / 4)];
    }
#pragma unroll
    for (int j = 0; j != COUNT; ++j)
    {
        rgb.r = fma4(basis[j], cr[j], rgb.r);
        rgb.g = fma4(basis[j], cg[j], rgb.g);
        rgb.b = fma4(basis[j], cb[j], rgb.b);
    }
}

template <int L>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void sphericalHarmonicsKernel(int count, int shCount,
                                                                                   const float4* __restrict__ position,
                                                                                   const float* __restrict__ sh,
                                                                                   float3 cam, float4* __restrict__ color)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int group = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int first = group * kGroup + lane * kSplatsPerLane; // first of this lane's four splats
    if (first >= count) return;

    // Positions of the four owned splats (tail lanes clamp the index; their results are not stored).
    float4 p[4];
#pragma unroll
    for (int i = 0; i != 4; ++i) p[i] = position[min(first + i, count - 1)];

    // rayDir = normalize(camera - splat)  (GaussianRender.cu:171), component-wise over the four splats.
    float4 dx = make_float4(cam.x - p[0].x, cam.x - p[1].x, cam.x - p[2].x, cam.x - p[3].x);
    float4 dy = make_float4(cam.y - p[0].y, cam.y - p[1].y, cam.y - p[2].y, cam.y - p[3].y);
    float4 dz = make_float4(cam.z - p[0].z, cam.z - p[1].z, cam.z - p[2].z, cam.z - p[3].z);
    float4 len2 = fma4(dz, dz, fma4(dy, dy, mul4(dx, dx)));
    float4 inv = make_float4(1.0f / sqrtf(len2.x), 1.0f / sqrtf(len2.y), 1.0f / sqrtf(len2.z), 1.0f / sqrtf(len2.w));
    const float4 x = mul4(dx, inv), y = mul4(dy, inv), z = mul4(dz, inv);

    const float4* rows = reinterpret_cast<const float4*>(sh + (size_t) group * kGroup * shCount) + lane;

    Rgb4 rgb{splat4(0.0f), splat4(0.0f), splat4(0.0f)};
    {
        const float4 b0[1] = {splat4(0.282094792f)};
        accumulateLevel<0, 1>(rows, b0, rgb);
    }
    if constexpr (L > 0)
    {
        const float4 b1[3] = {scale4(0.488602512f, y), scale4(0.488602512f, z), scale4(0.488602512f, x)};
        accumulateLevel<1, 3>(rows, b1, rgb);
    }
    if constexpr (L > 1)
    {
        const float4 xx = mul4(x, x), yy = mul4(y, y);
        const float4 s = add4(xx, yy), d = sub4(xx, yy);
        const float4 xy = mul4(x, y), yz = mul4(y, z), xz = mul4(x, z);
        {
            const float4 b2[5] = {scale4(1.09254843f, xy), scale4(1.09254843f, yz), affine4(-0.946174696f, s, 0.630783131f),
                                  scale4(1.09254843f, xz), scale4(0.546274215f, d)};
            accumulateLevel<4, 5>(rows, b2, rgb);
        }
        if constexpr (L > 2)
        {
            const float4 t3 = sub4(scale4(3.0f, xx), yy); // 3xx - yy
            const float4 u3 = sub4(xx, scale4(3.0f, yy)); // xx - 3yy
            {
                const float4 b3[7] = {scale4(0.59004359f, mul4(y, t3)),
                                      scale4(2.89061144f, mul4(xy, z)),
                                      mul4(scale4(3.6563664f, y), affine4(-0.625f, s, 0.5f)),
                                      mul4(scale4(0.373176333f, z), affine4(-5.0f, s, 2.0f)),
                                      mul4(scale4(0.457045799f, x), affine4(-5.0f, s, 4.0f)),
                                      scale4(1.44530572f, mul4(z, d)),
                                      scale4(0.59004359f, mul4(x, u3))};
                accumulateLevel<9, 7>(rows, b3, rgb);
            }
            if constexpr (L > 3)
            {
                const float4 b4[9] = {scale4(2.50334294f, mul4(xy, d)),
                                      scale4(1.77013077f, mul4(yz, t3)),
                                      mul4(scale4(1.26156626f, xy), affine4(-5.25f, s, 4.5f)),
                                      mul4(scale4(1.78412412f, yz), affine4(-2.625f, s, 1.5f)),
                                      fma4(s, affine4(3.70249414f, s, -4.23142188f), splat4(0.846284375f)),
                                      mul4(scale4(0.669046544f, xz), affine4(-7.0f, s, 4.0f)),
                                      mul4(scale4(-0.473087348f, d), affine4(7.0f, s, -6.0f)),
                                      scale4(1.77013077f, mul4(xz, u3)),
                                      scale4(0.625835735f, fma4(d, d, scale4(-4.0f, mul4(xx, yy))))};
                accumulateLevel<16, 9>(rows, b4, rgb);
            }
        }
    }

    // clamp(result + 0.5, 0, 1) (GaussianRender.cu:154); alpha = position.w (:168,180)
    const float rr[4] = {rgb.r.x, rgb.r.y, rgb.r.z, rgb.r.w};
    const float gg[4] = {rgb.g.x, rgb.g.y, rgb.g.z, rgb.g.w};
    const float bb[4] = {rgb.b.x, rgb.b.y, rgb.b.z, rgb.b.w};
#pragma unroll
    for (int i = 0; i != 4; ++i)
    {
        if (first + i < count)
        {
            color[first + i] = make_float4(__saturatef(rr[i] + 0.5f), __saturatef(gg[i] + 0.5f), __saturatef(bb[i] + 0.5f), p[i].w);
        }
    }
}

} // namespace

void launchSphericalHarmonics(hipStream_t stream, int count, int degree, int shCount, const float4* position, const float* sh,
                              float3 cameraPosition, float4* color)
{
    if (count <= 0) return;
    const int groups = (count + kGroup - 1) / kGroup;
    const dim3 grid((groups + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
    switch (degree)
    {
    case 0: hipLaunchKernelGGL(sphericalHarmonicsKernel<0>, grid, block, 0, stream, count, shCount, position, sh, cameraPosition, color); break;
    case 1: hipLaunchKernelGGL(sphericalHarmonicsKernel<1>, grid, block, 0, stream, count, shCount, position, sh, cameraPosition, color); break;
    case 2: hipLaunchKernelGGL(sphericalHarmonicsKernel<2>, grid, block, 0, stream, count, shCount, position, sh, cameraPosition, color); break;
    case 3: hipLaunchKernelGGL(sphericalHarmonicsKernel<3>, grid, block, 0, stream, count, shCount, position, sh, cameraPosition, color); break;
    default: hipLaunchKernelGGL(sphericalHarmonicsKernel<4>, grid, block, 0, stream, count, shCount, position, sh, cameraPosition, color); break;
    }
}

} // namespace gs
