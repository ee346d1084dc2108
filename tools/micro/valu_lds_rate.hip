// valu_lds_rate.hip -- issue-rate calibration for the raster kernel's inner loop on gfx950 (diagnostics, not product code):
// cycles per wave64 instruction for v_fma_f32, v_pk_fma_f32, v_exp_f32, broadcast ds_read_b128 (all lanes one address),
// ds_read_b128 with one address per lane, and s_load_dwordx8 -- each with 1, 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_lds_rate.hip -o tools/micro/valu_lds_rate && tools/micro/valu_lds_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                                          \
    do                                                                                    \
    {                                                                                     \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess)                                                             \
        {                                                                                 \
            printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__);              \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

constexpr int kIters = 4096;
typedef float float2v __attribute__((ext_vector_type(2)));

// every kernel: out[0..1] = shader cycles (s_memtime) of wave 0 of block 0, out[2] keeps the results alive
__global__ __launch_bounds__(256) void fmaKernel(float* sink, unsigned long long* clk)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 1e-9f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i != kIters; ++i)
    {
        a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
        a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ __launch_bounds__(256) void pkFmaKernel(float* sink, unsigned long long* clk)
{
    float2v a0 = float2v(threadIdx.x), a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float2v m = float2v(1.0000001f), c = float2v(1e-9f);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i != kIters; ++i)
    {
        a0 = __builtin_elementwise_fma(a0, m, c); a1 = __builtin_elementwise_fma(a1, m, c); a2 = __builtin_elementwise_fma(a2, m, c);
        a3 = __builtin_elementwise_fma(a3, m, c); a4 = __builtin_elementwise_fma(a4, m, c); a5 = __builtin_elementwise_fma(a5, m, c);
        a6 = __builtin_elementwise_fma(a6, m, c); a7 = __builtin_elementwise_fma(a7, m, c);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; }
    const float2v s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

__global__ __launch_bounds__(256) void expKernel(float* sink, unsigned long long* clk)
{
    float a0 = -1e-3f * threadIdx.x, a1 = a0 - 1, a2 = a0 - 2, a3 = a0 - 3, a4 = a0 - 4, a5 = a0 - 5, a6 = a0 - 6, a7 = a0 - 7;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i != kIters; ++i)
    {
        a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
        a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// exp interleaved with independent pk_fma work: does the transcendental unit run beside the FMA pipe?
__global__ __launch_bounds__(256) void expPlusFmaKernel(float* sink, unsigned long long* clk)
{
    float e0 = -1e-3f * threadIdx.x, e1 = e0 - 1;
    float2v a0 = float2v(threadIdx.x), a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f;
    const float2v m = float2v(1.0000001f), c = float2v(1e-9f);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i != kIters; ++i)
    {
        e0 = __builtin_amdgcn_exp2f(e0); e1 = __builtin_amdgcn_exp2f(e1);
        a0 = __builtin_elementwise_fma(a0, m, c); a1 = __builtin_elementwise_fma(a1, m, c); a2 = __builtin_elementwise_fma(a2, m, c);
        a3 = __builtin_elementwise_fma(a3, m, c); a4 = __builtin_elementwise_fma(a4, m, c); a5 = __builtin_elementwise_fma(a5, m, c);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; }
    const float2v s = a0 + a1 + a2 + a3 + a4 + a5;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + e0 + e1;
}

template <bool BROADCAST>
__global__ __launch_bounds__(256) void ldsKernel(float* sink, unsigned long long* clk)
{
    __shared__ float4 table[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) table[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    const int lane = BROADCAST ? 0 : (threadIdx.x & 63);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i != kIters; ++i)
    {
        const int base = (i * 8) & 511;
#pragma unroll
        for (int k = 0; k != 8; ++k)
        {
            const float4 v = table[base + k * (BROADCAST ? 1 : 64) % 512 + lane];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

// uniform-address loads of 32 bytes from a table in global memory: the compiler turns them into s_load_dwordx8
__global__ __launch_bounds__(256) void scalarLoadKernel(float* sink, unsigned long long* clk, const float4* __restrict__ table, int mask)
{
    float4 acc = make_float4(0, 0, 0, 0);
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + (threadIdx.x >> 6)) * 977);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i != kIters; ++i)
    {
#pragma unroll
        for (int k = 0; k != 8; ++k)
        {
            const int idx = ((wave + i * 8 + k) * 2) & mask;
            const float4 v0 = table[idx], v1 = table[idx + 1];
            acc.x += v0.x + v1.x; acc.y += v0.y + v1.y; acc.z += v0.z + v1.z; acc.w += v0.w + v1.w;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main()
{
    float* sink = nullptr;
    unsigned long long* clk = nullptr;
    float4* table = nullptr;
    const int tableEntries = 1 << 16; // 1 MB: L2-resident
    CHECK(hipMalloc((void**) &sink, sizeof(float) * 256 * 4096));
    CHECK(hipMalloc((void**) &clk, 64));
    CHECK(hipMalloc((void**) &table, sizeof(float4) * tableEntries));
    CHECK(hipMemset(table, 0, sizeof(float4) * tableEntries));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs; %d iterations\n", prop.gcnArchName, cus, kIters);
    printf("(waves of one SIMD are served oldest first, so wave 0's own cycle count barely changes with the occupancy: the issue cost of an\n"
           " instruction is read from the TIME added per extra resident wave)\n");
    printf("%-28s %6s %12s %16s\n", "kernel", "w/SIMD", "time us", "wave-0 cycles");
    struct Test
    {
        const char* name;
        int instrPerIter;
        int which;
    };
    const Test tests[] = {{"v_fma_f32", 8, 0}, {"v_pk_fma_f32", 8, 1}, {"v_exp_f32", 8, 2}, {"2 exp + 6 pk_fma", 8, 3},
                          {"ds_read_b128 broadcast", 8, 4}, {"ds_read_b128 per lane", 8, 5}, {"s_load 32 B (L2 table)", 8, 6}};
    for (const Test& t : tests)
    {
        double timeAt[5] = {0, 0, 0, 0, 0};
        for (int wavesPerSimd : {1, 2, 4})
        {
            const int blocks = cus * wavesPerSimd; // 256 threads = 4 waves = one per SIMD
            float ms = 0;
            for (int rep = 0; rep != 2; ++rep)
            {
                CHECK(hipEventRecord(e0));
                switch (t.which)
                {
                case 0: hipLaunchKernelGGL(fmaKernel, dim3(blocks), dim3(256), 0, 0, sink, clk); break;
                case 1: hipLaunchKernelGGL(pkFmaKernel, dim3(blocks), dim3(256), 0, 0, sink, clk); break;
                case 2: hipLaunchKernelGGL(expKernel, dim3(blocks), dim3(256), 0, 0, sink, clk); break;
                case 3: hipLaunchKernelGGL(expPlusFmaKernel, dim3(blocks), dim3(256), 0, 0, sink, clk); break;
                case 4: hipLaunchKernelGGL(ldsKernel<true>, dim3(blocks), dim3(256), 0, 0, sink, clk); break;
                case 5: hipLaunchKernelGGL(ldsKernel<false>, dim3(blocks), dim3(256), 0, 0, sink, clk); break;
                default: hipLaunchKernelGGL(scalarLoadKernel, dim3(blocks), dim3(256), 0, 0, sink, clk, table, tableEntries - 2); break;
                }
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            unsigned long long cycles = 0;
            CHECK(hipMemcpy(&cycles, clk, 8, hipMemcpyDeviceToHost));
            timeAt[wavesPerSimd] = ms * 1e3;
            printf("%-28s %6d %12.1f %16llu\n", t.name, wavesPerSimd, ms * 1e3, cycles);
        }
        printf("%-28s -> %.2f ns of SIMD time per wave64 instruction (loop body of %d)\n", t.name,
               1e3 * (timeAt[4] - timeAt[1]) / 3.0 / ((double) kIters * t.instrPerIter), t.instrPerIter);
    }
    return 0;
}
