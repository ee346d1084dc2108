// Micro-benchmark: how fast does the GPU start workgroups?  Empty kernels (and kernels that only touch LDS / wait on one
// dependent global load) with many 256-thread workgroups; prints microseconds per launch and waves per microsecond.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void emptyKernel(int* out) { if (out && threadIdx.x == 1024) out[0] = 1; }
template <int LDS>
__global__ void ldsKernel(int* out)
{
    __shared__ int s[LDS / 4];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (s[(threadIdx.x + 1) & 255] == -1) out[0] = 1;
}
__global__ void loadKernel(const int* in, int* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = in[i % n];          // first load
    const int k = in[(j & 0x7FFFFFFF) % n]; // dependent second load
    if (k == -12345) out[0] = 1;
}

template <typename F>
float timeIt(F f, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1000.0f / reps;
}

int main()
{
    int* out;
    hipMalloc(&out, 4);
    const int n = 1 << 24;
    int* in;
    hipMalloc(&in, sizeof(int) * n);
    std::vector<int> h(n);
    unsigned x = 12345;
    for (int i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (int) (x >> 1); }
    hipMemcpy(in, h.data(), sizeof(int) * n, hipMemcpyHostToDevice);
    for (int groups : {256, 1024, 4096, 16384, 65536})
    {
        const float e = timeIt([&] { hipLaunchKernelGGL(emptyKernel, dim3(groups), dim3(256), 0, 0, out); }, 50);
        const float l1 = timeIt([&] { hipLaunchKernelGGL(ldsKernel<1024>, dim3(groups), dim3(256), 0, 0, out); }, 50);
        const float l16 = timeIt([&] { hipLaunchKernelGGL(ldsKernel<16384>, dim3(groups), dim3(256), 0, 0, out); }, 50);
        const float ld = timeIt([&] { hipLaunchKernelGGL(loadKernel, dim3(groups), dim3(256), 0, 0, in, out, n); }, 50);
        printf("%6d groups x 256: empty %7.1f us (%6.0f waves/us)  lds1K %7.1f  lds16K %7.1f  2 dependent loads %7.1f us\n", groups, e,
               groups * 4 / e, l1, l16, ld);
    }
    return 0;
}
