// graph_memset.hip -- does a hipMemsetAsync node inside a captured graph keep writing its fill value when the graph also holds a
// memcpy node and many kernel nodes with large by-value arguments?  (r01 saw the range table of captured band frames come back
// with bytes that were not 0xFF and replaced the memset by a fill; ADVICE r01 asked for a standalone reproducer.)
// The graph mirrors the captured band frame: H2D memcpy of a parameter block, memset of a 65 280-byte table to 0xFF, K kernels
// taking a 96-byte struct by value (one of them scribbles into ANOTHER buffer), then a checker that counts table words != -1
// BEFORE a writer kernel dirties the table for the next replay.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/graph_memset.hip -o tools/micro/graph_memset && tools/micro/graph_memset
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

struct Big
{
    float m[16];
    int a, b, c, d;
    float e, f, g, h;
};

__global__ void consume(Big big, const Big* params, int* other, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) other[i] = big.a + params->b + (int) big.m[i & 15];
}
__global__ void check(const int* table, int n, unsigned* bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && table[i] != -1) atomicAdd(bad, 1u);
}
__global__ void dirty(int* table, int n, int value)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) table[i] = value + i;
}

int main()
{
    const int words = 2 * 8160, kernels = 24, replays = 400;
    int *table, *other;
    unsigned* bad;
    Big *paramsDevice, *paramsHost;
    hipMalloc(&table, words * 4);
    hipMalloc(&other, 1 << 20);
    hipMalloc(&bad, 4);
    hipMalloc(&paramsDevice, sizeof(Big));
    hipHostMalloc(&paramsHost, sizeof(Big), hipHostMallocDefault);
    hipMemset(bad, 0, 4);
    hipMemset(table, 0x5A, words * 4);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int variant = 0; variant != 2; ++variant) // 0: memset node first, 1: memset node after half of the kernels
    {
        hipGraph_t g;
        hipGraphExec_t e;
        Big big{};
        big.a = 3;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        hipMemcpyAsync(paramsDevice, paramsHost, sizeof(Big), hipMemcpyHostToDevice, s);
        if (variant == 0) hipMemsetAsync(table, 0xFF, words * 4, s);
        for (int k = 0; k != kernels; ++k)
        {
            big.m[k & 15] = (float) k;
            hipLaunchKernelGGL(consume, dim3(64), dim3(256), 0, s, big, paramsDevice, other, 1 << 14);
            if (variant == 1 && k == kernels / 2) hipMemsetAsync(table, 0xFF, words * 4, s);
        }
        hipLaunchKernelGGL(check, dim3((words + 255) / 256), dim3(256), 0, s, table, words, bad);
        hipLaunchKernelGGL(dirty, dim3((words + 255) / 256), dim3(256), 0, s, table, words, 12345);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
        unsigned total = 0;
        for (int r = 0; r != replays; ++r)
        {
            hipStreamSynchronize(s);
            std::memset(paramsHost, r & 0xFF, sizeof(Big)); // new parameter bytes before every replay, like the camera block
            paramsHost->b = r;
            hipGraphLaunch(e, s);
        }
        hipStreamSynchronize(s);
        hipMemcpy(&total, bad, 4, hipMemcpyDeviceToHost);
        printf("variant %d (%s): %u table words were not 0xFFFFFFFF over %d replays of a %d-node graph\n", variant,
               variant == 0 ? "memset node right after the memcpy node" : "memset node between kernel nodes", total, replays, kernels + 4);
        hipMemset(bad, 0, 4);
        hipGraphExecDestroy(e);
        hipGraphDestroy(g);
    }
    return 0;
}
