// Micro-benchmark: host cost of replaying a captured linear chain of K small kernels (hipGraphLaunch) versus launching the K
// kernels directly, per launch and per node.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void tiny(int* p, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = v; }

int main()
{
    int* d;
    hipMalloc(&d, 4);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int k : {1, 4, 8, 16, 32, 64})
    {
        hipGraph_t g;
        hipGraphExec_t e;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < k; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s, d, i);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
        for (int i = 0; i < 10; ++i) hipGraphLaunch(e, s);
        hipStreamSynchronize(s);
        const int reps = 200;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) hipGraphLaunch(e, s);
        auto t1 = std::chrono::steady_clock::now(); // host time to enqueue
        hipStreamSynchronize(s);
        auto t2 = std::chrono::steady_clock::now();
        const double host = std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
        const double total = std::chrono::duration<double, std::micro>(t2 - t0).count() / reps;
        // direct launches
        for (int i = 0; i < k; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s, d, i);
        hipStreamSynchronize(s);
        auto u0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r)
            for (int i = 0; i < k; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s, d, i);
        auto u1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(s);
        auto u2 = std::chrono::steady_clock::now();
        const double dhost = std::chrono::duration<double, std::micro>(u1 - u0).count() / reps;
        const double dtotal = std::chrono::duration<double, std::micro>(u2 - u0).count() / reps;
        printf("%2d nodes: graph host %6.1f us (%.2f/node) done %6.1f us (%.2f/node) | direct host %6.1f us (%.2f/node) done %6.1f us (%.2f/node)\n", k,
               host, host / k, total, total / k, dhost, dhost / k, dtotal, dtotal / k);
        hipGraphExecDestroy(e);
        hipGraphDestroy(g);
    }
    return 0;
}
