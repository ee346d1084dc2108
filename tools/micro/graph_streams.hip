// graph_streams.hip -- do captured chains of small kernels on DIFFERENT streams overlap, or is the chip's command processing the
// shared limit?  S host threads, each replaying a 26-node chain (the node count of an 8-way band frame) on its own stream.
// If dispatch were per-stream, S streams would finish S chains in the time of one; if a shared front end serialises launches,
// the time per chain stays constant.  (diagnostics)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/graph_streams.hip -o tools/micro/graph_streams -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

// `work` selects what a node does: 0 = nothing, 1 = ~3 us of work on 64 workgroups, 2 = ~3 us on 2048 workgroups (fills the chip)
__global__ void node(float* p, int iters)
{
    float a = threadIdx.x;
    for (int i = 0; i < iters; ++i) a = a * 1.0000001f + 1e-9f;
    if (a == 12345.678f) p[0] = a;
}

int main()
{
    const int kNodes = 26, reps = 300;
    float* d;
    hipMalloc(&d, 4);
    for (int work = 0; work != 3; ++work)
        for (int S : {1, 2, 3, 4, 6, 8})
        {
            std::vector<hipStream_t> streams(S);
            std::vector<hipGraphExec_t> execs(S);
            for (int s = 0; s != S; ++s)
            {
                hipStreamCreateWithFlags(&streams[s], hipStreamNonBlocking);
                hipGraph_t g;
                hipStreamBeginCapture(streams[s], hipStreamCaptureModeThreadLocal);
                for (int i = 0; i < kNodes; ++i)
                    hipLaunchKernelGGL(node, dim3(work == 2 ? 2048 : 64), dim3(256), 0, streams[s], d, work == 0 ? 0 : 1500);
                hipStreamEndCapture(streams[s], &g);
                hipGraphInstantiate(&execs[s], g, nullptr, nullptr, 0);
                hipGraphLaunch(execs[s], streams[s]);
                hipStreamSynchronize(streams[s]);
            }
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> threads;
            for (int s = 0; s != S; ++s)
                threads.emplace_back([&, s] {
                    for (int i = 0; i < reps; ++i)
                    {
                        hipGraphLaunch(execs[s], streams[s]);
                        if ((i & 1) == 1) hipStreamSynchronize(streams[s]); // at most two chains queued per stream, like the renderer
                    }
                    hipStreamSynchronize(streams[s]);
                });
            for (auto& th : threads) th.join();
            auto t1 = std::chrono::steady_clock::now();
            const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
            printf("work %d, %d streams: %7.1f us per chain of %d nodes (%5.2f us per node), %6.0f chains/s\n", work, S, us / (reps * S), kNodes,
                   us / (reps * S) / kNodes, 1e6 * reps * S / us);
            for (int s = 0; s != S; ++s)
            {
                hipGraphExecDestroy(execs[s]);
                hipStreamDestroy(streams[s]);
            }
        }
    return 0;
}
