// frame_pump.cpp -- the pipelined frame loop of bench.py driven by NATIVE host threads (one per frame in flight) straight over
// the C ABI, to separate what the GPU can do from what the Python harness adds (GIL, futures, ctypes marshalling).
//   frame_pump scene.ply cams.bin W H rowBegin rowEnd inflight frames [graph=1] [mask=0] [views=1] [storage=0]
// storage: 0 = the splats stored in the file's order, 1 = along a Morton curve (gs_renderer_set_storage_order).
// mask: 0 = plain streams; 1 = every stream restricted to its own share of the compute units, bits dealt round-robin
// (bit i -> stream i % inflight: with the driver's XCC-interleaved mask layout a share of whole XCDs when inflight divides 8);
// 2 = contiguous shares of the mask (bit i -> stream i / (256 / inflight)).
// cams.bin: consecutive GsCameraData structs (224 bytes each), written by tools/frame_pump.py.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "gsplat_hip.h"
#include "gsplat_hip_debug.h" // gs_debug_host_times

int main(int argc, char** argv)
{
    if (argc < 9)
    {
        std::fprintf(stderr, "usage: %s scene.ply cams.bin W H rowBegin rowEnd inflight frames [graph]\n", argv[0]);
        return 2;
    }
    const char* plyPath = argv[1];
    const int W = std::atoi(argv[3]), H = std::atoi(argv[4]), rowBegin = std::atoi(argv[5]), rowEnd = std::atoi(argv[6]);
    const int D = std::atoi(argv[7]), frames = std::atoi(argv[8]);
    const int graph = argc > 9 ? std::atoi(argv[9]) : 1;
    const int maskMode = argc > 10 ? std::atoi(argv[10]) : 0;
    const int views = argc > 11 ? std::atoi(argv[11]) : 1; // cameras per chain of kernels (gs_renderer_set_views)
    const int storage = argc > 12 ? std::atoi(argv[12]) : 0;
    std::vector<GsCameraData> cams;
    {
        std::FILE* f = std::fopen(argv[2], "rb");
        if (!f) return 1;
        GsCameraData c;
        while (std::fread(&c, sizeof(c), 1, f) == 1) cams.push_back(c);
        std::fclose(f);
    }
    GsPly ply;
    char err[256];
    if (gs_parse_ply(plyPath, &ply, err, sizeof(err)) < 0)
    {
        std::fprintf(stderr, "%s\n", err);
        return 1;
    }
    gs_set_device(0);
    std::vector<GsRenderer*> rs(D);
    std::vector<hipStream_t> streams(D);
    for (int d = 0; d != D; ++d)
    {
        rs[d] = gs_renderer_create(W, H);
        if (views > 1 && gs_renderer_set_views(rs[d], views) != 0) return 4;
        if (rowEnd > rowBegin) gs_renderer_set_tile_rows(rs[d], rowBegin, rowEnd);
        gs_renderer_set_storage_order(rs[d], storage ? GS_STORAGE_ORDER_MORTON : GS_STORAGE_ORDER_CALLER);
        gs_renderer_set_scene(rs[d], ply.vertexCount, ply.position, ply.scaleAndRotation, ply.color, ply.sphericalHarmonics,
                              ply.sphericalHarmonicsDegree, ply.sphericalHarmonicsCount);
        if (maskMode == 0) (void) hipStreamCreateWithFlags(&streams[d], hipStreamNonBlocking);
        else
        {
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // 256 compute units
            for (int cu = 0; cu != 256; ++cu)
            {
                const int owner = maskMode == 1 ? cu % D : cu / ((256 + D - 1) / D);
                if (owner == d) mask[cu >> 5] |= 1u << (cu & 31);
            }
            if (hipExtStreamCreateWithCUMask(&streams[d], 8, mask) != hipSuccess) return 3;
        }
        gs_renderer_set_stream(rs[d], streams[d]);
        gs_renderer_set_graph(rs[d], graph);
    }
    auto run = [&](int count) {
        std::vector<std::thread> threads;
        for (int d = 0; d != D; ++d)
            threads.emplace_back([&, d] {
                gs_set_device(0);
                std::vector<GsCameraData> group(views);
                for (int c = d; c < count / views; c += D) // chain c renders frames c*views .. c*views + views-1
                {
                    for (int v = 0; v != views; ++v) group[v] = cams[(c * views + v) % cams.size()];
                    if (gs_renderer_render_views(rs[d], group.data()) != 0) std::abort();
                }
                (void) hipStreamSynchronize(streams[d]);
            });
        for (auto& t : threads) t.join();
    };
    run(8 * D);
    (void) gs_debug_host_times(nullptr, nullptr, 1);
    const auto t0 = std::chrono::steady_clock::now();
    run(frames);
    const auto t1 = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count() / frames;
    double launchUs = 0.0, awaitUs = 0.0;
    (void) gs_debug_host_times(&launchUs, &awaitUs, 0);
    std::printf("{\"inflight\": %d, \"views\": %d, \"rows\": [%d, %d], \"graph\": %d, \"cu_mask\": %d, \"morton\": %d, \"ms_per_frame\": %.4f, \"frames_per_s\": %.1f, "
                "\"host_us_in_graph_launch\": %.1f, \"host_us_awaiting_total\": %.1f}\n",
                D, views, rowBegin, rowEnd, graph, maskMode, storage, ms, 1e3 / ms, launchUs, awaitUs);
    for (int d = 0; d != D; ++d) gs_renderer_destroy(rs[d]);
    gs_free_ply(&ply);
    return 0;
}
