// gsplat_demo.cpp -- headless C++ host over GaussianRender.hpp: what the reference's Demo.cpp does per frame
// (src/Demo.cpp:314-331 buffers, :352-479 frame, :541-562 per-stage averages), without the window.
//
//   gsplat_demo scene.ply [width height frames out.ppm]     render a 3DGS .ply from its default camera pose
//   gsplat_demo --procedural N [width height frames out.ppm] N random splats (Demo.cpp:119-143, seeded)
//   ... --world N --rank R --id-file PATH                    one process per GPU (device R): rank R builds, sorts and rasterises
//                                                            its band of tile rows only, and the bands are gathered on rank 0
//                                                            with the library's RCCL exchange (gs_gather_bands) every frame; the
//                                                            128-byte RCCL id travels through PATH.  Rank 0 writes the image.
//
//   ... --events FILE                                        the interactive viewer without the window (Demo.cpp:371-392): one line per
//                                                            frame, "dt mouseButtons keys cursorX cursorY" (GS_MOUSE_* / GS_KEY_* bit masks,
//                                                            cursor in pixels from the top-left corner), fed to gs::CameraControls::update before
//                                                            the frame is rendered; frames beyond the file repeat its last line with dt = 0.
//                                                            The pose after the last frame is printed.
//   ... --builder sorted|reference-order                     gs_set_staged_builder (default sorted: buildTileList emits in the order the
//                                                            sort leaves, sortTileList looks before it sorts)
//   ... --one-call                                           the same frames through gs_renderer_render (one call
//                                                            instead of eight, captured frames, a device synchronise after each like
//                                                            Demo.cpp:478); prints the wall clock per frame
//   ... --one-call --events FILE [--tile-order]              the viewer's session through the one-call renderer; --tile-order: the raster's tiles started
//                                                            longest-first by the previous frame's consumption (gs_renderer_set_tile_start_order:
//                                                            an interactive camera moves a little per frame, Demo.cpp:371-392; same image bytes).
//                                                            Without --one-call the flag switches the STAGED loop's form on (gs_set_staged_tile_start_order)
//   ... --warm-up W                                          W more frames first that the printed averages leave out
//   ... --dump-every K                                       every K-th frame as <out.ppm>.<frame>.ppm (the session as an image sequence)
//
// Build: make -C cudagaussianrenderer_amd/csrc demo
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "GaussianRender.hpp"

int main(int argc, char** argv)
{
    if (argc < 2)
    {
        std::fprintf(stderr, "usage: %s scene.ply|--procedural N [width height frames out.ppm]\n", argv[0]);
        return 2;
    }
    // multi-GPU options (anywhere on the command line)
    int world = 1, rank = 0;
    const char* idFile = nullptr;
    const char* eventsFile = nullptr;
    int dumpEvery = 0;
    int warmUp = 0; // frames rendered before the statistics start (first frames allocate scratch and meet cold clocks)
    bool oneCall = false; // the frame loop body as ONE call (gs_renderer_render, captured frames) instead of the eight staged ones
    bool tileOrder = false; // --one-call: gs_renderer_set_tile_start_order(PREVIOUS_FRAME)
    std::vector<char*> positional;
    for (int i = 0; i < argc; ++i)
    {
        if (std::strcmp(argv[i], "--world") == 0 && i + 1 < argc) world = std::atoi(argv[++i]);
        else if (std::strcmp(argv[i], "--rank") == 0 && i + 1 < argc) rank = std::atoi(argv[++i]);
        else if (std::strcmp(argv[i], "--id-file") == 0 && i + 1 < argc) idFile = argv[++i];
        else if (std::strcmp(argv[i], "--events") == 0 && i + 1 < argc) eventsFile = argv[++i];
        else if (std::strcmp(argv[i], "--dump-every") == 0 && i + 1 < argc) dumpEvery = std::atoi(argv[++i]);
        else if (std::strcmp(argv[i], "--builder") == 0 && i + 1 < argc)
            gs_set_staged_builder(std::strcmp(argv[++i], "reference-order") == 0 ? GS_STAGED_BUILDER_REFERENCE_ORDER : GS_STAGED_BUILDER_SORTED);
        else if (std::strcmp(argv[i], "--warm-up") == 0 && i + 1 < argc) warmUp = std::atoi(argv[++i]);
        else if (std::strcmp(argv[i], "--one-call") == 0) oneCall = true;
        else if (std::strcmp(argv[i], "--tile-order") == 0)
        {
            tileOrder = true;
            gs_set_staged_tile_start_order(GS_TILE_START_ORDER_PREVIOUS_FRAME); // (the staged loop's switch; --one-call sets the renderer's)
        }
        else positional.push_back(argv[i]);
    }
    argc = (int) positional.size();
    argv = positional.data();
    if (world > 1) gs_set_device(rank);
    GsComm* comm = nullptr;
    {
        char id[GS_COMM_UNIQUE_ID_BYTES] = {0};
        if (world > 1)
        {
            if (idFile == nullptr) return 2;
            const std::string tmp = std::string(idFile) + ".tmp";
            if (rank == 0)
            {
                if (gs_comm_unique_id(id) != 0) return 1;
                std::FILE* f = std::fopen(tmp.c_str(), "wb");
                std::fwrite(id, 1, sizeof(id), f);
                std::fclose(f);
                std::rename(tmp.c_str(), idFile);
            }
            else
            {
                std::FILE* f = nullptr;
                for (int tries = 0; tries != 600 && (f = std::fopen(idFile, "rb")) == nullptr; ++tries)
                    std::this_thread::sleep_for(std::chrono::milliseconds(100));
                if (f == nullptr || std::fread(id, 1, sizeof(id), f) != sizeof(id)) return 1;
                std::fclose(f);
            }
        }
        comm = gs_comm_create(rank, world, id);
        if (comm == nullptr) return 1;
    }
    int arg = 1;
    std::vector<gs::float4_t> position, scaleAndRotation, color;
    std::vector<float> sphericalHarmonics;
    int shDegree = 0, shCount = 0;
    gs::vec3_t boundsMin{-4, -4, -4}, boundsMax{4, 4, 4};
    if (std::strcmp(argv[arg], "--procedural") == 0)
    {
        const int n = std::atoi(argv[arg + 1]);
        arg += 2;
        position.resize(n);
        scaleAndRotation.resize(n);
        color.resize(n);
        const float lo[3] = {-4, -4, -4}, hi[3] = {4, 4, 4};
        gs::generateRandomGaussians(position, scaleAndRotation, color, 0.01f, 0.5f, lo, hi);
    }
    else
    {
        gs::parsePly(argv[arg++], position, scaleAndRotation, color, sphericalHarmonics, shDegree, shCount, boundsMin, boundsMax);
        if (shDegree != 0)
        {
            std::vector<float> aligned;
            gs::realignSphericalHarmonics(sphericalHarmonics, aligned, GS_SH_GROUP_SIZE, shCount, (int) position.size());
            sphericalHarmonics.swap(aligned);
        }
    }
    const int width = argc > arg ? std::atoi(argv[arg]) : 1024;
    const int height = argc > arg + 1 ? std::atoi(argv[arg + 1]) : 1024;
    const int frames = argc > arg + 2 ? std::atoi(argv[arg + 2]) : 100;
    const char* outPath = argc > arg + 3 ? argv[arg + 3] : nullptr;
    const int32_t splatCount = (int32_t) position.size();

    if (oneCall && world == 1)
    {
        // the loop body of Demo.cpp:352-479 as one call.  (The scene goes in as parsePly returned it: the
        // coefficient rows are re-read in the file's [splat][count] layout, so they are parsed again rather than un-aligned.)
        std::vector<float> rows;
        if (shDegree != 0)
        {
            std::vector<gs::float4_t> p2, s2, c2;
            int d2 = 0, n2 = 0;
            gs::vec3_t lo2{}, hi2{};
            gs::parsePly(argv[1], p2, s2, c2, rows, d2, n2, lo2, hi2);
        }
        GsRenderer* renderer = gs_renderer_create(width, height);
        if (renderer == nullptr) return 1;
        if (gs_renderer_set_scene(renderer, splatCount, &position[0].x, &scaleAndRotation[0].x, &color[0].x, shDegree != 0 ? rows.data() : nullptr, shDegree,
                                  shCount) != 0)
            return 1;
        gs_renderer_set_graph(renderer, 1);
        // one frame at a time: the renderer chooses its depth layers for a lone frame (two from 16 M pairs per frame; same image bytes)
        gs_renderer_set_depth_layers(renderer, GS_DEPTH_LAYERS_CHOOSE, nullptr);
        if (tileOrder) gs_renderer_set_tile_start_order(renderer, GS_TILE_START_ORDER_PREVIOUS_FRAME);
        gs::CameraControls controls((float) width, (float) height);
        controls.setBounds(boundsMin, boundsMax);
        gs::CameraData cam = controls.getCameraData();
        // the viewer's recorded session, if any (one line per frame: dt mouseButtons keys cursorX cursorY; see --events above)
        struct Step { float dt; GsCameraInput input; };
        std::vector<Step> steps;
        if (eventsFile != nullptr)
        {
            std::FILE* f = std::fopen(eventsFile, "r");
            if (f == nullptr)
            {
                std::fprintf(stderr, "cannot open %s\n", eventsFile);
                return 2;
            }
            Step e{};
            unsigned buttons = 0, keys = 0;
            while (std::fscanf(f, "%f %u %u %lf %lf", &e.dt, &buttons, &keys, &e.input.cursorX, &e.input.cursorY) == 5)
            {
                e.input.mouseButtons = buttons;
                e.input.keys = keys;
                steps.push_back(e);
            }
            std::fclose(f);
        }
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        for (int frame = -warmUp; frame != frames; ++frame)
        {
            if (frame == 0) t0 = std::chrono::steady_clock::now();
            if (!steps.empty() && frame >= 0) // Demo.cpp:371-392
            {
                Step e = steps[std::min((size_t) frame, steps.size() - 1)];
                if ((size_t) frame >= steps.size()) e.dt = 0.0f;
                controls.setInput(e.input);
                controls.update(e.dt);
                cam = controls.getCameraData();
            }
            if (gs_renderer_render(renderer, &cam, 0, nullptr) != 0) return 1;
            gs_synchronize(); // Demo.cpp:478 (the renderer's frames run on the library's current stream here)
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (frames > 0 ? frames : 1);
        std::printf("splats %d, %dx%d, %d frames (one call per frame, captured, synchronised after each)\n", splatCount, width, height, frames);
        std::printf("Frame loop wall clock %.6f ms per frame (one-call renderer%s%s)\n", ms, steps.empty() ? "" : ", recorded session",
                    tileOrder ? ", tile start order from the previous frame" : "");
        if (!steps.empty())
        {
            const GsCameraControlsState st = controls.getState();
            std::printf("camera after %d frames: position %.9g %.9g %.9g rotation %.9g %.9g %.9g %.9g anchor %.9g %.9g %.9g mode %d\n", frames,
                        st.position[0], st.position[1], st.position[2], st.rotationXYZW[0], st.rotationXYZW[1], st.rotationXYZW[2], st.rotationXYZW[3],
                        st.anchorPosition[0], st.anchorPosition[1], st.anchorPosition[2], st.mode);
        }
        if (outPath != nullptr)
        {
            std::vector<unsigned char> image((size_t) width * height * 4);
            gs_renderer_read_back_buffer(renderer, image.data());
            std::FILE* fp = std::fopen(outPath, "wb");
            if (fp == nullptr) return 1;
            std::fprintf(fp, "P6\n%d %d\n255\n", width, height);
            for (int y = height - 1; y >= 0; --y)
                for (int x = 0; x != width; ++x) std::fwrite(&image[((size_t) y * width + x) * 4], 1, 3, fp);
            std::fclose(fp);
        }
        gs_renderer_destroy(renderer);
        return 0;
    }

    int tilesX = 0, tilesY = 0;
    gs_tile_counts(width, height, &tilesX, &tilesY);
    // contiguous bands of tile rows, sizes differing by at most one (1080p, 8 ranks: 9,9,9,9,8,8,8,8)
    std::vector<int32_t> bands(2 * (size_t) world);
    for (int r = 0, begin = 0; r != world; ++r)
    {
        const int rows = tilesY / world + (r < tilesY % world ? 1 : 0);
        bands[2 * r] = begin;
        bands[2 * r + 1] = begin += rows;
    }

    // Device buffers, owned by the caller as in the reference.
    gs::DeviceBuffer<gs::float4_t> positionBuffer(position), scaleAndRotationBuffer(scaleAndRotation), colorBuffer(color);
    gs::DeviceBuffer<float> shBuffer(sphericalHarmonics);
    gs::DeviceBuffer<gs::float4_t> conicBuffer(splatCount), ellipseBuffer(splatCount);
    gs::DeviceBuffer<gs::float2_t> clipXYBuffer(splatCount);
    gs::DeviceBuffer<float> clipZBuffer(splatCount);
    gs::DeviceBuffer<int32_t> tileRangeBuffer((size_t) tilesX * tilesY * 2);
    gs::DeviceBuffer<gs::uchar4_t> backBuffer((size_t) width * height);
    int32_t capacity = splatCount * 8;
    gs::DeviceBuffer<uint64_t> keysA(capacity), keysB(capacity);
    gs::DeviceBuffer<int32_t> valuesA(capacity), valuesB(capacity);
    void* sortTemp = nullptr;
    size_t sortTempBytes = 0;

    // Camera: CameraControls placed by the scene's bounds (Demo.cpp:336-337), moved by the recorded input when there is any.
    gs::CameraControls cameraControls((float) width, (float) height);
    cameraControls.setBounds(boundsMin, boundsMax);
    struct Event { float dt; GsCameraInput input; };
    std::vector<Event> events;
    if (eventsFile != nullptr)
    {
        std::FILE* f = std::fopen(eventsFile, "r");
        if (f == nullptr)
        {
            std::fprintf(stderr, "cannot open %s\n", eventsFile);
            return 2;
        }
        Event e{};
        unsigned buttons = 0, keys = 0;
        while (std::fscanf(f, "%f %u %u %lf %lf", &e.dt, &buttons, &keys, &e.input.cursorX, &e.input.cursorY) == 5)
        {
            e.input.mouseButtons = buttons;
            e.input.keys = keys;
            events.push_back(e);
        }
        std::fclose(f);
    }
    gs::CameraData camera = cameraControls.getCameraData();

    // PPM writer (Demo.cpp:146-147 convention: row 0 of the buffer is the bottom of the image)
    const auto writePpm = [&](const char* path) {
        std::vector<gs::uchar4_t> image;
        backBuffer.copyTo(image, (size_t) width * height);
        std::FILE* fp = std::fopen(path, "wb");
        if (fp == nullptr) return false;
        std::fprintf(fp, "P6\n%d %d\n255\n", width, height);
        std::vector<unsigned char> row((size_t) width * 3);
        for (int y = height - 1; y >= 0; --y)
        {
            for (int x = 0; x != width; ++x)
            {
                const gs::uchar4_t p = image[(size_t) y * width + x];
                row[3 * (size_t) x] = p.x, row[3 * (size_t) x + 1] = p.y, row[3 * (size_t) x + 2] = p.z;
            }
            std::fwrite(row.data(), 1, row.size(), fp);
        }
        std::fclose(fp);
        return true;
    };

    gs::GpuTimer timer;
    double msSh = 0, msClip = 0, msBuild = 0, msSort = 0, msRanges = 0, msRaster = 0;
    bool saturated = false;
    std::chrono::steady_clock::time_point loopStart = std::chrono::steady_clock::now();
    for (int frame = -warmUp; frame != frames; ++frame)
    {
        if (frame == 0)
        {
            msSh = msClip = msBuild = msSort = msRanges = msRaster = 0;
            loopStart = std::chrono::steady_clock::now();
        }
        if (saturated) // grow the list when the previous frame filled it
        {
            capacity <<= 1;
            keysA.resizeIfNeeded(capacity);
            keysB.resizeIfNeeded(capacity);
            valuesA.resizeIfNeeded(capacity);
            valuesB.resizeIfNeeded(capacity);
        }
        if (!events.empty() && frame >= 0) // Demo.cpp:371-392: update input & camera controls, build the camera data
        {
            Event e = events[std::min((size_t) frame, events.size() - 1)];
            if ((size_t) frame >= events.size()) e.dt = 0.0f;
            cameraControls.setInput(e.input);
            cameraControls.update(e.dt);
            camera = cameraControls.getCameraData();
        }
        backBuffer.clearMemory(0);
        tileRangeBuffer.clearMemory(255);

        gs::GlobalArgs args{};
        args.cameraData = camera;
        args.position = &positionBuffer.getPtr()->x;
        args.scaleAndRotation = &scaleAndRotationBuffer.getPtr()->x;
        args.color = &colorBuffer.getPtr()->x;
        args.sphericalHarmonicsDegree = shDegree;
        args.sphericalHarmonicsCount = shCount;
        args.sphericalHarmonics = shBuffer.getPtr();
        args.positionClipSpaceXY = &clipXYBuffer.getPtr()->x;
        args.positionClipSpaceZ = clipZBuffer.getPtr();
        args.conic = &conicBuffer.getPtr()->x;
        args.screenEllipse = &ellipseBuffer.getPtr()->x;
        args.splatCount = splatCount;
        args.backBuffer = &backBuffer.getPtr()->x;
        args.tileRange = tileRangeBuffer.getPtr();
        args.width = width;
        args.height = height;
        args.tileRowBegin = world > 1 ? bands[2 * rank] : 0;
        args.tileRowEnd = world > 1 ? bands[2 * rank + 1] : 0;
        gs::setGlobalArgs(&args);

        gs::TileListArgs list{keysA.getPtr(), valuesA.getPtr(), 0, capacity};
        gs::setTileListArgs(&list);

        if (shDegree != 0)
        {
            gs::evaluateSphericalHarmonics(timer, splatCount);
            msSh += timer.getElapseTimedMs();
        }
        gs::evaluateSplatClipData(timer, splatCount);
        msClip += timer.getElapseTimedMs();

        list.size = gs::buildTileList(timer, 256 * 8, capacity);
        msBuild += timer.getElapseTimedMs();
        saturated = list.size == capacity;
        gs::setTileListArgs(&list);

        gs::DoubleBuffer<uint64_t> keys(keysA.getPtr(), keysB.getPtr());
        gs::DoubleBuffer<int32_t> values(valuesA.getPtr(), valuesB.getPtr());
        gs::sortTileList(timer, list.size, sortTemp, sortTempBytes, keys, values);
        msSort += timer.getElapseTimedMs();
        list.keys = keys.current();
        list.values = values.current();
        gs::setTileListArgs(&list);

        if (list.size != 0)
        {
            gs::evaluateTileRange(timer, list.size);
            msRanges += timer.getElapseTimedMs();
        }
        gs::rasterizeTile(timer);
        msRaster += timer.getElapseTimedMs();
        // the one exchange of the frame: every band's pixel rows -> rank 0's back buffer (no-op in a world of one)
        if (gs_gather_bands(comm, &backBuffer.getPtr()->x, width, height, bands.data(), 0, gs_get_stream()) != 0) return 1;
        gs_synchronize();
        if (dumpEvery > 0 && frame >= 0 && rank == 0 && outPath != nullptr && (frame + 1) % dumpEvery == 0)
        {
            // the viewer's "screen", headless: every dumpEvery-th frame next to the final image as <out>.<frame>.ppm
            char path[1024];
            std::snprintf(path, sizeof(path), "%s.%05d.ppm", outPath, frame + 1);
            if (!writePpm(path)) return 1;
        }
    }
    const double wallMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - loopStart).count();
    gs_device_free(sortTemp);

    const double f = frames > 0 ? frames : 1;
    std::printf("splats %d, %dx%d, %d frames (staged calls, blocking per-stage timers)\n", splatCount, width, height, frames);
    std::printf("Evaluate Spherical Harmonics %.6f ms\nEvaluate Clip Data %.6f ms\nBuild Tile List %.6f ms\nSort Tile List %.6f ms\n"
                "Evaluate Tile Ranges %.6f ms\nRasterize %.6f ms\nTotal %.6f ms\n",
                msSh / f, msClip / f, msBuild / f, msSort / f, msRanges / f, msRaster / f,
                (msSh + msClip + msBuild + msSort + msRanges + msRaster) / f);
    std::printf("Frame loop wall clock %.6f ms per frame (%s builder%s)\n", wallMs / f,
                gs_get_staged_builder() == GS_STAGED_BUILDER_SORTED ? "sorted" : "reference-order",
                gs_get_staged_tile_start_order() == GS_TILE_START_ORDER_PREVIOUS_FRAME ? ", tile start order from the previous frame" : "");

    if (!events.empty())
    {
        const GsCameraControlsState st = cameraControls.getState();
        std::printf("camera after %d frames: position %.9g %.9g %.9g rotation %.9g %.9g %.9g %.9g anchor %.9g %.9g %.9g mode %d\n", frames,
                    st.position[0], st.position[1], st.position[2], st.rotationXYZW[0], st.rotationXYZW[1], st.rotationXYZW[2], st.rotationXYZW[3],
                    st.anchorPosition[0], st.anchorPosition[1], st.anchorPosition[2], st.mode);
    }
    gs_comm_destroy(comm);
    if (outPath != nullptr && rank == 0 && !writePpm(outPath)) return 1;
    return 0;
}
