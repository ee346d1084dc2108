// GaussianRender.hpp -- header-only C++ facade over the C ABI (gsplat_hip.h) that re-creates the names,
// argument order and ownership rules of the reference's host interface, so that its frame loop
// (src/Demo.cpp:394-479) compiles against this library with only the cuda* types renamed:
//
//   reference (src/GaussianRender.cuh, src/DeviceBuffer.cu, src/Utilities.h)      here
//   ---------------------------------------------------------------------------   --------------------------
//   CameraData / GlobalArgs / TileListArgs              .cuh:17-26,56-74,77-83     GsCameraData / GsGlobalArgs / GsTileListArgs (aliases below)
//   DoubleBuffer<T>                                     .cuh:87-116                gs::DoubleBuffer<T>
//   DeviceBuffer<T>                                     DeviceBuffer.cu:8-127      gs::DeviceBuffer<T>
//   CudaTimer                                           Utilities.h:155-187        gs::GpuTimer
//   setGlobalArgs ... rasterizeTile (8 functions)       .cuh:118-143               same names in namespace gs
//   parsePly / encodeVec4                               PlyParser.h:11-21          gs::parsePly / gs::encodeVec4
//   realignSphericalHarmonics / generateRandomGaussians Demo.cpp:119-143,160-194   same names in namespace gs
//   CameraControls (interactive camera)                 CameraControls.h:7-76      gs::CameraControls (input handed in, not polled from GLFW)
//   CudaGraphicsResource (GL buffer -> device pointer)  Utilities.h:113-153        gs::GraphicsResource
//
// No glm dependency: vectors are plain structs with the same memory layout as float4 / glm::vec3.
#pragma once

#include <cassert>
#include <cstdint>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "gsplat_hip.h"

namespace gs
{

struct float4_t { float x, y, z, w; };
struct float2_t { float x, y; };
struct vec3_t { float x, y, z; };
struct uchar4_t { unsigned char x, y, z, w; };

using CameraData = GsCameraData;
using GlobalArgs = GsGlobalArgs;
using TileListArgs = GsTileListArgs;

// Utilities.h:155-187
struct GpuTimer
{
    GsTimer* handle;
    GpuTimer() : handle(gs_timer_create()) {}
    ~GpuTimer() { gs_timer_destroy(handle); }
    GpuTimer(const GpuTimer&) = delete;
    GpuTimer& operator=(const GpuTimer&) = delete;
    void start() { gs_timer_start(handle); }
    void stop() { gs_timer_stop(handle); }
    float getElapseTimedMs() const { return gs_timer_elapsed_ms(handle); }
};

// DeviceBuffer.cu:8-127
template <typename T>
class DeviceBuffer
{
    T* m_Ptr{nullptr};
    std::size_t m_Size{0};
    std::size_t m_AllocatedSize{0};
    void releaseIfNeeded()
    {
        if (m_Ptr != nullptr) gs_device_free(m_Ptr);
        m_Ptr = nullptr;
        m_Size = m_AllocatedSize = 0;
    }

  public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(std::size_t size, bool shouldClearMemory = false)
    {
        resizeIfNeeded(size);
        if (shouldClearMemory) clearMemory(0);
    }
    explicit DeviceBuffer(const std::vector<T>& data) { copyFrom(data); }
    ~DeviceBuffer() { releaseIfNeeded(); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    std::size_t size() const { return m_Size; }
    T* getPtr() const { return m_Ptr; }
    bool resizeIfNeeded(std::size_t size) // grow-only, DeviceBuffer.cu:47-64
    {
        if (size != m_Size)
        {
            if (size > m_AllocatedSize)
            {
                releaseIfNeeded();
                m_Size = m_AllocatedSize = size;
                m_Ptr = static_cast<T*>(gs_device_alloc(size * sizeof(T)));
                return true;
            }
            m_Size = size;
        }
        return false;
    }
    void copyFrom(const std::vector<T>& data)
    {
        resizeIfNeeded(data.size());
        if (!data.empty()) gs_device_upload(m_Ptr, data.data(), data.size() * sizeof(T));
    }
    void copyTo(std::vector<T>& data, std::size_t count)
    {
        assert(count <= m_Size);
        if (data.size() != count) data.resize(count);
        if (count) gs_device_download(data.data(), m_Ptr, count * sizeof(T));
    }
    void copyTo(std::vector<T>& data) { copyTo(data, data.size()); }
    void clearMemory(int32_t value) { gs_device_memset(m_Ptr, value, m_Size * sizeof(T)); }
};

// GaussianRender.cuh:87-116
template <typename T>
struct DoubleBuffer
{
  private:
    T* m_Buffers[2];
    int m_Selector;

  public:
    DoubleBuffer(T* current, T* alternate) : m_Buffers{current, alternate}, m_Selector(0) {}
    DoubleBuffer(T* b0, T* b1, int selector) : m_Buffers{b0, b1}, m_Selector(selector) {}
    T* current() const { return m_Buffers[m_Selector]; }
    T* alternate() const { return m_Buffers[m_Selector ^ 1]; }
    int selector() const { return m_Selector; }
};

// GaussianRender.cuh:118-143
// GaussianRender.cu:8-9: `#define EPANECHNIKOV_KERNEL`, a build option of the reference, is a run-time switch here.
inline void setEpanechnikovKernel(bool enabled) { gs_set_footprint(enabled ? GS_FOOTPRINT_EPANECHNIKOV : GS_FOOTPRINT_GAUSSIAN); }
// Additions of the library (no counterpart in the reference): how buildTileList emits (sorted: sortTileList then finds nothing to do), and
// the per-thread scratch of the staged calls handed back.
inline void setSortedTileListBuilder(bool enabled) { gs_set_staged_builder(enabled ? GS_STAGED_BUILDER_SORTED : GS_STAGED_BUILDER_REFERENCE_ORDER); }
inline void releaseThreadScratch() { gs_release_thread_scratch(); }
inline void setGlobalArgs(GlobalArgs* globalArgs) { gs_set_global_args(globalArgs); }
inline void setTileListArgs(TileListArgs* tileListArgs) { gs_set_tile_list_args(tileListArgs); }
inline void evaluateSphericalHarmonics(GpuTimer& timer, int32_t count) { gs_evaluate_spherical_harmonics(timer.handle, count); }
inline void evaluateSplatClipData(GpuTimer& timer, int32_t count) { gs_evaluate_splat_clip_data(timer.handle, count); }
inline int32_t buildTileList(GpuTimer& timer, int32_t numBlocks, int32_t tileListCapacity)
{
    return gs_build_tile_list(timer.handle, numBlocks, tileListCapacity);
}
inline void sortTileList(GpuTimer& timer, int32_t tileListSize, void*& deviceTempStorage, size_t& tempStorageSizeInBytes,
                         DoubleBuffer<uint64_t>& keys, DoubleBuffer<int32_t>& values)
{
    GsDoubleBufferU64 k{{keys.current(), keys.alternate()}, 0};
    GsDoubleBufferI32 v{{values.current(), values.alternate()}, 0};
    gs_sort_tile_list(timer.handle, tileListSize, &deviceTempStorage, &tempStorageSizeInBytes, &k, &v);
    keys = DoubleBuffer<uint64_t>(k.buffers[k.selector], k.buffers[k.selector ^ 1]);
    values = DoubleBuffer<int32_t>(v.buffers[v.selector], v.buffers[v.selector ^ 1]);
}
inline void evaluateTileRange(GpuTimer& timer, int32_t tileListSize) { gs_evaluate_tile_range(timer.handle, tileListSize); }
inline void rasterizeTile(GpuTimer& timer) { gs_rasterize_tile(timer.handle); }

// PlyParser.h:11-21
inline uint32_t encodeVec4(const float v[4]) { return gs_encode_vec4(v); }
inline int parsePly(const char* filePath, std::vector<float4_t>& position, std::vector<float4_t>& scaleAndRotation,
                    std::vector<float4_t>& color, std::vector<float>& sphericalHarmonics, int& sphericalHarmonicsDegree,
                    int& sphericalHarmonicsCount, vec3_t& boundsMin, vec3_t& boundsMax)
{
    GsPly ply;
    char err[512] = {0};
    const int n = gs_parse_ply(filePath, &ply, err, sizeof(err));
    if (n < 0) throw std::invalid_argument(err); // PlyParser.cpp throws std::invalid_argument with the same text
    auto copy4 = [n](std::vector<float4_t>& dst, const float* src) {
        dst.resize((size_t) n);
        if (n) std::memcpy(dst.data(), src, sizeof(float4_t) * (size_t) n);
    };
    copy4(position, ply.position);
    copy4(scaleAndRotation, ply.scaleAndRotation);
    copy4(color, ply.color);
    sphericalHarmonicsDegree = ply.sphericalHarmonicsDegree;
    sphericalHarmonicsCount = ply.sphericalHarmonicsCount;
    if (ply.sphericalHarmonicsDegree != 0)
        sphericalHarmonics.assign(ply.sphericalHarmonics, ply.sphericalHarmonics + (size_t) n * ply.sphericalHarmonicsCount);
    boundsMin = vec3_t{ply.boundsMin[0], ply.boundsMin[1], ply.boundsMin[2]};
    boundsMax = vec3_t{ply.boundsMax[0], ply.boundsMax[1], ply.boundsMax[2]};
    gs_free_ply(&ply);
    return n;
}

// Demo.cpp:160-194
inline void realignSphericalHarmonics(const std::vector<float>& srcSh, std::vector<float>& dstSh, const int groupSize,
                                      const int shCount, const int splatCount)
{
    dstSh.resize(gs_realigned_sh_size(groupSize, shCount, splatCount));
    gs_realign_spherical_harmonics(srcSh.data(), dstSh.data(), groupSize, shCount, splatCount);
}

// Demo.cpp:119-143 (seeded instead of std::rand)
inline void generateRandomGaussians(std::vector<float4_t>& position, std::vector<float4_t>& scaleAndRotation,
                                    std::vector<float4_t>& color, const float minScale, const float maxScale,
                                    const float minPosition[3], const float maxPosition[3], uint64_t seed = 1234)
{
    gs_generate_random_gaussians((int32_t) position.size(), seed, minScale, maxScale, minPosition, maxPosition, &position[0].x,
                                 &scaleAndRotation[0].x, &color[0].x);
}

// CameraControls.h:7-76 without the window.  The reference's update() polls GLFW for the mouse button, the cursor and the W A S D Q E
// keys; here the host sets them (setInput) before update(dt) -- from a recorded session, or from whatever toolkit owns its window.
// Left button: look around; middle: orbit the anchor under the pointer; right: pan; keys: move.  Same poses as the reference's class,
// bit for bit (csrc/gs_camera.cpp).
class CameraControls
{
    GsCameraControls* m_Handle{nullptr};
    GsCameraInput m_Input{};

  public:
    CameraControls(float screenWidth, float screenHeight) : m_Handle(gs_camera_controls_create(screenWidth, screenHeight))
    {
        if (m_Handle == nullptr) throw std::bad_alloc();
    }
    ~CameraControls() { gs_camera_controls_destroy(m_Handle); }
    CameraControls(const CameraControls&) = delete;
    CameraControls& operator=(const CameraControls&) = delete;
    void setInput(const GsCameraInput& input) { m_Input = input; }
    void setBounds(const vec3_t& min, const vec3_t& max) { gs_camera_controls_set_bounds(m_Handle, &min.x, &max.x); }
    void update(float dt) { gs_camera_controls_update(m_Handle, &m_Input, dt); }
    vec3_t getPosition() const
    {
        GsCameraControlsState s;
        gs_camera_controls_get_state(m_Handle, &s);
        return vec3_t{s.position[0], s.position[1], s.position[2]};
    }
    GsCameraControlsState getState() const
    {
        GsCameraControlsState s;
        gs_camera_controls_get_state(m_Handle, &s);
        return s;
    }
    // getView / getProjection / getViewProjection + the CameraData fill of the frame loop (Demo.cpp:377-392) in one call
    CameraData getCameraData() const
    {
        CameraData c;
        gs_camera_controls_camera_data(m_Handle, &c);
        return c;
    }
};

// Utilities.h:113-153 CudaGraphicsResource: a GL buffer object registered with the device; getBinding<T>() maps it for the scope of the
// binding and hands out the device pointer (Demo.cpp:338,396,419: `globalArgs.backBuffer = colorBinding.getPtr()`).  The GL side of the
// presentation (GLBuffer, GLVertexArray, the textured quad of Demo.cpp:481-525) is plain GL and stays the host's own code.
struct GraphicsResource
{
  private:
    GsGraphicsResource* m_Resource;

  public:
    template <typename T>
    struct Binding
    {
      private:
        T* m_Ptr;
        GsGraphicsResource* m_Resource;

      public:
        explicit Binding(GsGraphicsResource* resource) : m_Ptr(static_cast<T*>(gs_graphics_map(resource, nullptr))), m_Resource(resource) {}
        ~Binding() { gs_graphics_unmap(m_Resource); }
        Binding(const Binding&) = delete;
        Binding& operator=(const Binding&) = delete;
        T* getPtr() const { return m_Ptr; }
    };
    GraphicsResource(std::uint32_t bufferId, std::uint32_t flags) : m_Resource(gs_graphics_gl_register_buffer(bufferId, flags)) {}
    ~GraphicsResource() { gs_graphics_unregister(m_Resource); }
    GraphicsResource(const GraphicsResource&) = delete;
    GraphicsResource& operator=(const GraphicsResource&) = delete;
    bool registered() const { return m_Resource != nullptr; } // addition: the reference finds out at its first map
    template <typename T>
    Binding<T> getBinding() { return Binding<T>(m_Resource); } // (C++17: the prvalue is constructed in place, as the reference relies on)
};

} // namespace gs
