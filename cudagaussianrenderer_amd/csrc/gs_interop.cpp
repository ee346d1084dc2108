// gs_interop.cpp -- the one CUDA-specific piece of the reference's presentation: `CudaGraphicsResource` (src/Utilities.h:113-153), the GL
// pixel-unpack buffer registered with the device so that the frame is rendered straight into it (Demo.cpp:311,338,396,419).  Everything
// else of Demo.cpp:481-525 is plain GL and stays with the host as it is.  HIP carries the same interop (hip_gl_interop.h); the host needs
// no HIP header for it: a GL buffer id goes in, a device pointer comes out, which is what GlobalArgs.backBuffer /
// gs_renderer_set_back_buffer take.  SURVEY.md section 8(f4).  Host-only source (no kernels).
#include <cstdio>
#include <cstdlib>

#include <hip/hip_runtime_api.h>
#include <hip/hip_gl_interop.h>

#include "gsplat_hip.h"

struct GsGraphicsResource
{
    hipGraphicsResource* resource;
    bool mapped;
};

namespace
{
// the reference's checkCudaErrors (CudaErrorCheck.cpp:2-11): print and exit
void check(hipError_t e, const char* what)
{
    if (e == hipSuccess) return;
    std::fprintf(stderr, "HIP error in %s: %s\n", what, hipGetErrorString(e));
    std::exit(EXIT_FAILURE);
}
} // namespace

extern "C" GsGraphicsResource* gs_graphics_gl_register_buffer(uint32_t glBufferId, uint32_t flags)
{
    // Utilities.h:139-142.  The reference ignores cudaGraphicsGLRegisterBuffer's result and dies at the first map (checkCudaErrors,
    // :127); here the failed registration is reported where it happens and the handle is NULL -- mapping it ends the process the same way.
    hipGraphicsResource* resource = nullptr;
    const hipError_t e = hipGraphicsGLRegisterBuffer(&resource, (GLuint) glBufferId, flags);
    if (e != hipSuccess || resource == nullptr)
    {
        (void) hipGetLastError();
        std::fprintf(stderr, "gsplat_hip: hipGraphicsGLRegisterBuffer(%u) failed: %s (is a GL context current on this thread?)\n", glBufferId,
                     hipGetErrorString(e));
        return nullptr;
    }
    return new GsGraphicsResource{resource, false};
}

extern "C" void* gs_graphics_map(GsGraphicsResource* r, size_t* sizeInBytesOrNull)
{
    // Utilities.h:122-130 Binding(): map on the library's current stream (the reference: stream 0), then the mapped pointer
    if (r == nullptr) check(hipErrorInvalidResourceHandle, "gs_graphics_map (the buffer was never registered)");
    check(hipGraphicsMapResources(1, &r->resource, (hipStream_t) gs_get_stream()), "hipGraphicsMapResources");
    r->mapped = true;
    void* ptr = nullptr;
    size_t size = 0;
    check(hipGraphicsResourceGetMappedPointer(&ptr, &size, r->resource), "hipGraphicsResourceGetMappedPointer");
    if (sizeInBytesOrNull != nullptr) *sizeInBytesOrNull = size;
    return ptr;
}

extern "C" void gs_graphics_unmap(GsGraphicsResource* r)
{
    // Utilities.h:131-134 ~Binding(): unchecked there, and here
    if (r == nullptr || !r->mapped) return;
    (void) hipGraphicsUnmapResources(1, &r->resource, (hipStream_t) gs_get_stream());
    r->mapped = false;
}

extern "C" void gs_graphics_unregister(GsGraphicsResource* r)
{
    // Utilities.h:143-146
    if (r == nullptr) return;
    gs_graphics_unmap(r);
    check(hipGraphicsUnregisterResource(r->resource), "hipGraphicsUnregisterResource");
    delete r;
}
