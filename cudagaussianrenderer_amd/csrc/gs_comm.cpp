// gs_comm.cpp -- the one exchange step of the multi-GPU split: every rank renders a band of tile rows
// into its own full-size back buffer, and one grouped RCCL send/receive set per frame moves the bands' pixel rows to the root
// rank, straight into the destination rows of its back buffer (xGMI is point to point: the root's receives arrive over distinct
// links side by side; no ring, no staging copy).  The reference is single-GPU (src/Demo.cpp): this has no counterpart there.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the library torch.distributed uses too), so libgsplat_hip.so itself has no
// link-time dependency on it and a single-GPU host never loads it.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "gsplat_hip.h"
#include "gsplat_hip_debug.h"

namespace
{
struct Rccl
{
    void* handle = nullptr;
    ncclResult_t (*getUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*commInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*commDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*groupStart)() = nullptr;
    ncclResult_t (*groupEnd)() = nullptr;
    ncclResult_t (*send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*errorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"})
        {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle)
        {
            std::fprintf(stderr, "gsplat_hip: librccl.so not found (%s)\n", dlerror());
            return;
        }
        auto sym = [&](const char* s) { return dlsym(r.handle, s); };
        r.getUniqueId = (decltype(r.getUniqueId)) sym("ncclGetUniqueId");
        r.commInitRank = (decltype(r.commInitRank)) sym("ncclCommInitRank");
        r.commDestroy = (decltype(r.commDestroy)) sym("ncclCommDestroy");
        r.groupStart = (decltype(r.groupStart)) sym("ncclGroupStart");
        r.groupEnd = (decltype(r.groupEnd)) sym("ncclGroupEnd");
        r.send = (decltype(r.send)) sym("ncclSend");
        r.recv = (decltype(r.recv)) sym("ncclRecv");
        r.errorString = (decltype(r.errorString)) sym("ncclGetErrorString");
        r.ok = r.getUniqueId && r.commInitRank && r.commDestroy && r.groupStart && r.groupEnd && r.send && r.recv && r.errorString;
        if (!r.ok) std::fprintf(stderr, "gsplat_hip: librccl.so lacks a required entry point\n");
    });
    return r;
}

bool check(ncclResult_t rc, const char* what)
{
    if (rc == ncclSuccess) return true;
    std::fprintf(stderr, "gsplat_hip: RCCL error in %s: %s\n", what, rccl().errorString ? rccl().errorString(rc) : "?");
    return false;
}
} // namespace

struct GsComm
{
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" int gs_comm_unique_id(char out[GS_COMM_UNIQUE_ID_BYTES])
{
    static_assert(GS_COMM_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    if (!rccl().ok) return -1;
    ncclUniqueId id;
    if (!check(rccl().getUniqueId(&id), "ncclGetUniqueId")) return -1;
    std::memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

extern "C" GsComm* gs_comm_create(int rank, int world, const char uniqueId[GS_COMM_UNIQUE_ID_BYTES])
{
    if (world < 1 || rank < 0 || rank >= world) return nullptr;
    GsComm* c = new GsComm{};
    c->rank = rank;
    c->world = world;
    if (world == 1) return c; // nothing to exchange: RCCL is not even loaded
    if (!rccl().ok || uniqueId == nullptr)
    {
        delete c;
        return nullptr;
    }
    ncclUniqueId id;
    std::memcpy(id.internal, uniqueId, NCCL_UNIQUE_ID_BYTES);
    if (!check(rccl().commInitRank(&c->comm, world, id, rank), "ncclCommInitRank")) // uses the calling thread's current device
    {
        delete c;
        return nullptr;
    }
    return c;
}

extern "C" void gs_comm_destroy(GsComm* c)
{
    if (!c) return;
    if (c->comm) (void) rccl().commDestroy(c->comm);
    delete c;
}

// Diagnostics: the same RCCL entry points gs_gather_bands uses (communicator of one rank, grouped send + receive to itself) on
// ONE GPU, so that the run-time binding, the call signatures and the stream handling are exercised where only one GPU exists.
extern "C" int gs_debug_rccl_self_exchange(void* dst, const void* src, size_t bytes, void* hipStream)
{
    return gs_debug_rccl_self_exchange_parts(dst, src, bytes, 1, hipStream);
}

// `parts` send/receive pairs with the same peer inside ONE group (what a chain of several views asks of the root and of every sender).
extern "C" int gs_debug_rccl_self_exchange_parts(void* dst, const void* src, size_t bytes, int parts, void* hipStream)
{
    const Rccl& r = rccl();
    if (!r.ok) return -1;
    if (parts < 1 || bytes % (size_t) parts != 0) return -1;
    ncclUniqueId id;
    if (!check(r.getUniqueId(&id), "ncclGetUniqueId")) return -2;
    ncclComm_t comm = nullptr;
    if (!check(r.commInitRank(&comm, 1, id, 0), "ncclCommInitRank")) return -2;
    const size_t part = bytes / (size_t) parts;
    bool ok = check(r.groupStart(), "ncclGroupStart");
    for (int p = 0; ok && p != parts; ++p)
    {
        ok = ok && check(r.send((const char*) src + part * (size_t) p, part, ncclUint8, 0, comm, (hipStream_t) hipStream), "ncclSend");
        ok = ok && check(r.recv((char*) dst + part * (size_t) p, part, ncclUint8, 0, comm, (hipStream_t) hipStream), "ncclRecv");
    }
    ok = check(r.groupEnd(), "ncclGroupEnd") && ok;
    if (hipStreamSynchronize((hipStream_t) hipStream) != hipSuccess) ok = false;
    (void) r.commDestroy(comm);
    return ok ? 0 : -2;
}

// What bench.py does with several frames in flight: `comms` communicators (of one rank each here), every one used on a stream of its
// own, all exchanges in flight at once, two rounds so that every communicator is used again while the others are busy.  Slice k of the
// block travels through communicator k.
extern "C" int gs_debug_rccl_self_exchange_streams(void* dst, const void* src, size_t bytes, int comms)
{
    const Rccl& r = rccl();
    if (!r.ok) return -1;
    if (comms < 1 || comms > 8 || bytes % (size_t) comms != 0) return -1;
    std::vector<ncclComm_t> comm((size_t) comms, nullptr);
    std::vector<hipStream_t> stream((size_t) comms, nullptr);
    bool ok = true;
    for (int k = 0; ok && k != comms; ++k)
    {
        ncclUniqueId id;
        ok = check(r.getUniqueId(&id), "ncclGetUniqueId") && check(r.commInitRank(&comm[k], 1, id, 0), "ncclCommInitRank") &&
             hipStreamCreateWithFlags(&stream[k], hipStreamNonBlocking) == hipSuccess;
    }
    const size_t slice = bytes / (size_t) comms;
    for (int round = 0; ok && round != 2; ++round)
        for (int k = 0; ok && k != comms; ++k)
        {
            // round 0 moves the slice to its place, round 1 moves it again from there (same bytes): both must complete in order on stream k
            const char* from = round == 0 ? (const char*) src + slice * (size_t) k : (const char*) dst + slice * (size_t) k;
            ok = check(r.groupStart(), "ncclGroupStart");
            ok = ok && check(r.send(from, slice, ncclUint8, 0, comm[k], stream[k]), "ncclSend");
            ok = ok && check(r.recv((char*) dst + slice * (size_t) k, slice, ncclUint8, 0, comm[k], stream[k]), "ncclRecv");
            ok = check(r.groupEnd(), "ncclGroupEnd") && ok;
        }
    for (int k = 0; k != comms; ++k)
    {
        if (stream[k] && hipStreamSynchronize(stream[k]) != hipSuccess) ok = false;
        if (comm[k]) (void) r.commDestroy(comm[k]);
        if (stream[k]) (void) hipStreamDestroy(stream[k]);
    }
    return ok ? 0 : -2;
}

extern "C" int gs_comm_rank(const GsComm* c) { return c ? c->rank : -1; }
extern "C" int gs_comm_world(const GsComm* c) { return c ? c->world : 0; }

extern "C" void gs_band_pixel_rows(int tileRowBegin, int tileRowEnd, int height, int* rowBegin, int* rowEnd)
{
    // the last band may end on a partial tile row (1080 = 67.5 tiles)
    int a = tileRowBegin * GS_TILE_SIZE, b = tileRowEnd * GS_TILE_SIZE;
    if (a > height) a = height;
    if (b > height) b = height;
    if (b < a) b = a;
    if (rowBegin) *rowBegin = a;
    if (rowEnd) *rowEnd = b;
}

extern "C" int gs_gather_bands(GsComm* c, uint8_t* backBuffer, int width, int height, const int32_t* tileRowBands, int root, void* hipStream)
{
    return gs_gather_bands_views(c, backBuffer, width, height, 1, tileRowBands, root, hipStream);
}

// The bands of `views` images stored one after another (gs_renderer_set_views), ONE grouped send/receive set for all of them.
extern "C" int gs_gather_bands_views(GsComm* c, uint8_t* backBuffer, int width, int height, int views, const int32_t* tileRowBands, int root,
                                     void* hipStream)
{
    if (!c || !backBuffer || !tileRowBands || root < 0 || root >= c->world || views < 1) return -1;
    if (c->world == 1) return 0;
    const size_t rowBytes = (size_t) width * 4, imageBytes = rowBytes * (size_t) height;
    hipStream_t stream = (hipStream_t) hipStream;
    const Rccl& r = rccl();
    bool ok = check(r.groupStart(), "ncclGroupStart");
    for (int v = 0; ok && v != views; ++v)
    {
        uint8_t* image = backBuffer + imageBytes * (size_t) v;
        if (c->rank == root)
        {
            for (int src = 0; ok && src != c->world; ++src)
            {
                if (src == root) continue;
                int a, b;
                gs_band_pixel_rows(tileRowBands[2 * src], tileRowBands[2 * src + 1], height, &a, &b);
                if (b > a) ok = check(r.recv(image + rowBytes * (size_t) a, rowBytes * (size_t) (b - a), ncclUint8, src, c->comm, stream), "ncclRecv");
            }
        }
        else
        {
            int a, b;
            gs_band_pixel_rows(tileRowBands[2 * c->rank], tileRowBands[2 * c->rank + 1], height, &a, &b);
            if (b > a) ok = check(r.send(image + rowBytes * (size_t) a, rowBytes * (size_t) (b - a), ncclUint8, root, c->comm, stream), "ncclSend");
        }
    }
    // the group is always closed, also after a failed call inside it
    ok = check(r.groupEnd(), "ncclGroupEnd") && ok;
    return ok ? 0 : -2;
}
