"""Screen-space multi-GPU split: tile-row bands, splat buffers replicated, one framebuffer gather per frame.

The reference is single-GPU (src/Demo.cpp); this is the one parallel strategy the port adds.
Rank r renders tile rows bands[r] of the frame into its own full-size framebuffer tensor; `gather_bands` then
moves every band to rank 0 with ONE grouped send/recv set (torch.distributed P2P -> RCCL over xGMI on GPUs, gloo on
CPU for the tests).  xGMI is point to point, so the root's receives land over distinct links concurrently and go
straight into the destination rows of its framebuffer (no staging copy, no ring).
"""
import ctypes as C

import torch.distributed as dist

from . import TILE_SIZE, _capi
from .scenes import tile_row_bands  # noqa: F401  (re-exported)


class BandComm:
    """The library's RCCL communicator for the band gather (include/gsplat_hip.h: gs_comm_*, gs_gather_bands): one rank per
    process and GPU.  The 128-byte RCCL unique id is created on rank 0 and handed to the others through torch.distributed
    (which is only the bootstrap here: the per-frame transfers are the library's own grouped ncclSend/ncclRecv)."""

    def __init__(self, rank, world, device=None):
        import torch

        lib = _capi.load()
        buf = C.create_string_buffer(128)
        if world > 1:
            ident = torch.zeros(128, dtype=torch.uint8, device=device if device is not None else "cpu")
            if rank == 0:
                if lib.gs_comm_unique_id(buf) != 0:
                    raise RuntimeError("gs_comm_unique_id failed (librccl.so not loadable?)")
                ident.copy_(torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8))
            dist.broadcast(ident, src=0)
            buf = C.create_string_buffer(bytes(ident.cpu().numpy().tobytes()), 128)
        self.handle = lib.gs_comm_create(rank, world, buf)
        if not self.handle:
            raise RuntimeError("gs_comm_create failed")
        self.rank, self.world = rank, world

    def gather(self, framebuffer, bands, root=0, stream=None):
        """framebuffer: (H, W, 4), or (V, H, W, 4) for the V images of a chain of views -- one grouped send/receive set either way."""
        import torch

        flat = (C.c_int32 * (2 * len(bands)))(*[int(v) for b in bands for v in b])
        s = stream if stream is not None else torch.cuda.current_stream(framebuffer.device).cuda_stream
        views = int(framebuffer.shape[0]) if framebuffer.dim() == 4 else 1
        h, w = int(framebuffer.shape[-3]), int(framebuffer.shape[-2])
        rc = _capi.load().gs_gather_bands_views(self.handle, framebuffer.data_ptr(), w, h, views, flat, root, s)
        if rc != 0:
            raise RuntimeError(f"gs_gather_bands_views failed ({rc})")

    def close(self):
        if self.handle:
            _capi.load().gs_comm_destroy(self.handle)
            self.handle = None


def pixel_rows(band, height):
    """Pixel-row range [a, b) of a tile-row band; the last band may end on a partial tile row."""
    return band[0] * TILE_SIZE, min(band[1] * TILE_SIZE, height)


def gather_bands(framebuffer, bands, rank, world, root=0, comm=None):
    """framebuffer: (H, W, 4) uint8 tensor, contiguous (or (V, H, W, 4): the V images of a chain of views); every rank has filled
    rows pixel_rows(bands[rank]).

    Returns after the transfers are ordered on the current stream (GPU) or complete (CPU); on `root` the
    framebuffer then holds the whole frame.  No-op for world == 1.  With `comm` (a BandComm) the transfers are the library's
    grouped ncclSend/ncclRecv (gs_gather_bands); without it, torch.distributed point-to-point operations (gloo on CPU tensors
    for the tests, and the rehearsal of several ranks on one GPU).
    """
    if world == 1:
        return
    if comm is not None and framebuffer.is_cuda:
        comm.gather(framebuffer, bands, root)  # (V, H, W, 4): the images of a chain of views in ONE grouped set
        return
    if framebuffer.dim() == 4:
        for image in framebuffer:
            gather_bands(image, bands, rank, world, root, comm)
        return
    height = framebuffer.shape[0]
    if framebuffer.is_cuda and dist.get_backend() == "gloo":
        # Rehearsal of the multi-rank flow without RCCL (several ranks sharing one GPU): gloo moves host tensors only,
        # so the band goes through a host copy.  Never the measured path.
        if rank == root:
            for src in range(world):
                a, b = pixel_rows(bands[src], height)
                if src != root and b > a:
                    staging = framebuffer[a:b].cpu()
                    dist.recv(staging, src)
                    framebuffer[a:b].copy_(staging)
        else:
            a, b = pixel_rows(bands[rank], height)
            if b > a:
                dist.send(framebuffer[a:b].cpu(), root)
        return
    # The per-frame set of transfers never changes for a given framebuffer: build the P2POp list once and reuse it (the
    # descriptors only name the op, the tensor view and the peer), so a frame costs one batch_isend_irecv call.
    key = (framebuffer.data_ptr(), tuple(tuple(b) for b in bands), rank, world, root)
    ops = _OPS_CACHE.get(key)
    if ops is None:
        ops = []
        if rank == root:
            for src in range(world):
                if src == root:
                    continue
                a, b = pixel_rows(bands[src], height)
                if b > a:
                    ops.append(dist.P2POp(dist.irecv, framebuffer[a:b], src))
        else:
            a, b = pixel_rows(bands[rank], height)
            if b > a:
                ops.append(dist.P2POp(dist.isend, framebuffer[a:b], root))
        _OPS_CACHE[key] = ops
    if ops:
        for work in dist.batch_isend_irecv(ops):
            work.wait()


_OPS_CACHE = {}
