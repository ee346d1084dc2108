"""Screen-space multi-GPU split: tile-row bands, splat buffers replicated, one framebuffer gather per frame.

The reference is single-GPU (src/Demo.cpp); this is the one parallel strategy the port adds.
Rank r renders tile rows bands[r] of the frame into its own full-size framebuffer tensor; `gather_bands` then
moves every band to rank 0 with ONE grouped send/recv set (torch.distributed P2P -> RCCL over xGMI on GPUs, gloo on
CPU for the tests).  xGMI is point to point, so the root's receives land over distinct links concurrently and go
straight into the destination rows of its framebuffer (no staging copy, no ring).
"""
import torch.distributed as dist

from . import TILE_SIZE
from .scenes import tile_row_bands  # noqa: F401  (re-exported)


def pixel_rows(band, height):
    """Pixel-row range [a, b) of a tile-row band; the last band may end on a partial tile row."""
    return band[0] * TILE_SIZE, min(band[1] * TILE_SIZE, height)


def gather_bands(framebuffer, bands, rank, world, root=0):
    """framebuffer: (H, W, 4) uint8 tensor, contiguous; every rank has filled rows pixel_rows(bands[rank]).

    Returns after the transfers are ordered on the current stream (GPU) or complete (CPU); on `root` the
    framebuffer then holds the whole frame.  No-op for world == 1.
    """
    if world == 1:
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
