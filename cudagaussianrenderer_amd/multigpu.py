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
    (which is only the bootstrap here: the per-frame transfers are the library's own grouped ncclSend/ncclRecv).

    Every rank takes part in the same collectives whatever happens locally: rank 0 broadcasts a status byte with the id (a failed
    gs_comm_unique_id travels as status 0 instead of leaving the other ranks inside the broadcast), and after gs_comm_create the
    ranks agree (MIN) on whether everybody has a communicator -- so either all ranks get a BandComm or all of them raise."""

    def __init__(self, rank, world, device=None):
        import torch

        lib = _capi.load()
        buf = C.create_string_buffer(128)
        self.handle = None
        self.rank, self.world = rank, world
        if world > 1:
            where = device if device is not None else "cpu"
            packet = torch.zeros(129, dtype=torch.uint8, device=where)  # [0] = status, [1:] = the id
            if rank == 0 and lib.gs_comm_unique_id(buf) == 0:
                packet.copy_(torch.frombuffer(bytearray(b"\x01" + buf.raw), dtype=torch.uint8))
            dist.broadcast(packet, src=0)
            host = packet.cpu().numpy().tobytes()
            if host[0] != 1:
                raise RuntimeError("gs_comm_unique_id failed on rank 0 (librccl.so not loadable?)")
            buf = C.create_string_buffer(host[1:], 128)
        handle = lib.gs_comm_create(rank, world, buf)
        if world > 1:
            ok = torch.tensor([1 if handle else 0], dtype=torch.int32, device=where)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if handle:
                    lib.gs_comm_destroy(handle)
                raise RuntimeError("gs_comm_create failed" + ("" if not handle else " on another rank"))
        elif not handle:
            raise RuntimeError("gs_comm_create failed")
        self.handle = handle

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


# ---- the control flow of an N-rank run (used by bench.py with GPU renderers, and by tests/test_multigpu_cpu.py with the CPU oracle) ----
#
# A "rig" is whatever renders this rank's band: it offers run_frames(count, gather=True) -> frames rendered, sync_all(),
# local_frame_ms(frames) -> this rank's own milliseconds per frame without the gather, close(), and optionally
# gather_frame_ms() -> what the framebuffer exchange costs this rank's GPU per frame when nothing overlaps it (a collective: every
# rank calls it, in the same place).  Every function below is a fixed
# sequence of collectives that all ranks enter in the same order whatever their local results: decisions are taken from all-gathered
# or all-reduced values only, so no rank can leave a loop that another rank is still in.
def all_gather_floats(value, world, tensor_device):
    """[value of rank 0, ..., value of rank world-1], the same list on every rank.  `value` may be a tuple of floats (the same
    length on every rank): the result is then a list of tuples -- ONE collective whatever the length."""
    import torch

    scalar = not isinstance(value, (tuple, list))
    mine_host = [float(value)] if scalar else [float(v) for v in value]
    if world == 1:
        return [mine_host[0] if scalar else tuple(mine_host)]
    mine = torch.tensor(mine_host, dtype=torch.float64, device=tensor_device)
    everyone = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    rows = [t.cpu().tolist() for t in everyone]
    return [row[0] if scalar else tuple(row) for row in rows]


def rebalance_bands(make_rig, rig, bands, row_cost, world, rounds, tensor_device, frames=256, log=None, gather_weight=1.0):
    """Re-cut cost-balanced bands from what every rank MEASURES for its own band (untimed set-up).  The cost model behind the first
    cut does not know the kernels' fixed costs; here the rows of a band that took longer than the mean get more weight (exponent 2:
    only part of a band's time follows its rows), the cut with the smallest slowest band is kept.  Also absorbs speed differences
    between the GPUs.  A rank's time is its band rendered WITHOUT the exchange (with it, every rank would measure the lock-step of
    the slowest) PLUS gather_weight x what the exchange alone costs its GPU per frame (rig.gather_frame_ms, if the rig has it):
    rank 0 receives world - 1 bands per frame with RCCL kernels that run on the GPU that also renders band 0, so its band must
    come out smaller by that much.  make_rig(bands) builds a rig for a new cut; returns (rig, bands, per-rank milliseconds of the
    kept cut -- exchange share included)."""
    from .scenes import balanced_tile_row_bands

    cost = [float(c) for c in row_cost]
    best = None
    sees_exchange = gather_weight > 0 and world > 1 and hasattr(rig, "gather_frame_ms")  # the same on every rank
    for it in range(rounds + 1):
        own = rig.local_frame_ms(frames)
        exchange = rig.gather_frame_ms() if sees_exchange else 0.0
        # one collective per round whoever prints: `log` may be set on some ranks only
        both = all_gather_floats((own + gather_weight * exchange, exchange), world, tensor_device)
        times = [b[0] for b in both]
        if log:
            shares = [b[1] for b in both] if sees_exchange else []
            log(f"rebalance round {it}: bands {bands} ms {[round(t, 4) for t in times]}" +
                (f" of which exchange {[round(t, 4) for t in shares]}" if shares else ""))
        if best is None or max(times) < best[0]:
            best = (max(times), bands, times)
        if it == rounds:
            break
        mean = sum(times) / len(times)
        for (a, b), t in zip(bands, times):
            for y in range(a, b):
                cost[y] *= (t / mean) ** 2.0
        recut = balanced_tile_row_bands(cost, world)
        if recut == bands:
            break
        bands = recut
        rig.close()
        rig = make_rig(bands)
    if bands != best[1]:
        bands = best[1]
        rig.close()
        rig = make_rig(bands)
    return rig, bands, best[2]


def spin_up(rig, seconds, world, tensor_device, chunk=256):
    """Untimed rendering for about `seconds` (a freshly leased GPU starts from idle clocks); with several ranks everybody leaves after
    the same number of frames -- each frame is a matched send/receive set -- i.e. when the SLOWEST clock says the time is up."""
    import time

    import torch

    if seconds <= 0:
        return 0
    frames, t0 = 0, time.perf_counter()
    while True:
        frames += rig.run_frames(chunk)
        rig.sync_all()
        done = time.perf_counter() - t0 >= seconds
        if world > 1:
            flag = torch.tensor([1 if done else 0], dtype=torch.int32, device=tensor_device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            done = bool(int(flag.item()))
        if done:
            return frames


def timed_frames(rig, steps, warmup, world, tensor_device):
    """W untimed warm-up frames, then `steps` frames between barrier + synchronize pairs (rig.sync_all); MAX over ranks.  With V frames
    per chain both counts are rounded UP to whole chains by the rig: the timed region then holds >= steps frames of work while the
    caller still divides `steps` by the time -- never in the measurement's favour.  Returns (seconds, frames in the timed region,
    this rank's own seconds)."""
    import time

    import torch

    rig.run_frames(warmup)
    rig.sync_all()
    t0 = time.perf_counter()
    frames = rig.run_frames(steps)
    rig.sync_all()
    mine = time.perf_counter() - t0
    elapsed = mine
    if world > 1:
        t = torch.tensor([mine], dtype=torch.float64, device=tensor_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, frames, mine


class Watchdog:
    """A rank that makes no progress for `seconds` prints where it was and ends the process (exit code 5): under a launcher
    (torch.distributed.run) the other ranks are then torn down too, so a hang anywhere in an N-rank job -- a peer that died, a
    collective that never completes -- becomes a diagnosable failure instead of a run that waits for ever.  `beat(what)` marks
    progress; the thread is a daemon and costs nothing between beats."""

    def __init__(self, seconds, rank, stream=None):
        import sys
        import threading
        import time

        self.seconds, self.rank, self.what, self.last = float(seconds), rank, "start", time.monotonic()
        self._stop = threading.Event()
        self._out = stream if stream is not None else sys.stderr
        if self.seconds > 0:
            self._thread = threading.Thread(target=self._watch, name="gsplat-watchdog", daemon=True)
            self._thread.start()

    def beat(self, what):
        import time

        self.what, self.last = what, time.monotonic()

    def _watch(self):
        import os
        import time

        while not self._stop.wait(min(1.0, self.seconds / 4)):
            idle = time.monotonic() - self.last
            if idle > self.seconds:
                print(f"gsplat watchdog: rank {self.rank} made no progress for {idle:.0f} s in '{self.what}': giving up (exit 5)",
                      file=self._out, flush=True)
                os._exit(5)

    def stop(self):
        self._stop.set()
