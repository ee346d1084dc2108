"""The N > 1 control flow of bench.py on ONE GPU: two ranks share cuda:0 and gather over gloo (`--rehearse`; RCCL needs one
GPU per rank, which the test box does not have).  Covers the band split, culled band renders with frames in flight, captured
frames driven from worker threads, the per-frame gather ordering, and the byte-for-byte check of the gathered frame against a
single-GPU render."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ranks,width,height,splats,sh_degree", [(2, 1920, 1080, 200000, 3), (3, 1920, 1080, 200000, 3),
                                                                 # the widest rehearsals the GPU box allows (six processes on the card: the ranks, this process, a margin of one):
                                                                 # 1080p with its half-height last tile row, and 4K (135 tile rows)
                                                                 (4, 1920, 1080, 120000, 3), (4, 3840, 2160, 200000, 3),
                                                                 # SH degrees 2 and 4 (rows of 27 / 75 floats: refused by the chains of four views until round 6)
                                                                 (2, 1280, 720, 80000, 2), (2, 1280, 720, 60000, 4)])
def test_bench_multirank_rehearsal(ranks, width, height, splats, sh_degree):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + ranks + 10 * sh_degree), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "24", "--warmup", "6",
           "--splats", str(splats), "--width", str(width), "--height", str(height), "--sh-degree", str(sh_degree), "--cpu-frames", "0", "--rehearse",
           "--rebalance-frames", "64", "--spin-up", "0.2"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    result = json.loads(line)
    cfg = result["config"]
    assert result["n_gpus"] == ranks and result["steps"] == 24 and result["value"] > 0 and cfg["sh_degree"] == sh_degree
    assert result["scaling"] == "strong"
    bands = cfg["tile_row_bands"]
    assert len(bands) == ranks and bands[0][0] == 0 and bands[-1][1] == (height + 15) // 16
    assert all(b[1] > b[0] for b in bands) and all(bands[i][1] == bands[i + 1][0] for i in range(ranks - 1))
    assert cfg["gathered_frame_equals_single_gpu_frame"] is True
    assert cfg["frames_per_chain"] == 4 and cfg["frames_ahead"] == 16 and cfg["frames_in_timed_region"] >= 24
    assert len(cfg["band_ms_measured_by_its_rank"]) == ranks  # the re-cut ran: every rank reported its band time
    per_rank = cfg["per_rank"]
    assert len(per_rank["timed_region_s_own_clock"]) == ranks and len(per_rank["gather_ms_per_chain_alone"]) == ranks
    assert all(v > 0 for v in per_rank["timed_region_s_own_clock"])


def test_rccl_is_loadable_and_world_one_gather_is_a_noop():
    """The library binds RCCL at run time (gs_comm.cpp): the unique id call proves librccl.so.1 loads and resolves on this box;
    a one-rank communicator gathers nothing and leaves the frame alone.  (No torch GPU tensors here: earlier tests of this
    process have loaded the library first, and torch's bundled HIP runtime would then find no device -- DESIGN.md section 8.)"""
    import ctypes as C

    import numpy as np

    import cudagaussianrenderer_amd as gs
    from cudagaussianrenderer_amd import multigpu

    lib = gs.lib()
    lib.gs_set_device(0)
    ident = C.create_string_buffer(128)
    assert lib.gs_comm_unique_id(ident) == 0 and any(ident.raw)
    comm = multigpu.BandComm(0, 1)
    fb = gs.DeviceBuffer(np.full((64, 48, 4), 7, np.uint8))
    bands = (C.c_int32 * 2)(0, 4)
    assert lib.gs_gather_bands(comm.handle, fb.ptr, 48, 64, bands, 0, None) == 0
    gs.synchronize()
    assert (fb.copy_to(dtype=np.uint8) == 7).all()
    comm.close()


_SELF_EXCHANGE = """
import sys
import numpy as np
if sys.argv[1] == "torch-first":
    import torch  # bench.py's order: torch's bundled HIP runtime and RCCL are in the process before the library binds them
import cudagaussianrenderer_amd as gs
lib = gs.lib()
lib.gs_set_device(0)
rows = np.random.default_rng(3).integers(0, 256, (144, 1920, 4), dtype=np.uint8)  # 9 tile rows of a 1080p frame
src = gs.DeviceBuffer(rows)
dst = gs.DeviceBuffer(np.zeros_like(rows))
rc = lib.gs_debug_rccl_self_exchange(dst.ptr, src.ptr, rows.nbytes, None)
assert rc == 0, rc
assert np.array_equal(dst.copy_to(dtype=np.uint8).reshape(rows.shape), rows)
# four send/receive pairs with one peer inside one group: what a chain of four views asks of gs_gather_bands_views
dst4 = gs.DeviceBuffer(np.zeros_like(rows))
rc = lib.gs_debug_rccl_self_exchange_parts(dst4.ptr, src.ptr, rows.nbytes, 4, None)
assert rc == 0, rc
assert np.array_equal(dst4.copy_to(dtype=np.uint8).reshape(rows.shape), rows)
# four communicators on four streams, all in flight, each used twice: bench.py's one communicator per frame in flight
dst5 = gs.DeviceBuffer(np.zeros_like(rows))
rc = lib.gs_debug_rccl_self_exchange_streams(dst5.ptr, src.ptr, rows.nbytes, 4)
assert rc == 0, rc
assert np.array_equal(dst5.copy_to(dtype=np.uint8).reshape(rows.shape), rows)
print("exchanged", rows.nbytes)
"""


@pytest.mark.parametrize("order", ["library-only", "torch-first"])
def test_rccl_send_recv_path_on_one_gpu(order):
    """The grouped ncclSend/ncclRecv calls of gs_gather_bands, through the run-time-bound RCCL, on a one-rank communicator that sends
    a band-sized block to itself: proves binding, signatures and stream handling on the hardware the test box has -- in a process
    of its own (one HIP runtime per process: a test process that loaded the library before torch has two), without torch (a C++
    host) and with torch imported first (bench.py: the library and its RCCL then bind torch's bundled runtime)."""
    out = subprocess.run([sys.executable, "-c", _SELF_EXCHANGE, order], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "exchanged" in out.stdout, (out.stdout[-1000:], out.stderr[-2000:])


def _two_gpu_worker(rank, world, port, out_path):
    import numpy as np
    import torch
    import torch.distributed as dist

    import cudagaussianrenderer_amd as gs
    from cudagaussianrenderer_amd import multigpu, scenes

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(rank)
    gs.lib().gs_set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        w, h = 640, 368
        pos, sr, col = gs.generate_random_gaussians(20000, seed=5, min_scale=0.02, max_scale=0.3)
        cam = gs.camera_from_bounds([-4] * 3, [4] * 3, w, h)
        bands = scenes.tile_row_bands(gs.tile_counts(w, h)[1], world)
        comm = multigpu.BandComm(rank, world, torch.device("cuda", rank))
        r = gs.Renderer(w, h, tile_rows=bands[rank])
        r.set_scene(pos, sr, col)
        fb = torch.zeros((h, w, 4), dtype=torch.uint8, device=f"cuda:{rank}")
        r.set_back_buffer(fb.data_ptr())
        stream = torch.cuda.Stream()
        r.set_stream(stream.cuda_stream)
        for _ in range(3):
            r.render(cam)
            with torch.cuda.stream(stream):
                multigpu.gather_bands(fb, bands, rank, world, comm=comm)
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 0:
            np.save(out_path, fb.cpu().numpy())
        comm.close()
        r.close()
    finally:
        dist.destroy_process_group()


def test_library_gather_on_two_gpus(tmp_path):
    """gs_gather_bands over real RCCL: needs two GPUs (skipped on the one-GPU test box; the driver's scaling run exercises the
    same call through bench.py and checks the gathered frame there)."""
    import numpy as np
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp

    import cudagaussianrenderer_amd as gs

    out = str(tmp_path / "frame.npy")
    mp.spawn(_two_gpu_worker, args=(2, 29641, out), nprocs=2, join=True)
    pos, sr, col = gs.generate_random_gaussians(20000, seed=5, min_scale=0.02, max_scale=0.3)
    cam = gs.camera_from_bounds([-4] * 3, [4] * 3, 640, 368)
    r = gs.Renderer(640, 368)
    r.set_scene(pos, sr, col)
    r.render(cam)
    assert np.array_equal(np.load(out), r.read_image())
    r.close()
