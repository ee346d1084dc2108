"""N > 1 path on CPU: world_size-2 (and 3) `gloo` processes split a frame into tile-row bands, each rank produces
its band (here with the CPU oracle standing in for the GPU renderer -- test infrastructure only), the bands are
gathered with the same `gather_bands` bench.py uses, and rank 0 must hold the byte-identical full frame."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, width, height, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import cudagaussianrenderer_amd as gs
    from cudagaussianrenderer_amd import multigpu, scenes
    from oracle import gsref

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pos, sr, col = gs.generate_random_gaussians(3000, seed=1234, min_scale=0.02, max_scale=0.4)
        cam = gsref.camera_from_bounds([-4] * 3, [4] * 3, width, height)
        tx, ty = gs.tile_counts(width, height)
        bands = scenes.tile_row_bands(ty, world)
        xy, z, ell, con = gsref.evaluate_splat_clip_data(pos, sr, cam, width, height)
        # every rank builds ONLY its band's list (as the GPU path does) and rasterises only its rows
        k, v, _ = gsref.build_tile_list(xy, z, ell, width, height, rows=bands[rank])
        k, v = gsref.sort_tile_list(k, v, gsref.sort_end_bit(width, height))
        rng = gsref.evaluate_tile_ranges(k, tx * ty)
        img, _ = gsref.rasterize_tiles(width, height, rng, v, xy, con, col, rows=bands[rank])
        fb = torch.from_numpy(img.copy())
        a, b = multigpu.pixel_rows(bands[rank], height)
        assert not fb[:a].any() and not fb[b:].any()  # a rank touches only its own rows
        multigpu.gather_bands(fb, bands, rank, world)
        dist.barrier()
        if rank == 0:
            np.save(out_path, fb.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,width,height", [(2, 320, 200), (3, 256, 256)])
def test_band_split_and_gather_matches_single_process(tmp_path, world, width, height):
    import torch.multiprocessing as mp

    sys.path.insert(0, ROOT)
    import cudagaussianrenderer_amd as gs
    from oracle import gsref

    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), width, height, out), nprocs=world, join=True)
    got = np.load(out)
    pos, sr, col = gs.generate_random_gaussians(3000, seed=1234, min_scale=0.02, max_scale=0.4)
    cam = gsref.camera_from_bounds([-4] * 3, [4] * 3, width, height)
    want, _, _ = gsref.render(pos, sr, col, None, 0, 0, cam, width, height)
    assert np.array_equal(got, want)


def _views_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from cudagaussianrenderer_amd import multigpu, scenes

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        views, height, width = 3, 200, 64
        bands = scenes.tile_row_bands((height + 15) // 16, world)
        a, b = multigpu.pixel_rows(bands[rank], height)
        fb = torch.zeros((views, height, width, 4), dtype=torch.uint8)
        for v in range(views):
            fb[v, a:b] = 10 * (v + 1) + rank  # what this rank "rendered" of view v
        multigpu.gather_bands(fb, bands, rank, world)  # (V, H, W, 4): the images of one chain of views
        dist.barrier()
        if rank == 0:
            np.save(out_path, fb.numpy())
    finally:
        dist.destroy_process_group()


def test_gather_of_a_chain_of_views(tmp_path):
    """bench.py on N > 1 renders several frames per chain of kernels (gs_renderer_set_views) and gathers their images together: the
    stacked (V, H, W, 4) form of gather_bands must put every rank's rows of every view into rank 0's stack."""
    import torch.multiprocessing as mp

    sys.path.insert(0, ROOT)
    from cudagaussianrenderer_amd import multigpu, scenes

    world, views, height = 2, 3, 200
    out = str(tmp_path / "views.npy")
    mp.spawn(_views_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    bands = scenes.tile_row_bands((height + 15) // 16, world)
    for v in range(views):
        for rank in range(world):
            a, b = multigpu.pixel_rows(bands[rank], height)
            assert (got[v, a:b] == 10 * (v + 1) + rank).all()
