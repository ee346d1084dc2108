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


# ---- the N-rank control flow of bench.py (multigpu.rebalance_bands / spin_up / timed_frames / all_gather_floats) at world 8 ----------
class _OracleRig:
    """What bench.Rig is to a GPU, with the CPU oracle rendering this rank's band: chains of V views, gathered as one stacked image."""

    def __init__(self, scene, cameras, bands, rank, world, width, height, views):
        import torch

        self.scene, self.cameras, self.bands, self.rank, self.world = scene, cameras, bands, rank, world
        self.W, self.H, self.V = width, height, views
        self.fb = torch.zeros((views, height, width, 4), dtype=torch.uint8)
        self.next_chain, self.closed = 0, False

    def _render_chain(self, chain):
        from oracle import gsref

        import cudagaussianrenderer_amd as gs

        pos, sr, col = self.scene
        tx, ty = gs.tile_counts(self.W, self.H)
        band = self.bands[self.rank]
        for v in range(self.V):
            cam = self.cameras[(chain * self.V + v) % len(self.cameras)]
            xy, z, ell, con = gsref.evaluate_splat_clip_data(pos, sr, cam, self.W, self.H)
            k, val, _ = gsref.build_tile_list(xy, z, ell, self.W, self.H, rows=band)
            k, val = gsref.sort_tile_list(k, val, gsref.sort_end_bit(self.W, self.H))
            rng = gsref.evaluate_tile_ranges(k, tx * ty)
            img, _ = gsref.rasterize_tiles(self.W, self.H, rng, val, xy, con, col, rows=band)
            self.fb[v] = 0
            self.fb[v].copy_(__import__("torch").from_numpy(img))

    def run_frames(self, count, gather=True):
        from cudagaussianrenderer_amd import multigpu

        chains = (count + self.V - 1) // self.V
        for chain in range(self.next_chain, self.next_chain + chains):
            self._render_chain(chain)
            if gather:
                multigpu.gather_bands(self.fb, self.bands, self.rank, self.world)
        self.next_chain += chains
        return chains * self.V

    def sync_all(self):
        import torch.distributed as dist

        dist.barrier()

    def local_frame_ms(self, frames):
        import time

        t0 = time.perf_counter()
        done = self.run_frames(frames, gather=False)
        # a deterministic "measurement" on top of the real one, so that the re-cut takes the same path in every run of the test:
        # half of a band's time follows its rows (the rest is fixed cost, as on the GPU), ranks 0 and world-1 are three times slower
        a, b = self.bands[self.rank]
        rows_per_rank = self.bands[-1][1] / self.world
        return 1e3 * (time.perf_counter() - t0) / done * 1e-6 + (3.0 if self.rank in (0, self.world - 1) else 1.0) * (0.5 + 0.5 * (b - a) / rows_per_rank)

    def gather_frame_ms(self):
        """The exchange's share of a rank's frame (a collective in the real rig: here one barrier).  The root pays a fixed extra cost per
        frame -- its world - 1 receives --, nobody else does: the re-cut must shrink the root's band for it."""
        import torch.distributed as dist

        dist.barrier()
        return 2.5 if self.rank == 0 else 0.0

    def close(self):
        self.closed = True


def _control_flow_worker(rank, world, port, width, height, views, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import cudagaussianrenderer_amd as gs
    from cudagaussianrenderer_amd import multigpu, scenes
    from oracle import gsref

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dog = multigpu.Watchdog(120.0, rank)
    try:
        scene = gs.generate_random_gaussians(1500, seed=77, min_scale=0.02, max_scale=0.3)
        cams = [gsref.camera_from_pose(p, q, width, height) for p, q in scenes.orbit_camera_poses([-4] * 3, [4] * 3, 8)]
        tx, ty = gs.tile_counts(width, height)
        row_cost = [1.0 + (y % 5) for y in range(ty)]
        bands = scenes.balanced_tile_row_bands(row_cost, world)
        assert len(bands) == world and bands[0][0] == 0 and bands[-1][1] == ty
        made = []

        def make(b):
            made.append(b)
            return _OracleRig(scene, cams, b, rank, world, width, height, views)

        first_cut = bands
        # the same re-cut twice: blind to the exchange (round 3), then with the root's share of it (rig.gather_frame_ms) in its time
        rig = make(bands)
        rig, blind_bands, _ = multigpu.rebalance_bands(make, rig, bands, row_cost, world, 2, "cpu", frames=views, gather_weight=0.0)
        rig.close()
        del made[:]
        rig = make(first_cut)
        # `log` on rank 0 only: printing must not add a collective that the silent ranks never enter (this would hang)
        lines = []
        rig, bands, times = multigpu.rebalance_bands(make, rig, first_cut, row_cost, world, 2, "cpu", frames=views,
                                                     log=lines.append if rank == 0 else None)
        dog.beat("rebalanced")
        assert (len(lines) >= 1 and "of which exchange" in lines[0]) if rank == 0 else not lines
        assert len(times) == world and len(bands) == world and bands[-1][1] == ty
        # the slow outer ranks must have lost rows to the inner ones
        first_cut = made[0]
        assert (bands[0][1] - bands[0][0]) <= (first_cut[0][1] - first_cut[0][0])
        # ... and the root, which also pays for the exchange, more than in the cut that does not see the gather (every band is timed
        # without it either way)
        assert times[0] >= 2.5 and (bands[0][1] - bands[0][0]) < (blind_bands[0][1] - blind_bands[0][0]), (bands, blind_bands, times)
        assert multigpu.spin_up(rig, 0.05, world, "cpu", chunk=views) >= views
        elapsed, frames, mine = multigpu.timed_frames(rig, 2, views, world, "cpu")
        assert frames == -(-2 // views) * views and elapsed >= mine > 0  # steps round up to whole chains of V frames; MAX over ranks
        everyone = multigpu.all_gather_floats(float(rank), world, "cpu")
        assert everyone == [float(r) for r in range(world)]
        assert multigpu.all_gather_floats((float(rank), 0.5), world, "cpu") == [(float(r), 0.5) for r in range(world)]
        # one more chain through the same path: rank 0 must hold all V complete frames
        rig.next_chain = 0
        rig.run_frames(views)
        dist.barrier()
        if rank == 0:
            np.save(out_path, rig.fb.numpy())
            np.save(out_path + ".bands.npy", np.array(bands, np.int32))
    finally:
        dog.stop()
        dist.destroy_process_group()


@pytest.mark.parametrize("width,height,views", [(1920, 1080, 4), (3840, 2160, 1)])
def test_bench_control_flow_at_world_8(tmp_path, width, height, views):
    """BASELINE.json configs[3] and [4] in their 8-rank shape, on CPU (gloo), through the very functions bench.py drives an N-rank run
    with: eight bands (1080p: 68 tile rows, the last one half height; 4K: 135 rows), the re-cut from all-gathered band times, the
    rank-synchronised spin-up, chains of V = 4 frames gathered as one stacked image, the timed region with MAX over ranks -- and the
    gathered frames byte-identical to single-process frames.  No rank can leave a loop another rank is still in: the test would hang
    (and its watchdog end it) otherwise."""
    import torch.multiprocessing as mp

    sys.path.insert(0, ROOT)
    import cudagaussianrenderer_amd as gs
    from cudagaussianrenderer_amd import scenes
    from oracle import gsref

    world = 8
    out = str(tmp_path / "frames.npy")
    mp.spawn(_control_flow_worker, args=(world, _free_port(), width, height, views, out), nprocs=world, join=True)
    got = np.load(out)
    bands = np.load(out + ".bands.npy")
    ty = gs.tile_counts(width, height)[1]
    assert bands.shape == (world, 2) and bands[0, 0] == 0 and bands[-1, 1] == ty and (bands[1:, 0] == bands[:-1, 1]).all()
    assert (bands[:, 1] > bands[:, 0]).all()
    pos, sr, col = gs.generate_random_gaussians(1500, seed=77, min_scale=0.02, max_scale=0.3)
    cams = [gsref.camera_from_pose(p, q, width, height) for p, q in scenes.orbit_camera_poses([-4] * 3, [4] * 3, 8)]
    for v in range(views):
        want, _, _ = gsref.render(pos, sr, col, None, 0, 0, cams[v], width, height)
        assert want[..., :3].any()
        assert np.array_equal(got[v], want), v


def test_watchdog_ends_a_rank_that_stops_making_progress():
    """multigpu.Watchdog: beats keep the process alive; a stall longer than the limit prints where the rank was and exits with code 5
    (a launcher then tears the other ranks down) -- nothing in an N-rank bench run can wait for ever."""
    import subprocess

    code = ("import sys, time\n"
            f"sys.path.insert(0, {ROOT!r})\n"
            "from cudagaussianrenderer_amd import multigpu\n"
            "dog = multigpu.Watchdog(1.0, 3)\n"
            "for i in range(4):\n"
            "    time.sleep(0.5); dog.beat(f'chain {i}')\n"
            "print('alive after 2 s of beats', flush=True)\n"
            "time.sleep(30)\n"
            "print('not reached')\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 5, (out.returncode, out.stderr[-500:])
    assert "alive after 2 s of beats" in out.stdout and "not reached" not in out.stdout
    assert "rank 3 made no progress" in out.stderr and "'chain 3'" in out.stderr
    quiet = subprocess.run([sys.executable, "-c", code.replace("Watchdog(1.0, 3)", "Watchdog(0, 3)").replace("time.sleep(30)", "time.sleep(1.5)")],
                           capture_output=True, text=True, timeout=120)
    assert quiet.returncode == 0 and "not reached" in quiet.stdout  # limit 0 = off
