#!/usr/bin/env python3
"""Estimate multi-GPU scaling on ONE GPU: render each rank's tile-row band of the C3 scene in turn (no gather) and
report the per-band frame time; the N-GPU frame time is the slowest band (+ the framebuffer gather)."""
import argparse, os, sys, tempfile, time
import numpy as np
import torch  # before the library: both must share one HIP runtime (torch ships its own libamdhip64)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudagaussianrenderer_amd as gs
from cudagaussianrenderer_amd import scenes

ap = argparse.ArgumentParser()
ap.add_argument("--splats", type=int, default=1_000_000)
ap.add_argument("--worlds", type=str, default="1,2,4,8")
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--cull", type=int, default=0)
ap.add_argument("--inflight", type=int, default=3)
ap.add_argument("--balanced", type=int, default=1)
ap.add_argument("--graph", type=int, default=1)
ap.add_argument("--sort-backend", type=int, default=1, help="0 rocPRIM, 1 native")
ap.add_argument("--ppl", type=int, default=2, help="raster pixels per lane")
ap.add_argument("--pipeline", type=int, default=1)
ap.add_argument("--views", type=int, default=1, help="cameras per chain of kernels (gs_renderer_set_views)")
ap.add_argument("--rebalance", type=int, default=0, help="iterations of re-cutting the bands from their measured times")
ap.add_argument("--gamma", type=float, default=2.0)
ap.add_argument("--storage-order", choices=["caller", "morton"], default="morton", help="gs_renderer_set_storage_order: the caller's order, or along a 3-D Morton curve")
ap.add_argument("--ablate", type=int, default=0, help="gs_debug_set_ablation mask for the timed frames (diagnostic: stages left out)")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--depth-layers", type=int, default=1, help="gs_renderer_set_depth_layers of the bands' renderers (0 / -1: the renderer chooses)")
ap.add_argument("--block-cull", type=int, default=1, help="gs_renderer_debug_set_block_cull: 1 = the band cull tests blocks of 256 Morton neighbours first (round 6), 0 = every band projects all splats (round 5)")
args = ap.parse_args()
W, H = args.width, args.height
gs.lib().gs_set_sort_backend(args.sort_backend)
gs.lib().gs_set_raster_pixels_per_lane(args.ppl)
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "s.ply")
    scenes.write_synthetic_ply(path, args.splats, sh_degree=3, seed=1234)
    ply = gs.parse_ply(path)
cams = [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 8)]
tx, ty = gs.tile_counts(W, H)
base = None
row_cost = np.zeros(ty)
cal = gs.Renderer(W, H)
cal.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
for cam in cams:
    cal.render(cam)
    rng = cal.download(cal.buffers().tileRange, 2 * tx * ty, np.int32).reshape(ty, tx, 2)
    counts = (rng[..., 1] - rng[..., 0]).astype(np.float64)
    row_cost += counts.sum(1) + 6.0 * np.minimum(counts, 256.0).sum(1)
cal.close()
# the same D streams for every band: which hardware queue a new stream lands on depends on how many streams the process has
# created before, and a band must not look slow because its streams happened to share a queue
D = args.inflight
gs.lib().gs_debug_set_ablation(args.ablate)
shared_streams = [torch.cuda.Stream() for _ in range(D)]
kept_blocks = []
def measure(world, bands):
    times = []
    kept_blocks.clear()
    for band in bands:
        streams = shared_streams
        rs = []
        for d in range(D):
            r = gs.Renderer(W, H, tile_rows=band if world > 1 else None, band_cull=args.cull, pipeline=args.pipeline, views=args.views, storage_order=args.storage_order)
            r.set_block_cull(bool(args.block_cull))
            r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
            r.set_stream(streams[d].cuda_stream)
            r.set_graph(bool(args.graph))
            if args.graph:
                r.set_depth_layers(args.depth_layers)
            rs.append(r)
        from concurrent.futures import ThreadPoolExecutor
        from collections import deque
        pools = [ThreadPoolExecutor(1) for _ in range(D)]
        V = args.views
        def run(count, first=0):
            q = deque()
            for c in range(first // V, (first + count) // V):  # chain c renders frames c*V .. c*V + V-1
                if len(q) == D:
                    q.popleft().result()
                if V == 1:
                    q.append(pools[c % D].submit(rs[c % D].render, cams[c % 8]))
                else:
                    q.append(pools[c % D].submit(rs[c % D].render_views, [cams[(c * V + v) % 8] for v in range(V)]))
            while q:
                q.popleft().result()
        run(16)
        torch.cuda.synchronize()
        t = time.perf_counter()
        run(args.frames, 16)
        torch.cuda.synchronize()
        times.append(1e3 * (time.perf_counter() - t) / args.frames)
        kept, total = rs[0].block_cull_kept()
        if world > 1 and kept >= 0 and args.block_cull:
            kept_blocks.append(round(kept / max(total, 1), 3))
        for r in rs:
            r.close()
    return times


for world in [int(x) for x in args.worlds.split(",")]:
    cost = row_cost.copy()
    bands = scenes.balanced_tile_row_bands(cost, world) if args.balanced else scenes.tile_row_bands(ty, world)
    for it in range(args.rebalance + 1):
        print('bands', bands)
        times = measure(world, bands)
        worst = max(times)
        base = base or worst
        print(f"world {world}: per-band ms {[round(x, 3) for x in times]} -> slowest {worst:.3f} ms, speed-up vs 1 GPU {base / worst:.2f}x (gather not included)"
              + (f"; blocks of 256 splats the block test kept, per band: {kept_blocks}" if kept_blocks else ""))
        if world == 1 or it == args.rebalance:
            break
        # re-cut: the rows of a band that took longer than the mean get proportionally more weight (gamma > 1: only part of a band's
        # time follows its rows)
        mean = sum(times) / len(times)
        for (a, b), t in zip(bands, times):
            cost[a:b] *= (t / mean) ** args.gamma
        bands = scenes.balanced_tile_row_bands(cost, world)
