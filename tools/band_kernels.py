#!/usr/bin/env python3
"""One tile-row band of a multi-GPU split rendered as a rank renders it (captured chains, band cull), chains one at a time -- to be run
under `rocprofv3 --kernel-trace --stats` (tools/r06_band_kernels.sh): which kernels a band chain is made of and what each costs.
usage: band_kernels.py --splats N --width W --height H --rows A B [--views V] [--block-cull 0|1] [--depth-layers L] [--chains C]"""
import argparse, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudagaussianrenderer_amd as gs
from cudagaussianrenderer_amd import scenes

ap = argparse.ArgumentParser()
ap.add_argument("--splats", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--rows", type=int, nargs=2, default=[29, 34])
ap.add_argument("--views", type=int, default=4)
ap.add_argument("--block-cull", type=int, default=1)
ap.add_argument("--depth-layers", type=int, default=1)
ap.add_argument("--chains", type=int, default=64)
args = ap.parse_args()
W, H = args.width, args.height
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "s.ply")
    scenes.write_synthetic_ply(path, args.splats, sh_degree=3, seed=1234)
    ply = gs.parse_ply(path)
cams = [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 8)]
r = gs.Renderer(W, H, tile_rows=tuple(args.rows), views=args.views, storage_order="morton")
r.set_block_cull(bool(args.block_cull))
r.set_graph(True)
r.set_depth_layers(args.depth_layers)
r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
V = args.views
for c in range(args.chains):
    group = [cams[(c * V + v) % 8] for v in range(V)]
    if V == 1:
        r.render(group[0])
    else:
        r.render_views(group)
    gs.synchronize()
    r.read_image()  # (waits for the renderer's stream)
kept, total = r.block_cull_kept()
print(f"band {args.rows} of {args.splats} splats at {W}x{H}, {V} view(s) per chain, block cull {args.block_cull}: kept splats {r.band_kept_splats()} of {args.splats * V}, "
      f"blocks kept {kept} of {total}, depth layers in use {r.depth_layers_in_use()}")
r.close()
