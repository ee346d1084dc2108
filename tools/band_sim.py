#!/usr/bin/env python3
"""Estimate multi-GPU scaling on ONE GPU: render each rank's tile-row band of the C3 scene in turn (no gather) and
report the per-band frame time; the N-GPU frame time is the slowest band (+ the framebuffer gather)."""
import argparse, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudagaussianrenderer_amd as gs
from cudagaussianrenderer_amd import scenes

ap = argparse.ArgumentParser()
ap.add_argument("--splats", type=int, default=1_000_000)
ap.add_argument("--worlds", type=str, default="1,2,4,8")
ap.add_argument("--frames", type=int, default=32)
ap.add_argument("--cull", type=int, default=0)
args = ap.parse_args()
W, H = 1920, 1080
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "s.ply")
    scenes.write_synthetic_ply(path, args.splats, sh_degree=3, seed=1234)
    ply = gs.parse_ply(path)
cams = [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 8)]
tx, ty = gs.tile_counts(W, H)
base = None
for world in [int(x) for x in args.worlds.split(",")]:
    times = []
    for band in scenes.tile_row_bands(ty, world):
        r = gs.Renderer(W, H, tile_rows=band if world > 1 else None, band_cull=args.cull)
        r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
        for i in range(8):
            r.render(cams[i % 8])
        gs.synchronize()
        t = time.perf_counter()
        for i in range(args.frames):
            r.render(cams[i % 8])
        gs.synchronize()
        times.append(1e3 * (time.perf_counter() - t) / args.frames)
        r.close()
    worst = max(times)
    base = base or worst
    print(f"world {world}: per-band ms {[round(x, 3) for x in times]} -> slowest {worst:.3f} ms, speed-up vs 1 GPU {base / worst:.2f}x (gather not included)")
