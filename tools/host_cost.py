#!/usr/bin/env python3
"""Host-side cost per frame: one host thread drives D renderers round-robin (captured frames return right after the
graph launch, so no thread pool is needed). Prints ms/frame for a band of the C3 scene."""
import argparse, os, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudagaussianrenderer_amd as gs
from cudagaussianrenderer_amd import scenes

ap = argparse.ArgumentParser()
ap.add_argument("--splats", type=int, default=1_000_000)
ap.add_argument("--band", type=str, default="29,34")
ap.add_argument("--frames", type=int, default=512)
ap.add_argument("--depths", type=str, default="1,2,3,4")
ap.add_argument("--graphs", type=str, default="1,0")
args = ap.parse_args()
W, H = 1920, 1080
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "s.ply")
    scenes.write_synthetic_ply(path, args.splats, sh_degree=3, seed=1234)
    ply = gs.parse_ply(path)
cams = [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 8)]
band = tuple(int(x) for x in args.band.split(",")) if args.band != "full" else None
for graph in [int(x) for x in args.graphs.split(',')]:
    for D in [int(x) for x in args.depths.split(',')]:
        rs = []
        for d in range(D):
            r = gs.Renderer(W, H, tile_rows=band)
            r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
            r.set_stream(torch.cuda.Stream().cuda_stream)
            r.set_graph(bool(graph))
            rs.append(r)
        for i in range(16 * D):
            rs[i % D].render(cams[i % 8])
        torch.cuda.synchronize()
        t = time.perf_counter()
        host = 0.0
        for i in range(args.frames):
            t0 = time.perf_counter()
            rs[i % D].render(cams[i % 8])
            host += time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(f"graph={graph} D={D}: {1e3 * dt / args.frames:.3f} ms/frame, time inside render() {1e3 * host / args.frames:.3f} ms")
        for r in rs:
            r.close()
