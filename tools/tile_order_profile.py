#!/usr/bin/env python3
"""1-degree orbit steps on one renderer, plain launches, frame then synchronise, with gs_renderer_set_tile_start_order MODE (argv[1]: 0 / 1):
run under rocprofv3 --kernel-trace --stats to see what the option does to the clip kernel (whose first eight workgroups build the table) and
to the raster kernel."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudagaussianrenderer_amd as gs
from cudagaussianrenderer_amd import scenes
mode = int(sys.argv[1])
graph = len(sys.argv) > 2 and sys.argv[2] == "graph"
W, H = 1920, 1080
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "s.ply")
    scenes.write_synthetic_ply(path, 1_000_000, sh_degree=3, seed=1234)
    ply = gs.parse_ply(path)
fine = [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 360)]
cull = 1 if (len(sys.argv) > 3 and sys.argv[3] == "cull") else None  # (band cull on the full frame: the builder runs as a kernel of its own)
r = gs.Renderer(W, H, storage_order="morton", band_cull=cull)
r.set_graph(graph)
r.set_tile_start_order(mode)
r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
for i in range(200):
    r.render(fine[i % 360])
    r.captured_retries() if False else None
    gs.synchronize()
    r.band_kept_splats() if False else None
print("state", r.tile_order_state())
r.close()
