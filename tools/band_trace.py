#!/usr/bin/env python3
"""Render one band repeatedly (no pipelining) -- for rocprofv3 timelines of a single multi-GPU rank's frame."""
import os, sys, tempfile
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudagaussianrenderer_amd as gs
from cudagaussianrenderer_amd import scenes
a, b = int(sys.argv[1]), int(sys.argv[2])
W, H = 1920, 1080
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "s.ply")
    scenes.write_synthetic_ply(path, 1_000_000, sh_degree=3, seed=1234)
    ply = gs.parse_ply(path)
cams = [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 8)]
r = gs.Renderer(W, H, tile_rows=(a, b))
r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
for i in range(24):
    st = r.render(cams[i % 8], stats=True)
print("P", st.tileListSize, "frame ms", st.msFrame)
r.close()
