#!/usr/bin/env python3
"""Writes the bench scene (.ply) and the 8-view orbit (cams.bin) for tools/frame_pump (native frame loop), builds it, and runs it for
the full frame and for the bands of an 8-way split."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cudagaussianrenderer_amd as gs
from cudagaussianrenderer_amd import scenes
W, H, N = 1920, 1080, 1_000_000
out = os.environ.get("PUMP_DIR", "/tmp/gs_pump")  # (not under gpurun_out/: the scene is 250 MB)
os.makedirs(out, exist_ok=True)
ply_path = os.path.join(out, "scene.ply")
scenes.write_synthetic_ply(ply_path, N, sh_degree=3, seed=1234)
ply = gs.parse_ply(ply_path)
with open(os.path.join(out, "cams.bin"), "wb") as f:
    for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 8):
        f.write(bytes(gs.camera_from_pose(p, q, W, H)))
exe = os.path.join(ROOT, "tools", "frame_pump")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "frame_pump.cpp"), "-o", exe,
                       "-L" + os.path.join(ROOT, "cudagaussianrenderer_amd"), "-lgsplat_hip", "-Wl,-rpath," + os.path.join(ROOT, "cudagaussianrenderer_amd"), "-lpthread"])
env = dict(os.environ)
for extra in sys.argv[1:]:
    k, v = extra.split("=")
    env[k] = v
masks = [int(v) for v in os.environ.get("PUMP_MASKS", "0").split(",")]
depths = [int(v) for v in os.environ.get("PUMP_DEPTHS", "1,2,3,4,6").split(",")]
for rows in [(0, 0), (29, 34), (0, 17)]:
    for mask in masks:
        for d in depths:
            try:
                print(subprocess.run([exe, ply_path, os.path.join(out, "cams.bin"), str(W), str(H), str(rows[0]), str(rows[1]), str(d), "512", "1", str(mask)],
                                     env=env, stdout=subprocess.PIPE, text=True, timeout=60).stdout.strip(), flush=True)
            except subprocess.TimeoutExpired:
                print("timeout", rows, mask, d, flush=True)
                sys.exit(1)  # no further GPU work after a hang
if not os.environ.get("PUMP_KEEP"):
    os.remove(ply_path)
