#!/usr/bin/env python3
"""Work distribution of the tile-list builder for one C3 view: AABB units per group of 256 depth-sorted splats."""
import os, sys, tempfile
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudagaussianrenderer_amd as gs
from cudagaussianrenderer_amd import scenes

n, W, H = 1_000_000, 1920, 1080
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "s.ply")
    scenes.write_synthetic_ply(path, n, sh_degree=3, seed=1234)
    ply = gs.parse_ply(path)
cams = [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 8)]
r = gs.Renderer(W, H)
r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], 3, 48)
r.render(cams[0])
b = r.buffers()
ell = r.download(b.screenEllipse, 4 * n, np.float32).reshape(n, 4).astype(np.float64)
xy = r.download(b.positionClipSpaceXY, 2 * n, np.float32).reshape(n, 2).astype(np.float64)
z = r.download(b.positionClipSpaceZ, n, np.float32)
key = np.minimum(np.floor(np.clip((z.astype(np.float64) + 1) * 0.5, 0, 1).astype(np.float32).astype(np.float64) * 2.0 ** 32), 2.0 ** 32 - 1)
order = np.argsort(key, kind="stable")
c, s_, ex, ey = ell[:, 0], ell[:, 1], ell[:, 2], ell[:, 3]
hx = np.abs(c * ex) + np.abs(s_ * ey)
hy = np.abs(s_ * ex) + np.abs(c * ey)
tx, ty = gs.tile_counts(W, H)
x0 = np.clip(np.floor((xy[:, 0] - hx + 1) * 0.5 * W / 16), 0, tx); x1 = np.clip(np.ceil((xy[:, 0] + hx + 1) * 0.5 * W / 16), 0, tx)
y0 = np.clip(np.floor((xy[:, 1] - hy + 1) * 0.5 * H / 16), 0, ty); y1 = np.clip(np.ceil((xy[:, 1] + hy + 1) * 0.5 * H / 16), 0, ty)
area = np.maximum(x1 - x0, 0) * np.maximum(y1 - y0, 0)
area[(ex == 0) & (ey == 0)] = 0
units = np.ceil(area / 4)[order]
g = units[: n // 256 * 256].reshape(-1, 256).sum(1)
rounds = np.ceil(g / 256)
vis = g > 0
print("groups", len(g), "with work", int(vis.sum()), "total units", int(g.sum()), "total rounds", int(rounds.sum()))
print("rounds per working group: percentiles", {p: int(np.percentile(rounds[vis], p)) for p in (10, 50, 90, 99, 100)})
print("first working groups (nearest splats):", rounds[vis][:12].astype(int).tolist())
print("largest AABB areas:", np.sort(area)[-5:].astype(int).tolist(), "mean area of visible", float(area[area > 0].mean()))
