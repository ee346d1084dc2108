#!/usr/bin/env python3
"""How many (splat, tile) pairs would a two-layer frame handle?  Layer 1 = the nearest fraction a of the depth-sorted splats;
layer 2 = the rest, expanded only into the tiles that layer 1 did not saturate.  Uses a real frame of the bench scene
(lists from the renderer, consumed entries per tile from the raster trace).  Diagnostics for DESIGN.md section 9."""
import argparse, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--opacity-mean", type=float, default=1.0)
ap.add_argument("--view", type=int, default=0)
args = ap.parse_args()
import torch  # noqa
import cudagaussianrenderer_amd as gs
from cudagaussianrenderer_amd import scenes
lib = gs.lib(); lib.gs_set_device(0)
W, H, N = 1920, 1080, 1_000_000
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "s.ply")
    names, table = scenes.synthetic_vertex_table(N, sh_degree=3, seed=1234)
    table[:, names.index("opacity")] += np.float32(args.opacity_mean - 1.0)
    scenes.write_ply(path, names, table)
    ply = gs.parse_ply(path)
cams = [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 8)]
tx, ty = gs.tile_counts(W, H); tiles = tx * ty
r = gs.Renderer(W, H)
r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
trace = gs.DeviceBuffer(nbytes=32 * tiles, dtype=np.uint64)
trace.clear_memory(0)
lib.gs_debug_set_raster_trace(trace.ptr)
st = r.render(cams[args.view], stats=True)
lib.gs_debug_set_raster_trace(None)
t = trace.copy_to(dtype=np.uint64).reshape(tiles, 4)
consumed = (t[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
b = r.buffers()
P = b.tileListSize
values = r.download(b.values, P, np.int32)
rng = r.download(b.tileRange, 2 * tiles, np.int32).reshape(tiles, 2)
z = r.download(b.positionClipSpaceZ, N, np.float32)
length = (rng[:, 1] - rng[:, 0]).astype(np.int64)
visible = z > -100
order = np.argsort(np.where(visible, z, np.inf), kind="stable")
rank = np.empty(N, np.int64); rank[order] = np.arange(N)
nvis = int(visible.sum())
pair_rank = rank[values]
tile_of_pair = np.repeat(np.arange(tiles), length)
print(f"P = {P}, P_c = {consumed.sum()}, visible splats {nvis}, tiles with work {(length > 0).sum()}")
for a in (0.1, 0.15, 0.2, 0.25, 0.35, 0.5):
    n1 = int(a * nvis)
    near = pair_rank < n1
    near_len = np.bincount(tile_of_pair[near], minlength=tiles)
    flagged = consumed > near_len  # the tile went past its near entries (or has none): not saturated by layer 1
    flagged |= (consumed == near_len) & (length > near_len) & (consumed % 32 != 0)
    p1 = int(near.sum()); p2 = int((length - near_len)[flagged].sum())
    print(f"a = {a:4.2f}: layer 1 {p1:8d} pairs, layer 2 {p2:8d} pairs in {int(flagged.sum()):5d} tiles -> {p1 + p2:8d} = {100 * (p1 + p2) / P:5.1f} % of P")
