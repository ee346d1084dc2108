#!/usr/bin/env python3
"""CPU diagnostics (oracle): list entries consumed per tile by the raster, and list length per tile, for one C3 view."""
import os, sys, tempfile, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GSPLAT_NO_GPU", "1")
from oracle import gsref
from cudagaussianrenderer_amd import scenes
import cudagaussianrenderer_amd as gs

gsref.build()
n, W, H = 1_000_000, 1920, 1080
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "s.ply")
    scenes.write_synthetic_ply(path, n, sh_degree=3, seed=1234)
    ply = gs.parse_ply(path)
cams = [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 8)]
tx, ty = gs.tile_counts(W, H)
buf = np.zeros(tx * ty, np.int32)
gsref.lib().gsref_set_consumed_per_tile_buffer(buf.ctypes.data_as(C.POINTER(C.c_int32)))
sh = gs.realign_spherical_harmonics(ply["sh"], ply["shCount"], n)
img, st, _ = gsref.render(ply["position"], ply["scaleAndRotation"], ply["color"], sh, 3, 48, cams[0], W, H)
gsref.lib().gsref_set_consumed_per_tile_buffer(None)
c = buf.reshape(ty, tx)
print("pairs", st.tileListSize, "consumed", st.consumedPairs, "sum", int(c.sum()))
print("percentiles of consumed entries per tile:", {p: int(np.percentile(buf, p)) for p in (50, 90, 99, 99.9, 100)})
print("tiles by consumed:", {k: int((buf == k).sum()) for k in (0, 32, 64, 96, 128)}, "more", int((buf > 128).sum()))
print("row sums (consumed) per tile row:", c.sum(1).tolist())
# distinct splats among the consumed entries (how many SH evaluations the frame really needs)
xy, zz, ell, con = gsref.evaluate_splat_clip_data(ply["position"], ply["scaleAndRotation"], cams[0], W, H)
k, v, total = gsref.build_tile_list(xy, zz, ell, W, H)
k, v = gsref.sort_tile_list(k, v, gsref.sort_end_bit(W, H))
rng = gsref.evaluate_tile_ranges(k, tx * ty).reshape(-1, 2)
used = np.zeros(n, bool)
for t in np.nonzero(buf)[0]:
    used[v[rng[t, 0]: rng[t, 0] + buf[t]]] = True
print("splats in some list:", int(np.unique(v).size), "consumed entries:", int(buf.sum()), "distinct consumed splats:", int(used.sum()))
