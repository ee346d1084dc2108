"""Diagnostics (GPU box): how often the tile-list builder's guard band is entered on the bench scene.  Needs a library built with
-DGS_AB_GUARD_COUNT (GSPLAT_HIP_LIBRARY=ab_variants/lib_guard_count.so)."""
import ctypes as C
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import cudagaussianrenderer_amd as gs  # noqa: E402
from cudagaussianrenderer_amd import scenes  # noqa: E402

n, w, h = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 1920, 1080
path = os.path.join(tempfile.mkdtemp(), "scene.ply")
scenes.write_synthetic_ply(path, n, sh_degree=0, seed=1234, log_scale_mean=-3.6)
ply = gs.parse_ply(path)
lib = gs.lib()
fn = C.CDLL(os.environ["GSPLAT_HIP_LIBRARY"]).gs_debug_guard_stats
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
r = gs.Renderer(w, h)
r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"])
cams = [gs.camera_from_pose(p, q, w, h) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 4)]
out = (C.c_ulonglong * 4)()
fn(out, 1)
for cam in cams:
    st = r.render(cam, stats=True)
    fn(out, 1)
    units, entered, exact, changed = [int(v) for v in out]
    print(f"pairs {st.tileListSize}: units {units}, entered the band {entered} ({100.0 * entered / max(units, 1):.4f} %, "
          f"{100.0 * (1 - (1 - entered / max(units, 1)) ** 64):.2f} % of wave rounds), tiles decided exactly {exact}, decisions changed {changed}")
