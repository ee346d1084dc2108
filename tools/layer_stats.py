#!/usr/bin/env python3
"""usage (GPU box): GSPLAT_HIP_LIBRARY=cudagaussianrenderer_amd/libgsplat_hip_audit.so python tools/layer_stats.py [splats] [w] [h] [layers]
What the later layers' count passes of a depth-layered frame do with their splats (diagnostics build: gs_debug_layer_stats)."""
import ctypes as C, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cudagaussianrenderer_amd as gs
from cudagaussianrenderer_amd import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
h = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
layers = int(sys.argv[4]) if len(sys.argv) > 4 else 3
lib = gs.lib()
lib.gs_debug_layer_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.gs_debug_layer_stats.restype = None
path = os.path.join(tempfile.mkdtemp(), "s.ply")
names, table = scenes.synthetic_vertex_table(n, sh_degree=3, seed=0)
scenes.write_ply(path, names, table)
ply = gs.parse_ply(path)
cams = [gs.camera_from_pose(p, q, w, h) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 8)]
r = gs.Renderer(w, h, storage_order="morton")
r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
r.set_graph(True)
r.set_depth_layers(layers)
lib.gs_renderer_debug_set_layer_compaction(r.handle, 0)  # (the counters sit in the count pass's own test: every splat of the later layers must reach it)
out = (C.c_ulonglong * 8)()
for v, cam in enumerate(cams[:3]):
    r.render(cam)
    lib.gs_debug_layer_stats(out, 1)
    r.render(cam)
    lib.gs_debug_layer_stats(out, 1)
    s = list(out)
    print(f"view {v}: listed {r.listed_pairs()} | later layers: splats on screen {s[0]}, box fully stopped {s[1]} ({s[1]/max(1,s[0]):.3f}), too large to test {s[2]} "
          f"({s[3]} units), tested and kept {s[0]-s[1]-s[2]} ({s[4]} units; {s[5]} after shrinking)")
