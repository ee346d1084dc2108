import os, sys, tempfile
sys.path.insert(0, '.')
import numpy as np
import torch
import cudagaussianrenderer_amd as gs
from cudagaussianrenderer_amd import scenes
n = int(sys.argv[1]); band = (int(sys.argv[2]), int(sys.argv[3])); frames = int(sys.argv[4])
W, H = 1920, 1080
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "s.ply")
    scenes.write_synthetic_ply(path, n, sh_degree=3, seed=1234)
    ply = gs.parse_ply(path)
cams = [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 8)]
a = gs.Renderer(W, H, tile_rows=band); a.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], 3, 48)
r = gs.Renderer(W, H, tile_rows=band); r.set_graph(True); r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], 3, 48)
for i in range(frames):
    a.render(cams[i % 8]); r.render(cams[i % 8])
    ok = np.array_equal(a.read_image(), r.read_image())
    print('frame', i, 'equal', ok, r.buffers().tileListSize, a.buffers().tileListSize, flush=True)
ba, bb = a.buffers(), r.buffers()
va = a.download(ba.values, ba.tileListSize, np.int32); vb = r.download(bb.values, bb.tileListSize, np.int32)
ra = a.download(ba.tileRange, 2*120*68, np.int32); rb = r.download(bb.tileRange, 2*120*68, np.int32)
print('values equal', np.array_equal(va, vb), 'ranges equal', np.array_equal(ra, rb), 'max value', vb.max(), 'range max', rb.max(), rb.min())
if not np.array_equal(ra, rb):
    d = np.nonzero(ra != rb)[0]; print('first range diffs', d[:10], ra[d[:10]], rb[d[:10]])
if not np.array_equal(va, vb):
    d = np.nonzero(va != vb)[0]; print('value diffs', len(d), d[:10], va[d[:10]], vb[d[:10]])
