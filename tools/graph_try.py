import sys; sys.path.insert(0, '.')
import numpy as np
import cudagaussianrenderer_amd as gs
pos, sr, col = gs.generate_random_gaussians(20000, seed=5, min_scale=0.01, max_scale=0.2)
cam = gs.camera_from_bounds([-4]*3, [4]*3, 640, 360)
a = gs.Renderer(640, 360); a.set_scene(pos, sr, col); a.render(cam); want = a.read_image()
r = gs.Renderer(640, 360); r.set_graph(True); r.set_scene(pos, sr, col)
for i in range(3):
    r.render(cam); print('frame', i, 'ok', flush=True)
got = r.read_image(); print('equal', np.array_equal(got, want), r.buffers().tileListSize, a.buffers().tileListSize)
