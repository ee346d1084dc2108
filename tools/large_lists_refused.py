#!/usr/bin/env python3
"""A frame that needs more list entries than the int32 interface can express (gs_renderer_render's -2): the plain path, the captured path and
the reference stage order must all refuse it -- no fault, no truncated image -- and the renderer must render a smaller frame afterwards.
usage (GPU box): python tools/large_lists_refused.py [splats=5000000] [log_scale_mean=-1.4]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402  (before the library)

import cudagaussianrenderer_amd as gs  # noqa: E402
from cudagaussianrenderer_amd import scenes  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    log_scale = float(sys.argv[2]) if len(sys.argv) > 2 else -1.4
    W, H = 3840, 2160
    gs.lib().gs_set_device(0)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "s.ply")
        scenes.write_synthetic_ply(path, n, sh_degree=1, seed=1234, log_scale_mean=log_scale)
        ply = gs.parse_ply(path)
    cam = gs.camera_from_bounds(ply["boundsMin"], ply["boundsMax"], W, H)
    small = gs.camera_from_bounds(ply["boundsMin"], ply["boundsMax"], 640, 360)
    images = []
    for what, pipeline, graph in [("reference stage order", gs.PIPELINE_REFERENCE, False), ("two-level, plain", gs.PIPELINE_TWO_LEVEL, False),
                                  ("two-level, captured", gs.PIPELINE_TWO_LEVEL, True)]:
        r = gs.Renderer(W, H)
        r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
        r.set_pipeline(pipeline)
        r.set_graph(graph)
        try:
            r.render(cam)
            print(f"{what}: rendered?!")
            sys.exit(1)
        except RuntimeError as e:
            print(f"{what}: {e}", flush=True)
        r.close()
        torch.cuda.synchronize(0)
        # the same scene at a resolution whose list fits: the library is still in working order
        r = gs.Renderer(640, 360)
        r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
        r.set_pipeline(pipeline)
        r.set_graph(graph)
        r.render(small)
        st = r.render(small, stats=True)
        images.append(r.read_image().copy())
        print(f"   640x360 afterwards: {int(st.tileListSize)} pairs", flush=True)
        r.close()
    print("small frames equal:", [bool(np.array_equal(images[0], im)) for im in images[1:]])


if __name__ == "__main__":
    main()
