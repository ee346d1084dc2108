#!/usr/bin/env python3
"""gs_renderer_set_tile_start_order on other scenes than the headline one: the host that renders ONE frame at a time (a frame, a
synchronise, the next camera of an orbit in 1-degree steps), screen order and previous-frame order alternating in one process on one box.
Scenes: the headline scene; the same with low opacities (raster-heavy: tiles stop late); the Epanechnikov footprint; 5 M splats at
3840x2160 (the renderer chooses its depth layers for a lone frame: the start order then applies to every layer's raster launch).
usage (GPU box): python tools/tile_order_scenes.py [frames=600] [rounds=3]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402  (before the library)

import cudagaussianrenderer_amd as gs  # noqa: E402
from cudagaussianrenderer_amd import scenes  # noqa: E402


def scene(n, opacity_shift=0.0, **kw):
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "s.ply")
        names, table = scenes.synthetic_vertex_table(n, sh_degree=3, seed=1234, **kw)
        if opacity_shift:
            table[:, names.index("opacity")] += np.float32(opacity_shift)
        scenes.write_ply(path, names, table)
        return gs.parse_ply(path)


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    gs.lib().gs_set_device(0)
    stream = torch.cuda.Stream(0)
    for what, n, w, h, shift, footprint in [("headline scene (1 M splats, SH 3, 1920x1080)", 1_000_000, 1920, 1080, 0.0, gs.FOOTPRINT_GAUSSIAN),
                                            ("low opacities (raw opacity mean -3: raster-heavy)", 1_000_000, 1920, 1080, -4.0, gs.FOOTPRINT_GAUSSIAN),
                                            ("Epanechnikov footprint", 1_000_000, 1920, 1080, 0.0, gs.FOOTPRINT_EPANECHNIKOV),
                                            ("5 M splats at 3840x2160 (BASELINE.json configs[4] on one GPU)", 5_000_000, 3840, 2160, 0.0, gs.FOOTPRINT_GAUSSIAN)]:
        ply = scene(n, shift)
        cams = [gs.camera_from_pose(p, q, w, h) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 360)]
        r = gs.Renderer(w, h, storage_order="morton")
        r.set_footprint(footprint)
        r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
        r.set_stream(stream.cuda_stream)
        r.set_graph(True)
        r.set_depth_layers(gs.DEPTH_LAYERS_CHOOSE)  # (a lone frame's choice)
        ms = {0: [], 1: []}
        want = None
        for _ in range(rounds):
            for order in (0, 1):
                r.set_tile_start_order(gs.TILE_START_ORDER_PREVIOUS_FRAME if order else gs.TILE_START_ORDER_SCREEN)
                for k in range(40):
                    r.render(cams[k % 360])
                stream.synchronize()
                t = time.perf_counter()
                for k in range(frames):
                    r.render(cams[k % 360])
                    stream.synchronize()
                ms[order].append((time.perf_counter() - t) / frames * 1e3)
                r.render(cams[3])
                image = r.read_image().copy()
                if want is None:
                    want = image
                assert np.array_equal(image, want)
        a, b = float(np.mean(ms[0])), float(np.mean(ms[1]))
        print(f"{what}: depth layers in use {r.depth_layers_in_use()}; screen order {[round(v, 4) for v in ms[0]]} ms, previous-frame order {[round(v, 4) for v in ms[1]]} ms "
              f"-> {a:.4f} -> {b:.4f} ms per frame ({(b / a - 1) * 100:+.1f} %; {1e3 / a:.0f} -> {1e3 / b:.0f} frames/s), same bytes", flush=True)
        r.close()


if __name__ == "__main__":
    main()
