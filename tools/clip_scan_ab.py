#!/usr/bin/env python3
"""A/B on one box, alternating in one process: gs_debug_set_clip_scan_fusion off / on for the host that renders ONE frame at a time (the
reference's loop: a frame, a synchronise, the next camera), on the bench scene, with and without the tile start order.  With the switch
on the clip kernel rides in the launch of the depth sort's first column scan, the tile start order's builder with it; a library built
with -DGS_CLIP_SCAN_PARTS=2 or 3 deals the clip kernel's workgroups to two or three of the sort's scans (profiles/r06_clip_scan_slices.txt).
usage (GPU box): python tools/clip_scan_ab.py [frames=1500] [rounds=4]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402  (before the library)

import cudagaussianrenderer_amd as gs  # noqa: E402
from cudagaussianrenderer_amd import scenes  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    W, H = 1920, 1080
    lib = gs.lib()
    lib.gs_set_device(0)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "s.ply")
        scenes.write_synthetic_ply(path, 1_000_000, sh_degree=3, seed=1234)
        ply = gs.parse_ply(path)
    cams = [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 360)]
    stream = torch.cuda.Stream(0)
    r = gs.Renderer(W, H, storage_order="morton")
    r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
    r.set_stream(stream.cuda_stream)
    r.set_graph(True)
    results = {}
    want = None
    for rnd in range(rounds):
        for order in (0, 1):
            r.set_tile_start_order(gs.TILE_START_ORDER_PREVIOUS_FRAME if order else gs.TILE_START_ORDER_SCREEN)
            for fusion in (0, 1):
                lib.gs_debug_set_clip_scan_fusion(fusion)
                for k in range(60):
                    r.render(cams[k % 360])
                stream.synchronize()
                t = time.perf_counter()
                for k in range(frames):
                    r.render(cams[k % 360])
                    stream.synchronize()
                dt = (time.perf_counter() - t) / frames
                results.setdefault((order, fusion), []).append(dt * 1e3)
                r.render(cams[7])
                image = r.read_image().copy()
                if want is None:
                    want = image
                assert np.array_equal(image, want)
    lib.gs_debug_set_clip_scan_fusion(0)
    for (order, fusion), ms in sorted(results.items()):
        print(f"tile start order {'previous frame' if order else 'screen'}, clip kernel {'riding in the three scans' if fusion else 'by itself'}: "
              f"ms per frame {[round(v, 4) for v in ms]} -> mean {np.mean(ms):.4f} = {1e3 / np.mean(ms):.0f} frames/s")


if __name__ == "__main__":
    main()
