#!/usr/bin/env python3
"""Frames whose pair lists approach the 32-bit limits of the path (several hundred million pairs in ONE list), held to the bytes of
the reference's stage order (gs_renderer_set_pipeline(GS_PIPELINE_REFERENCE): staged builder + rocPRIM's 64-bit sort of all pairs --
an independent list path) and to the layered frames' bytes.  The CPU oracle is not practical at this size (its sort alone takes minutes);
what this checks is that no index, offset or grid size of the two-level path wraps when the list is a hundred times the bench's.
usage (GPU box): python tools/large_lists.py [splats=5000000] [log_scale_mean=-2.4] [width=3840] [height=2160] [sh_degree=1] [MIN:MAX]
With MIN:MAX the scene is `splats` procedural Gaussians with scales in that range (SH 0) -- for scenes of 10^8 splats, which as a .ply file would
not fit the box's /tmp."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402  (before the library: DESIGN.md section 8, load order)

import cudagaussianrenderer_amd as gs  # noqa: E402
from cudagaussianrenderer_amd import scenes  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    log_scale = float(sys.argv[2]) if len(sys.argv) > 2 else -2.4
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 3840
    H = int(sys.argv[4]) if len(sys.argv) > 4 else 2160
    degree = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    lib = gs.lib()
    lib.gs_set_device(0)
    if len(sys.argv) > 6:
        # many SMALL splats instead (the reference's procedural generator, Demo.cpp:119-143): "MIN_SCALE:MAX_SCALE"
        lo, hi = [float(v) for v in sys.argv[6].split(":")]
        pos, sr, col = gs.generate_random_gaussians(n, seed=17, min_scale=lo, max_scale=hi)
        ply = {"position": pos, "scaleAndRotation": sr, "color": col, "sh": None, "shDegree": 0, "shCount": 0, "boundsMin": [-4.0] * 3, "boundsMax": [4.0] * 3}
    else:
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "s.ply")
            scenes.write_synthetic_ply(path, n, sh_degree=degree, seed=1234, log_scale_mean=log_scale)
            ply = gs.parse_ply(path)
    cams = [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 8)][:2]
    stream = torch.cuda.Stream(0)

    def renderer(storage, pipeline, graph, layers):
        r = gs.Renderer(W, H, storage_order=storage)
        r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
        r.set_stream(stream.cuda_stream)
        r.set_pipeline(pipeline)
        r.set_graph(graph)
        if layers != 1:
            r.set_depth_layers(layers)
        return r

    def frames(r, what, graph):
        out = []
        for cam in cams:
            # (the first frame of a camera may grow the lists, the renderer's choice of layers follows the first frame's total: the third
            # frame is the steady one.  No statistics are asked for: a frame with statistics is a plain one, never a captured or layered one)
            r.render(cam)
            r.render(cam)
            stream.synchronize()
            t = time.perf_counter()
            r.render(cam)
            stream.synchronize()
            dt = time.perf_counter() - t
            out.append(r.read_image().copy())
            listed = f", depth layers in use {r.depth_layers_in_use()}, pairs listed (all layers) {r.listed_pairs()}" if graph else ""
            print(f"{what}: {dt * 1e3:.2f} ms{listed}", flush=True)
        pairs = r.render(cams[-1], stats=True)  # (a plain frame, for the scene's pair total)
        print(f"   pairs of the last camera's frame: {int(pairs.tileListSize)} (list capacity {int(pairs.tileListCapacity)})", flush=True)
        return out

    want = None
    for (what, storage, pipeline, graph, layers) in [
        ("reference stage order (rocPRIM 64-bit sort of all pairs), caller's storage order", "caller", gs.PIPELINE_REFERENCE, False, 1),
        ("two-level, one list, plain launches, caller's storage order", "caller", gs.PIPELINE_TWO_LEVEL, False, 1),
        ("two-level, one list, captured, Morton storage", "morton", gs.PIPELINE_TWO_LEVEL, True, 1),
        ("two-level, the renderer's choice of depth layers, captured, Morton storage", "morton", gs.PIPELINE_TWO_LEVEL, True, 0),
    ]:
        r = renderer(storage, pipeline, graph, layers)
        got = frames(r, what, graph)
        r.close()
        torch.cuda.synchronize(0)
        torch.cuda.empty_cache()
        if want is None:
            want = got
        else:
            same = [bool(np.array_equal(a, b)) for a, b in zip(want, got)]
            print(f"   images equal to the reference stage order's, byte for byte: {same}", flush=True)
            if not all(same):
                diff = [int(np.abs(a.astype(int) - b.astype(int)).max()) for a, b in zip(want, got)]
                print(f"   LARGEST DIFFERENCE {diff}")
                sys.exit(1)
    print("ok")


if __name__ == "__main__":
    main()
