#!/usr/bin/env python3
"""Round 6: what can a lone layered frame gain from fusing the live-splat compaction into the count pass's prologue?
Upper bound by construction: the SAME camera rendered again and again, the later layers walking the live lists the previous frame left
(gs_renderer_debug_set_layer_compaction_reuse 2: the three compaction kernels are simply not launched -- a fusion that costs nothing).
Frames one at a time (render, synchronise): one list | N layers | N layers, lists kept (sanity) | N layers, compaction for free."""
import argparse, os, sys, tempfile, time
import numpy as np
import torch  # before the library: both must share one HIP runtime
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudagaussianrenderer_amd as gs
from cudagaussianrenderer_amd import scenes

ap = argparse.ArgumentParser()
ap.add_argument("--splats", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--layers", type=int, default=2)
ap.add_argument("--frames", type=int, default=300)
args = ap.parse_args()
W, H = args.width, args.height
lib = gs.lib()
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "s.ply")
    scenes.write_synthetic_ply(path, args.splats, sh_degree=3, seed=1234)
    ply = gs.parse_ply(path)
cams = [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 8)]


stream = torch.cuda.Stream()


def lone_ms(r, cam, frames):
    for _ in range(16):
        r.render(cam)
    stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        r.render(cam)
        stream.synchronize()  # the reference's cudaDeviceSynchronize after every frame (Demo.cpp:478)
    return 1e3 * (time.perf_counter() - t0) / frames


r = gs.Renderer(W, H, storage_order="morton")
r.set_stream(stream.cuda_stream)
r.set_graph(True)
r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
rows = []
for v in (0, 3, 5):
    cam = cams[v]
    r.set_depth_layers(1)
    lib.gs_renderer_debug_set_layer_compaction_reuse(r.handle, 0)
    one = lone_ms(r, cam, args.frames)
    pairs = r.listed_pairs()
    want = r.read_image()
    r.set_depth_layers(args.layers)
    layered = lone_ms(r, cam, args.frames)
    assert np.array_equal(r.read_image(), want)
    listed = r.listed_pairs()
    lib.gs_renderer_debug_set_layer_compaction_reuse(r.handle, 1)
    kept = lone_ms(r, cam, args.frames)
    assert np.array_equal(r.read_image(), want)
    lib.gs_renderer_debug_set_layer_compaction_reuse(r.handle, 2)
    free = lone_ms(r, cam, args.frames)
    assert np.array_equal(r.read_image(), want) and r.listed_pairs() == listed
    lib.gs_renderer_debug_set_layer_compaction_reuse(r.handle, 0)
    rows.append((one, layered, kept, free))
    print(f"view {v}: {pairs} pairs ({listed} listed in {args.layers} layers); lone frame ms: one list {one:.4f} | {args.layers} layers {layered:.4f} ({100 * (layered / one - 1):+.1f} %) | "
          f"lists kept {kept:.4f} | compaction for free {free:.4f} ({100 * (free / one - 1):+.1f} % against one list, {100 * (free / layered - 1):+.1f} % against the layered frame)", flush=True)
m = np.array(rows).mean(0)
print(f"mean: one list {m[0]:.4f} ms | {args.layers} layers {m[1]:.4f} ({100 * (m[1] / m[0] - 1):+.1f} %) | compaction for free {m[3]:.4f} ({100 * (m[3] / m[0] - 1):+.1f} % against one list; "
      f"the three launches per later layer cost the lone frame {1e3 * (m[1] - m[3]):.1f} us)")
r.close()
