#!/usr/bin/env python3
"""Soak test of the pipelined product path: D renderers (one scene, shared), D host threads, captured frames back to back for S
seconds; every CHECK-th frame of every renderer is read back and compared byte for byte with the first image of its view; free
device memory before and after.  The first images are one-list frames; the soak runs under gs_renderer_set_depth_layers(layers)
(-1 = the renderer chooses, for several frames in flight: two layers at the default workload), so it also holds the layered frames to the
one-list frames' bytes.  tile_order = 1: gs_renderer_set_tile_start_order(PREVIOUS_FRAME) on every renderer (round 6); every thread then walks
its own slice of the orbit in small steps with a jump now and then, so that the table engages, is withdrawn and engages again under load.
usage (GPU box): python tools/soak.py [seconds=60] [renderers=4] [splats=1000000] [layers=-1] [tile_order=0]"""
import ctypes as C
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402  (before the library: DESIGN.md section 8, load order)

import cudagaussianrenderer_amd as gs  # noqa: E402
from cudagaussianrenderer_amd import scenes  # noqa: E402


def free_bytes():
    free, total = torch.cuda.mem_get_info(0)
    return free


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
    layers = int(sys.argv[4]) if len(sys.argv) > 4 else -1
    tile_order = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    W, H, CHECK = 1920, 1080, 997
    lib = gs.lib()
    lib.gs_set_device(0)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "s.ply")
        scenes.write_synthetic_ply(path, n, sh_degree=3, seed=1234)
        ply = gs.parse_ply(path)
    cams = [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 360 if tile_order else 8)]
    if tile_order:
        # 8 stretches of 12 one-degree steps, 45 degrees apart: small steps inside a stretch, a jump between stretches
        cams = [cams[(45 * s + k) % 360] for s in range(8) for k in range(12)]
    streams = [torch.cuda.Stream(0) for _ in range(depth)]
    renderers = []
    for d in range(depth):
        r = gs.Renderer(W, H, storage_order="morton")
        if d == 0:
            r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
        else:
            r.share_scene(renderers[0])
        r.set_stream(streams[d].cuda_stream)
        r.set_graph(True)
        renderers.append(r)
    # reference images per view, from renderer 0 (plain frames of the same path)
    want = []
    for cam in cams:
        renderers[0].render(cam)
        renderers[0].render(cam)
        want.append(renderers[0].read_image().copy())
    for r in renderers:  # every renderer captures its frames and sizes its scratch before the memory baseline is taken
        r.set_depth_layers(layers)
        if tile_order:
            r.set_tile_start_order(gs.TILE_START_ORDER_PREVIOUS_FRAME)
        for cam in cams[:4]:
            r.render(cam)
    torch.cuda.synchronize(0)
    before = free_bytes()
    stop = time.perf_counter() + seconds
    frames = [0] * depth
    checks = [0] * depth
    bad = []

    def worker(d):
        torch.cuda.set_device(0)
        lib.gs_set_device(0)
        r = renderers[d]
        i = d * 24 if tile_order else d  # (tile order: every thread walks the path frame by frame from its own starting point)
        while time.perf_counter() < stop and not bad:
            view = i % len(cams)
            r.render(cams[view])
            frames[d] += 1
            if frames[d] % CHECK == 0:
                img = r.read_image()
                checks[d] += 1
                if not np.array_equal(img, want[view]):
                    bad.append((d, frames[d], view, int((img != want[view]).sum())))
            i += 1

    threads = [threading.Thread(target=worker, args=(d,)) for d in range(depth)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize(0)
    elapsed = time.perf_counter() - t0
    after = free_bytes()
    retries = [r.captured_retries() for r in renderers]
    if tile_order:
        print(f"tile start order: state of every renderer after its last frame (in force, sum |now - before|, sum now, frames seen) {[r.tile_order_state() for r in renderers]}")
    print(f"depth layers: setting {layers}, in use {[r.depth_layers_in_use() for r in renderers]}, choice changes {[int(lib.gs_renderer_depth_layer_changes(r.handle)) for r in renderers]}")
    print(f"{sum(frames)} frames in {elapsed:.1f} s = {sum(frames) / elapsed:.0f} frames/s on {depth} renderers; {sum(checks)} read-backs compared byte for byte, "
          f"{len(bad)} differing {bad[:3]}; captured retries {retries}; free device memory {before / 2**20:.0f} -> {after / 2**20:.0f} MiB")
    for r in reversed(renderers):
        r.close()
    sys.exit(1 if bad or abs(before - after) > 64 * 2**20 else 0)


if __name__ == "__main__":
    main()
