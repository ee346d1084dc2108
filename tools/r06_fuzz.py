#!/usr/bin/env python3
"""Round 6 fuzz: the two features whose correctness is "changes no byte" over random resolutions, bands, splat counts, scenes and cameras.
  (1) block-level band cull (Morton storage): kept-splat count, sorted list size and image equal with the block test on and off;
  (2) tile start order from the previous frame: frames of a random walk of the camera equal the screen-order renderer's.
usage: r06_fuzz.py ITERATIONS [SEED] [riding]   (GPU box)
"riding": the renderer under test in (2) stores its splats along the Morton curve and gs_debug_set_clip_scan_fusion is on -- the clip
kernel then rides in the launch of the depth sort's first column scan, and the tile start order's builder with it."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudagaussianrenderer_amd as gs

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib = gs.lib()
riding = len(sys.argv) > 3 and sys.argv[3] == "riding"
if riding:
    lib.gs_debug_set_clip_scan_fusion(1)


def harsh(n):
    pos = np.ones((n, 4), np.float32)
    centres = rng.uniform(-4, 4, (32, 3))
    pos[:, :3] = (centres[rng.integers(0, 32, n)] + rng.normal(0, rng.uniform(0.2, 1.0), (n, 3))).astype(np.float32)
    sr = np.zeros((n, 4), np.float32)
    base = np.exp(rng.uniform(np.log(1e-4), np.log(1.5), n))
    ratio = np.exp(rng.uniform(np.log(1e-3), 0.0, (n, 3)))
    ratio[np.arange(n), rng.integers(0, 3, n)] = 1.0
    sr[:, :3] = (base[:, None] * ratio).astype(np.float32)
    bits = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    if rng.random() < 0.5:  # half of the scenes: proper unit quaternions (encodeVec4), the others arbitrary bytes
        q = rng.normal(0, 1, (n, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        b = np.clip(((q + 1) * 0.5 * 255).astype(np.int64), 0, 255).astype(np.uint32)
        bits = (b[:, 0] << 24) | (b[:, 1] << 16) | (b[:, 2] << 8) | b[:, 3]
    f = bits.view(np.float32).copy()
    f[~np.isfinite(f)] = 0.0
    sr[:, 3] = f
    col = rng.uniform(0, 1, (n, 4)).astype(np.float32)
    col[:, 3] = rng.uniform(0.05, 0.95, n)
    return pos, sr, col


def camera(w, h, near_scene):
    q = rng.normal(0, 1, 4)
    q /= np.linalg.norm(q)
    p = rng.uniform(-5, 5, 3) if near_scene else rng.uniform(-12, 12, 3)
    return p, q, gs.camera_from_pose(p, q, w, h)


t0 = time.time()
cull_frames = order_frames = rejected = total_blocks = 0
for it in range(iters):
    w, h = int(rng.integers(16, 241)) * 16 - int(rng.integers(0, 16)), int(rng.integers(9, 136)) * 16 - int(rng.integers(0, 16))
    n = int(rng.integers(4100, 120_000))
    scene = harsh(n)
    tx, ty = gs.tile_counts(w, h)
    views = int(rng.choice([1, 1, 2, 3, 4]))
    graph = views > 1 or rng.random() < 0.5
    footprint = int(rng.choice([gs.FOOTPRINT_GAUSSIAN, gs.FOOTPRINT_GAUSSIAN, gs.FOOTPRINT_EPANECHNIKOV]))
    # (1) block cull on / off
    pair = []
    for blocks in (True, False):
        r = gs.Renderer(w, h, band_cull=1, views=views, storage_order=gs.STORAGE_ORDER_MORTON)
        r.set_block_cull(blocks)
        r.set_footprint(footprint)
        r.set_graph(graph)
        r.set_scene(*scene)
        pair.append(r)
    for _ in range(4):
        rb = int(rng.integers(0, ty))
        band = (rb, int(rng.integers(rb + 1, ty + 1)))
        cams = [camera(w, h, rng.random() < 0.7)[2] for _ in range(views)]
        got = []
        for r in pair:
            lib.gs_renderer_set_tile_rows(r.handle, band[0], band[1])
            if views == 1:
                r.render(cams[0])
            else:
                r.render_views(cams)
            got.append((r.band_kept_splats(), [r.read_image(view=v) for v in range(views)]))
        assert got[0][0] == got[1][0], ("block cull kept count", it, w, h, n, band, views, got[0][0], got[1][0])
        assert all(np.array_equal(a, b) for a, b in zip(got[0][1], got[1][1])), ("block cull image", it, w, h, n, band, views)
        kept, total = pair[0].block_cull_kept()
        rejected += total - kept
        total_blocks += total
        cull_frames += views
    for r in pair:
        r.close()
    # (2) tile start order: a random walk of the camera (small steps, now and then a jump), optionally a band and depth layers
    band = None
    if rng.random() < 0.4:
        rb = int(rng.integers(0, ty))
        band = (rb, int(rng.integers(rb + 1, ty + 1)))
    layers = int(rng.choice([1, 1, 2, 3]))
    ref = gs.Renderer(w, h, tile_rows=band)
    ref.set_footprint(footprint)
    ref.set_scene(*scene)
    r = gs.Renderer(w, h, tile_rows=band, storage_order=gs.STORAGE_ORDER_MORTON if riding else None)
    r.set_footprint(footprint)
    r.set_graph(graph)
    if graph:
        r.set_depth_layers(layers)
    r.set_tile_start_order(gs.TILE_START_ORDER_PREVIOUS_FRAME)
    r.set_scene(*scene)
    p, q, cam = camera(w, h, True)
    for step in range(10):
        if rng.random() < 0.15:
            p, q, cam = camera(w, h, True)
        else:
            p = p + rng.normal(0, 0.02, 3)
            q = q + rng.normal(0, 0.004, 4)
            q /= np.linalg.norm(q)
            cam = gs.camera_from_pose(p, q, w, h)
        ref.render(cam)
        r.render(cam)
        assert np.array_equal(r.read_image(), ref.read_image()), ("tile start order image", it, w, h, n, band, layers, step)
        order_frames += 1
    ref.close()
    r.close()
    if (it + 1) % 25 == 0:
        print(f"{it + 1} scenes: {cull_frames} band frames with the block cull on and off (kept counts and images equal; {100 * rejected / max(total_blocks, 1):.1f} % of the "
              f"blocks rejected), {order_frames} frames with the tile start order (images equal) -- {time.time() - t0:.0f} s", flush=True)
how = " (Morton storage, the clip kernel and the builder riding in the launch of the sort's first scan)" if riding else ""
print(f"r06_fuzz: {iters} random scenes, 241x129 ... 3840x2160 with ragged last tiles, 4 100 ... 120 000 harsh splats, 1-4 views per chain, plain and captured, both footprints: "
      f"{cull_frames} band frames equal with the block-level cull on and off ({100 * rejected / max(total_blocks, 1):.1f} % of the blocks rejected); "
      f"{order_frames} frames of random camera walks equal with gs_renderer_set_tile_start_order on and off" + how + "; no difference found")
