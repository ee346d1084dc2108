#!/usr/bin/env python3
"""Round 6, DESIGN.md section 9 item 3: would starting the raster's long tiles first shorten its tail -- with the BEST predictor a host
could have?  BASELINE.json configs[2] (1 M splats, SH 3, 1920x1080), plain frames one at a time, raster stage by HIP events.
For every view of the orbit: (1) the XCD-aware screen order (the product); (2) the same camera again with the tiles started in the order of
THEIR OWN consumed entries, longest first (a perfect predictor: what a camera that does not move would have from its previous frame), tiles
kept on the XCD of their 8 x 4 block; (3) the same, XCDs ignored (slot q = q-th longest tile); (4) the order predicted by the frame of a
camera 1 degree (and 5 degrees) back on the orbit -- the interactive viewer's case (Demo.cpp:371-392).  Frames are byte-identical in all
orders (checked).  usage: tile_order_experiment.py [--splats N] [--frames F]"""
import argparse, ctypes as C, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudagaussianrenderer_amd as gs
from cudagaussianrenderer_amd import scenes

ap = argparse.ArgumentParser()
ap.add_argument("--splats", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--frames", type=int, default=48)
args = ap.parse_args()
W, H = args.width, args.height
lib = gs.lib()
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "s.ply")
    scenes.write_synthetic_ply(path, args.splats, sh_degree=3, seed=1234)
    ply = gs.parse_ply(path)
tx, ty = gs.tile_counts(W, H)
T = tx * ty
BW, BH = 8, 4


def orbit(count):
    return [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], count)]


r = gs.Renderer(W, H, storage_order="morton")
r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
slots = lib.gs_renderer_debug_raster_grid_slots(r.handle)
assert lib.gs_renderer_debug_tile_consumed(r.handle, None) == 1  # armed


def raster_ms(cam, frames):
    for _ in range(4):
        r.render(cam, timed=2)
    r.collect_stage_times()
    for _ in range(frames):
        r.render(cam, timed=2)
    n, sums = r.collect_stage_times()
    return 1e3 * sums.msRasterize / n


def consumed(cam):
    r.render(cam)
    out = np.zeros(T, np.int32)
    assert lib.gs_renderer_debug_tile_consumed(r.handle, out.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    return out


def set_order(table):
    arr = None if table is None else np.ascontiguousarray(table, np.int32)
    rc = lib.gs_renderer_debug_set_tile_order(r.handle, None if arr is None else arr.ctypes.data_as(C.POINTER(C.c_int32)), 0 if arr is None else len(arr))
    assert rc == 0


def screen_order_xcd():
    """XCD of every tile under the product's map (blocks of 8 x 4 tiles dealt round-robin), and the product's own slot -> tile table."""
    table = np.full(slots, -1, np.int32)
    blocks_x = (tx + BW - 1) // BW
    for q in range(slots):
        xcd, i = q & 7, q >> 3
        j = (i // (BW * BH)) * 8 + xcd
        inside = i % (BW * BH)
        bx, by = j % blocks_x, j // blocks_x
        x, y = bx * BW + inside % BW, by * BH + inside // BW
        if x < tx and y < ty:
            table[q] = y * tx + x
    xcd_of = np.full(T, -1, np.int32)
    valid = table >= 0
    xcd_of[table[valid]] = (np.arange(slots)[valid] & 7)
    return table, xcd_of


product_table, xcd_of = screen_order_xcd()
assert (np.sort(product_table[product_table >= 0]) == np.arange(T)).all()


def order_by(cons, keep_xcd):
    table = np.full(slots, -1, np.int32)
    if keep_xcd:
        for x in range(8):
            mine = np.flatnonzero(xcd_of == x)
            mine = mine[np.argsort(-cons[mine], kind="stable")]
            table[x + 8 * np.arange(len(mine))] = mine
    else:
        order = np.argsort(-cons, kind="stable")
        table[: T] = order
    return table


views = orbit(8)
fine = orbit(360)  # 1-degree steps
rows = []
for v, cam in enumerate(views):
    set_order(None)
    base_img = None
    r.render(cam)
    base_img = r.read_image()
    t_screen = raster_ms(cam, args.frames)
    cons = consumed(cam)
    set_order(product_table)  # the product's own order through the table: what the indirection itself costs
    t_table = raster_ms(cam, args.frames)
    set_order(order_by(cons, True))
    r.render(cam)
    assert np.array_equal(r.read_image(), base_img)
    t_own_xcd = raster_ms(cam, args.frames)
    set_order(order_by(cons, False))
    r.render(cam)
    assert np.array_equal(r.read_image(), base_img)
    t_own_global = raster_ms(cam, args.frames)
    # what the product's order kernel does differently from a plain sort (gs_raster.hip tileStartOrderKernel): the key is the largest
    # consumption in the tile's 3 x 3 neighbourhood, keys are bucketed by vote chunks of 32, and the order inside a bucket is arbitrary
    grid = cons.reshape(ty, tx)
    pad = np.pad(grid, 1)
    dil = np.max([pad[1 + dy: 1 + dy + ty, 1 + dx: 1 + dx + tx] for dy in (-1, 0, 1) for dx in (-1, 0, 1)], axis=0).reshape(-1)
    rng = np.random.default_rng(v)
    variants = {}
    for name, key in (("dilated", dil), ("bucketed", (cons >> 5) * 64 + rng.integers(0, 64, T)), ("dilated+bucketed", (dil >> 5) * 64 + rng.integers(0, 64, T))):
        set_order(order_by(key.astype(np.int64), True))
        r.render(cam)
        assert np.array_equal(r.read_image(), base_img)
        variants[name] = raster_ms(cam, args.frames)
    print(f"view {v}: own consumption with the order kernel's key: " + " | ".join(f"{k} {t:.1f}" for k, t in variants.items()), flush=True)
    # ... and the product itself: gs_renderer_set_tile_start_order on the SAME camera again and again -- the order kernel's table from a
    # perfect predictor (then the experiment's manual mode again)
    set_order(None)
    r.set_tile_start_order(gs.TILE_START_ORDER_PREVIOUS_FRAME)
    t_product = raster_ms(cam, args.frames)
    state = r.tile_order_state()
    r.render(cam)
    assert np.array_equal(r.read_image(), base_img)
    r.set_tile_start_order(gs.TILE_START_ORDER_SCREEN)
    print(f"view {v}: the order kernel's own table, same camera repeated: {t_product:.1f} us (state {state})", flush=True)
    preds = []
    for back in (1, 5):
        set_order(None)
        prev = consumed(fine[(45 * v - back) % 360])
        set_order(order_by(prev, True))
        preds.append(raster_ms(cam, args.frames))
    rows.append((v, int(cons.sum()), int(cons.max()), t_screen, t_table, t_own_xcd, t_own_global, preds[0], preds[1]))
    print(f"view {v}: consumed {cons.sum()} (longest tile {cons.max()}), raster us: screen order {t_screen:.1f} | same order through the table {t_table:.1f} | "
          f"own consumption, XCDs kept {t_own_xcd:.1f} | own consumption, global {t_own_global:.1f} | predicted from 1 deg back {preds[0]:.1f} | 5 deg back {preds[1]:.1f}", flush=True)
m = np.array([row[3:] for row in rows]).mean(0)
print(f"mean over the orbit, raster us: screen order {m[0]:.1f} | through the table {m[1]:.1f} | own consumption (XCDs kept) {m[2]:.1f} ({100 * (m[2] / m[0] - 1):+.1f} %) | "
      f"own consumption (global) {m[3]:.1f} ({100 * (m[3] / m[0] - 1):+.1f} %) | 1 deg back {m[4]:.1f} ({100 * (m[4] / m[0] - 1):+.1f} %) | 5 deg back {m[5]:.1f} ({100 * (m[5] / m[0] - 1):+.1f} %)")
r.close()
