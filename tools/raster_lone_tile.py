#!/usr/bin/env python3
"""Latency of ONE tile's waves alone on the chip (diagnostics): N low-opacity sub-pixel splats that all project into the middle of one
tile, so that the raster kernel is a single workgroup walking a list of N entries with nothing else resident.  The slope of the kernel
time over N is the per-entry latency of a lone tile -- what sets the raster kernel's tail.
   python tools/raster_lone_tile.py [--ppl 2] [--degree 3] [--tiles 1]
--tiles K spreads the same list over K tiles of one row (K workgroups, still far from filling the chip)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ppl", type=int, default=0)
    ap.add_argument("--degree", type=int, default=3)
    ap.add_argument("--tiles", type=int, default=1)
    ap.add_argument("--sizes", default="512,1024,2048,4096")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--phases", action="store_true", help="library built with -DGS_RASTER_PHASE_TRACE: where thread 0 of the tile spends its time")
    args = ap.parse_args()
    import cudagaussianrenderer_amd as gs

    lib = gs.lib()
    lib.gs_set_device(0)
    if args.ppl:
        lib.gs_set_raster_pixels_per_lane(args.ppl)
    W, H = 1920, 1080
    cam = gs.camera_from_bounds([-4] * 3, [4] * 3, W, H)
    quat = np.array([gs.encode_vec4([0.5, 0.5, 0.5, 1.0])], np.uint32).view(np.float32)[0]
    rng = np.random.default_rng(1)
    rows = []
    for n in [int(s) for s in args.sizes.split(",")]:
        total = n * args.tiles
        pos = np.zeros((total, 4), np.float32)
        # the origin projects to the screen centre (a tile corner in x): +0.084 world units in x = +8 px; 16 px per further tile
        tile_of = np.repeat(np.arange(args.tiles), n)
        pos[:, 0] = 0.084 + 0.1677 * tile_of + rng.uniform(-1e-3, 1e-3, total)
        pos[:, 1] = rng.uniform(-1e-3, 1e-3, total)
        pos[:, 2] = rng.uniform(-1e-3, 1e-3, total)
        pos[:, 3] = 0.05  # opacity: far pixels of the tile never saturate, so the whole list is consumed
        sr = np.zeros((total, 4), np.float32)
        sr[:, :3] = 1e-3
        sr[:, 3] = quat
        col = np.full((total, 4), 0.5, np.float32)
        col[:, 3] = 0.05
        sh_count = 3 * (args.degree + 1) ** 2 if args.degree else 0
        sh = (rng.standard_normal((total, sh_count)) * 0.1).astype(np.float32) if args.degree else None
        r = gs.Renderer(W, H)
        r.set_scene(pos, sr, col, sh, args.degree, sh_count)
        lib.gs_set_collect_raster_stats(1)
        best, consumed, pairs = 1e9, 0, 0
        for _ in range(args.reps):
            st = r.render(cam, timed=True, stats=True)
            best = min(best, st.msRasterize)
            consumed, pairs = int(st.consumedPairs), int(st.tileListSize)
        lib.gs_set_collect_raster_stats(0)
        if args.phases:
            trace = gs.DeviceBuffer(nbytes=32 * 120 * 68, dtype=np.uint64)
            trace.clear_memory(0)
            lib.gs_debug_set_raster_trace(trace.ptr)
            r.render(cam, timed=True)
            lib.gs_debug_set_raster_trace(None)
            tr = trace.copy_to(dtype=np.uint64).reshape(-1, 4)
            live = np.nonzero(tr[:, 3] & np.uint64(0xFFFFFFFF))[0]
            for tile in live[:2]:
                w0, w1, w2, w3 = [int(v) for v in tr[tile]]
                ph = {"barriers": w0 >> 32, "gather wait": w0 & 0xFFFFFFFF, "stage (SH + LDS writes)": w1 >> 32, "blend": w1 & 0xFFFFFFFF, "vote": w2}
                cons = w3 & 0xFFFFFFFF
                print(f"   tile {tile}: consumed {cons}; ns per entry: " + ", ".join(f"{k} {10.0 * v / cons:.1f}" for k, v in ph.items()) +
                      f" | total {10.0 * sum(ph.values()) / cons:.1f}")
        rng_t = r.download(r.buffers().tileRange, 2 * 120 * 68, np.int32).reshape(-1, 2)
        lengths = (rng_t[:, 1] - rng_t[:, 0])
        r.close()
        rows.append((n, best * 1e3, consumed, pairs, int((lengths > 0).sum()), int(lengths.max())))
        print(f"N={n:5d} x {args.tiles} tiles: raster {best * 1e3:8.1f} us  pairs {pairs} consumed {consumed} tiles with work {rows[-1][4]} longest list {rows[-1][5]}")
    if len(rows) >= 2:
        (n0, t0, *_), (n1, t1, *_) = rows[0], rows[-1]
        print(f"slope: {1e3 * (t1 - t0) / (n1 - n0):.1f} ns per list entry (lone tile, ppl {lib.gs_get_raster_pixels_per_lane()}, SH degree {args.degree})")


if __name__ == "__main__":
    main()
