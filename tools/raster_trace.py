#!/usr/bin/env python3
"""Per-tile timeline of the raster kernel on the bench scene (diagnostics, gs_debug_set_raster_trace):
   python tools/raster_trace.py [--opacity-mean -3] [--view 0] [--ppl 2]
Prints the kernel's span, how busy the chip was over time, and the tiles on the critical path."""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--splats", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--opacity-mean", type=float, default=1.0)
    ap.add_argument("--view", type=int, default=0)
    ap.add_argument("--ppl", type=int, default=0)
    ap.add_argument("--frames", type=int, default=5)
    args = ap.parse_args()
    import torch  # noqa: F401  (load torch's HIP runtime first)

    import cudagaussianrenderer_amd as gs
    from cudagaussianrenderer_amd import scenes

    lib = gs.lib()
    lib.gs_set_device(0)
    if args.ppl:
        lib.gs_set_raster_pixels_per_lane(args.ppl)
    W, H = args.width, args.height
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "scene.ply")
        names, table = scenes.synthetic_vertex_table(args.splats, sh_degree=3, seed=1234)
        table[:, names.index("opacity")] += np.float32(args.opacity_mean - 1.0)
        scenes.write_ply(path, names, table)
        ply = gs.parse_ply(path)
    cams = [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 8)]
    tx, ty = gs.tile_counts(W, H)
    tiles = tx * ty
    r = gs.Renderer(W, H)
    r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
    trace = gs.DeviceBuffer(nbytes=32 * tiles, dtype=np.uint64)
    for _ in range(3):
        r.render(cams[args.view])
    lib.gs_debug_set_raster_trace(trace.ptr)
    for f in range(args.frames):
        trace.clear_memory(0)
        st = r.render(cams[args.view], timed=True)
        t = trace.copy_to(dtype=np.uint64).reshape(tiles, 4)
        start, end, count = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2].astype(np.int64)
        consumed = (t[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
        slot = (t[:, 3] >> np.uint64(32)).astype(np.int64)
        live = end > 0
        t0 = start[live].min()
        s_us, e_us = (start - t0) / 100.0, (end - t0) / 100.0
        span = e_us[live].max()
        dur = (e_us - s_us)
        work = live & (count > 0)
        print(f"frame {f}: raster {st.msRasterize * 1e3:.1f} us (events), trace span {span:.1f} us, tiles with work {work.sum()}, "
              f"P_c {consumed.sum()}, sum of tile durations {dur[work].sum():.0f} us -> {dur[work].sum() / span:.0f} tiles resident on average")
    # the last frame in detail
    print("\ntiles resident per 5 us bucket:")
    for b in np.arange(0, span, 5.0):
        active = (work & (s_us < b + 5) & (e_us > b)).sum()
        started = (work & (s_us >= b) & (s_us < b + 5)).sum()
        print(f"  {b:5.0f}-{b + 5:3.0f} us: {active:5d} resident, {started:5d} started")
    order = np.argsort(-e_us * work)
    print("\nlast tiles to finish:   tile  slot  list  consumed  start_us  end_us  dur_us")
    for i in order[:16]:
        print(f"                      {i:6d} {slot[i]:5d} {count[i]:5d} {consumed[i]:8d} {s_us[i]:9.1f} {e_us[i]:7.1f} {dur[i]:7.1f}")
    print("\nduration vs consumed entries (tiles with work):")
    for lo, hi in ((1, 32), (33, 64), (65, 128), (129, 256), (257, 512), (513, 1024), (1025, 1 << 30)):
        m = work & (consumed >= lo) & (consumed <= hi)
        if m.any():
            print(f"  consumed {lo:5d}-{min(hi, 99999):5d}: {m.sum():5d} tiles, mean {dur[m].mean():6.1f} us, max {dur[m].max():6.1f} us, "
                  f"us per 64 entries {64 * dur[m].sum() / consumed[m].sum():5.2f}")
    lib.gs_debug_set_raster_trace(None)
    r.close()


if __name__ == "__main__":
    main()
