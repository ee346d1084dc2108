#!/usr/bin/env python3
"""Randomised comparison of the two staged builders (gs_set_staged_builder): SORTED (default: buildTileList emits in the order the sort
leaves, sortTileList looks first) against REFERENCE_ORDER + rocPRIM's 64-bit sort, over random resolutions, splat counts, scale ranges
(sub-pixel splats next to splats larger than the screen: many work units per group, rounds past the recorded ones, several flushes of
the emit pass's staging buffer), tile-row bands and capacities.  Lists (keys and values), ranges and images must be the same bytes.
usage (GPU box): python tools/staged_fuzz.py [iterations] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudagaussianrenderer_amd as gs  # noqa: E402
from cudagaussianrenderer_amd.staged import StagedFrame  # noqa: E402


def main():
    iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
    lib = gs.lib()
    worst = 0
    for it in range(iterations):
        w = int(rng.choice([16, 40, 256, 333, 640, 1000, 1920, 2560, 3840, 4800, 7680]))
        h = int(rng.choice([16, 24, 177, 256, 368, 600, 1080, 1440, 2160, 2700, 4320]))
        tiles = ((w + 15) // 16) * ((h + 15) // 16)
        n = int(rng.choice([0, 1, 63, 257, 1000, 5000, 20000, 60000]))
        lo = float(10 ** rng.uniform(-4, -1.5))
        hi = float(min(3.0, lo * 10 ** rng.uniform(0, 3.5)))
        if tiles > 20000:
            hi = min(hi, 0.3)  # keep the lists of the largest screens within a few hundred MB
        pos, sr, col = gs.generate_random_gaussians(max(n, 1), seed=int(rng.integers(1, 1 << 30)), min_scale=lo, max_scale=hi)
        pos, sr, col = pos[:n], sr[:n], col[:n]
        cam = gs.camera_from_bounds([-4] * 3, [4] * 3, w, h)
        ty = (h + 15) // 16
        rows = (0, 0)
        if rng.random() < 0.3 and ty > 2:
            a = int(rng.integers(0, ty - 1))
            rows = (a, int(rng.integers(a + 1, ty + 1)))
        probe = StagedFrame(pos, sr, col, cam, w, h, capacity=64, tile_rows=rows)
        probe.begin()
        probe.run_clip()
        probe.run_build()
        demand = int(lib.gs_last_tile_list_demand())
        if demand > 60_000_000:
            continue
        capacity = demand + 7 if rng.random() < 0.8 else max(1, int(demand * rng.uniform(0.05, 0.95)))
        out = {}
        for builder in (gs.STAGED_BUILDER_REFERENCE_ORDER, gs.STAGED_BUILDER_SORTED):
            lib.gs_set_staged_builder(builder)
            f = StagedFrame(pos, sr, col, cam, w, h, capacity=capacity, tile_rows=rows)
            f.begin()
            f.run_clip()
            keys, values, size = f.run_build()
            if builder == gs.STAGED_BUILDER_SORTED and size > 1:
                assert np.all(np.diff(keys) >= 0), (it, "not sorted out of buildTileList")
            sk, sv = f.run_sort()
            rngs = f.run_ranges()
            img = f.run_raster()
            out[builder] = (size, int(lib.gs_last_tile_list_demand()), sk, sv, rngs, img)
            f.free_sort_temp()
        lib.gs_set_staged_builder(gs.STAGED_BUILDER_SORTED)
        a, b = out[gs.STAGED_BUILDER_REFERENCE_ORDER], out[gs.STAGED_BUILDER_SORTED]
        what = f"iteration {it}: {w}x{h} rows {rows} n {n} scales {lo:.2g}..{hi:.2g} demand {demand} capacity {capacity}"
        assert a[0] == b[0] and a[1] == b[1] == demand, what
        if capacity >= demand:
            for i, name in ((2, "keys"), (3, "values"), (4, "ranges"), (5, "image")):
                assert np.array_equal(a[i], b[i]), (what, name)
        else:
            assert b[0] == capacity and len(set(zip(b[2].tolist(), b[3].tolist()))) == capacity, what
        worst = max(worst, demand)
        if it % 10 == 0:
            print(what + " ok", flush=True)
    print(f"{iterations} iterations, both builders agree (largest list {worst} pairs)")


if __name__ == "__main__":
    main()
