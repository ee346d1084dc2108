#!/usr/bin/env python3
"""CPU (oracle): would another START ORDER of the raster kernel's tiles shorten its tail?

Per tile of three views of the bench scene the oracle gives the list length L and the entries the reference's semantics consume, c.
(1) What L says about c.  (2) A processor-sharing model of the kernel calibrated on profiles/r0N_raster_trace.txt -- 2 048 tile
slots (8 two-wave workgroups per CU), 166 ns per entry and tile while the chip is full, 62 ns for a tile that has its SIMDs to
itself, 3 us of fixed cost per tile -- run over several dispatch orders, the oracle's own (longest consumption first) among them,
and over a two-kernel split (every tile up to K entries, the rest by a kernel whose lone tiles run 1.7x faster: the entry-split
kernel of tools/experiments/raster_entry_split.txt).
usage: GSPLAT_NO_GPU=1 python tools/raster_schedule_sim.py [raw opacity mean, default 1.0]"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gsref  # noqa: E402
from cudagaussianrenderer_amd import scenes  # noqa: E402
import cudagaussianrenderer_amd as gs  # noqa: E402


def tiles_of(view, ply, cams, W, H):
    tx, ty = gs.tile_counts(W, H)
    buf = np.zeros(tx * ty, np.int32)
    gsref.lib().gsref_set_consumed_per_tile_buffer(buf.ctypes.data_as(C.POINTER(C.c_int32)))
    gsref.render(ply["position"], ply["scaleAndRotation"], ply["color"], None, 0, 0, cams[view], W, H)
    gsref.lib().gsref_set_consumed_per_tile_buffer(None)
    xy, zz, ell, con = gsref.evaluate_splat_clip_data(ply["position"], ply["scaleAndRotation"], cams[view], W, H)
    k, v, total = gsref.build_tile_list(xy, zz, ell, W, H)
    k, v = gsref.sort_tile_list(k, v, gsref.sort_end_bit(W, H))
    rng = gsref.evaluate_tile_ranges(k, tx * ty).reshape(-1, 2)
    length = (rng[:, 1] - rng[:, 0]).astype(np.int32)
    keep = length > 0
    return buf[keep], length[keep]


def simulate(work, order, slots=2048, lone=62.0, full=166.0, overhead=3000.0, dt=100.0):
    """Makespan in us: tiles start in `order` as slots free up; every resident tile advances at min(1 / lone, (slots / full) / residents)."""
    rem = work[order].astype(np.float64)
    ovh = np.full(len(rem), overhead)
    nxt, active, t, done, n = 0, [], 0.0, 0, len(rem)
    thr = slots / full
    while done < n:
        while len(active) < slots and nxt < n:
            active.append(nxt)
            nxt += 1
        a = np.array(active)
        r = min(1.0 / lone, thr / len(a))
        spend = np.minimum(ovh[a], dt)
        ovh[a] -= spend
        rem[a] -= (dt - spend) * r
        fin = rem[a] <= 0
        done += int(fin.sum())
        active = [x for x, f in zip(active, fin) if not f]
        t += dt
    return t / 1000.0


def main():
    opacity = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    n, W, H = 1_000_000, 1920, 1080
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "s.ply")
        names, table = scenes.synthetic_vertex_table(n, sh_degree=0, seed=1234, log_scale_mean=-3.6)
        if opacity != 1.0:
            table[:, names.index("opacity")] += np.float32(opacity - 1.0)
        scenes.write_ply(path, names, table)
        ply = gs.parse_ply(path)
    cams = [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 8)]
    print(f"bench scene (1 M splats, seed 1234, raw opacity mean {opacity}; colours from SH band 0 -- lists and consumption do not depend on them), 1920x1080")
    for view in (0, 1, 3):
        c, L = tiles_of(view, ply, cams, W, H)
        print(f"\nview {view}: {len(c)} tiles with work, {int(L.sum())} pairs, {int(c.sum())} consumed, corr(L, c) = {np.corrcoef(L, c)[0, 1]:.2f}")
        bins = [1, 65, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 4097, 10 ** 9]
        for a, b in zip(bins[:-1], bins[1:]):
            s = (L >= a) & (L < b)
            if s.sum():
                print(f"  L in [{a:5d}, {b if b < 10**9 else 'inf':>5}): {int(s.sum()):5d} tiles, consumed mean {c[s].mean():6.1f}, max {int(c[s].max()):4d}, "
                      f"list exhausted {np.mean(c[s] >= L[s] - 31):4.0%}, consumed >= 192: {int(np.sum(c[s] >= 192)):4d}")
        rng = np.random.default_rng(0)
        orders = {"screen order (the product)": np.arange(len(c)), "longest list first": np.argsort(-L, kind="stable"),
                  "lists of <= 128 entries last": np.argsort(L <= 128, kind="stable"), "random": rng.permutation(len(c)),
                  "ORACLE: longest consumption first": np.argsort(-c, kind="stable")}
        print("  modelled kernel time by start order: " + "; ".join(f"{k} {simulate(c, o):.1f} us" for k, o in orders.items()))
        floor = max(c.sum() * 166.0 / 2048 / 1000 + len(c) * 3.0 / 2048, c.max() * 62.0 / 1000 + 3.0)
        print(f"  (bounds: all slots busy all the time {c.sum() * 166.0 / 2048 / 1000 + len(c) * 3.0 / 2048:.1f} us; the longest tile alone {c.max() * 62.0 / 1000 + 3.0:.1f} us -> no order beats {floor:.1f} us)")
        for K in (96, 128, 192):
            a = simulate(np.minimum(c, K), np.arange(len(c)))
            rest = c[c > K] - K
            b = simulate(rest, np.arange(len(rest)), slots=1024, lone=36.0, full=166.0 * 1.08 / 2) if len(rest) else 0.0
            print(f"  two kernels, cut at {K} entries: {a:.1f} + {b:.1f} us (+ ~2 us between them) for {len(rest)} tiles / {int(rest.sum())} entries in the second")


if __name__ == "__main__":
    main()
