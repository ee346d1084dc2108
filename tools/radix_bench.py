#!/usr/bin/env python3
"""Times the renderer's radix sort alone on a tile-sort-shaped input (u16 keys in emission order, int32 values)."""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudagaussianrenderer_amd as gs

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=5_450_000)
ap.add_argument("--bits", type=int, default=13)
ap.add_argument("--key-bytes", type=int, default=2)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--random", type=int, default=0)
ap.add_argument("--backend", type=int, default=1, help="1 native, 0 rocPRIM")
args = ap.parse_args()
rng = np.random.default_rng(1)
n = args.n
dt = np.uint16 if args.key_bytes == 2 else np.uint32
if args.random:
    keys = rng.integers(0, 1 << args.bits, n, dtype=np.uint64).astype(dt)
else:
    # emission order of the tile list: runs of ~14 pairs per splat walking a small rectangle of tiles
    tiles_x = 120
    runs = n // 14 + 1
    x0 = rng.integers(0, tiles_x - 4, runs)
    y0 = rng.integers(0, (1 << args.bits) // tiles_x - 4, runs)
    k = np.arange(14)
    keys = ((y0[:, None] + k[None, :] // 4) * tiles_x + x0[:, None] + k[None, :] % 4).reshape(-1)[:n].astype(dt)
values = np.arange(n, dtype=np.int32)
kb = [gs.DeviceBuffer(keys), gs.DeviceBuffer(nbytes=keys.nbytes, dtype=dt)]
vb = [gs.DeviceBuffer(values), gs.DeviceBuffer(nbytes=values.nbytes, dtype=np.int32)]
lib = gs.lib()
lib.gs_set_sort_backend(args.backend)
for i in range(args.reps + 5):
    if i == 5:
        lib.gs_synchronize()
        t = time.perf_counter()
    lib.gs_radix_sort_pairs(args.key_bytes, kb[0].ptr, kb[1].ptr, vb[0].ptr, vb[1].ptr, n, 0, args.bits)
lib.gs_synchronize()
print(f"n={n} bits={args.bits}: {1e6 * (time.perf_counter() - t) / args.reps:.1f} us per sort (backend {args.backend})")
