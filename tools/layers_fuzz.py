#!/usr/bin/env python3
"""Randomised comparison of depth-layered frames (gs_renderer_set_depth_layers: 2..4 layers, default and random cuts, the live-splat
compaction on and off) with one-list frames of the same renderer settings: random resolutions (tile counts that are and are not
multiples of 32: the open-bit words' edges), splat counts, scale ranges (sub-pixel splats next to splats larger than the screen: boxes
beyond the live test's 32 x 8 tiles), opacity shifts (tiles that stop at once ... never), SH degree 0 and 3 (colours from the buffer and
on demand), one to three views per chain, storage orders, tile-row bands (culled lists).  The images must be the same bytes.
usage (GPU box): python tools/layers_fuzz.py [iterations] [seed]"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudagaussianrenderer_amd as gs  # noqa: E402
from cudagaussianrenderer_amd import scenes  # noqa: E402


def main():
    iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
    lib = gs.lib()
    layered_frames = skipped = 0
    fewest = 1.0
    tmp = tempfile.mkdtemp()
    for it in range(iterations):
        w = int(rng.choice([272, 333, 512, 640, 1000, 1280, 1920, 2064, 2560, 3840]))
        h = int(rng.choice([272, 368, 528, 600, 720, 1080, 1440, 2160]))
        n = int(rng.choice([4096, 5000, 20_000, 60_000, 150_000, 400_000]))
        degree = int(rng.choice([0, 3]))
        views = int(rng.choice([1, 1, 2, 3]))
        storage = str(rng.choice(["caller", "morton"]))
        mean = float(rng.uniform(-4.6, -2.2))
        sigma = float(rng.uniform(0.3, 1.4))
        names, table = scenes.synthetic_vertex_table(n, sh_degree=degree, seed=int(rng.integers(1, 1 << 30)), log_scale_mean=mean, log_scale_sigma=sigma)
        table[:, names.index("opacity")] += np.float32(rng.uniform(-6.0, 4.0))
        path = os.path.join(tmp, "s.ply")
        scenes.write_ply(path, names, table)
        ply = gs.parse_ply(path)
        scene = (ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
        poses = scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 5)
        cams = [gs.camera_from_pose(p, q, w, h) for p, q in poses][: max(views, 2)]
        layers = int(rng.integers(2, 5))
        fractions = None
        if rng.random() < 0.5:
            fractions = sorted(float(f) for f in rng.uniform(0.0005, 0.9995, layers - 1))
            if any(b - a < 1e-4 for a, b in zip(fractions, fractions[1:])):
                fractions = None
        compaction = int(rng.random() < 0.8)
        rows = None  # a tile-row band (with its cull: the culled list's length lives on the device, the layers' slices are cut there)
        ty = (h + 15) // 16
        if rng.random() < 0.35 and ty > 3:
            a = int(rng.integers(0, ty - 1))
            rows = (a, int(rng.integers(a + 1, ty + 1)))

        def frames(count, fr):
            r = gs.Renderer(w, h, storage_order=storage, views=views, tile_rows=rows)
            r.set_scene(*scene)
            r.set_graph(True)
            r.set_depth_layers(count, fr)
            lib.gs_renderer_debug_set_layer_compaction(r.handle, compaction)
            out, used, listed = [], 1, 0
            for rep in range(2):  # both graph instances
                if views > 1:
                    r.render_views(cams[:views])
                    out.append([r.read_image(v).copy() for v in range(views)])
                else:
                    for cam in cams:
                        r.render(cam)
                        out.append([r.read_image().copy()])
                used, listed = r.depth_layers_in_use(), r.listed_pairs()
            r.close()
            return out, used, listed

        want, _, full = frames(1, None)
        if full > 40_000_000:
            skipped += 1
            continue
        got, used, listed = frames(layers, fractions)
        what = (f"iteration {it}: {w}x{h} rows {rows} n {n} degree {degree} views {views} {storage} scales e^{mean:.2f}+-{sigma:.2f} layers {layers} "
                f"cuts {fractions} compaction {compaction}: in use {used}, listed {listed} of {full}")
        for a, b in zip(want, got):
            for x, y in zip(a, b):
                assert np.array_equal(x, y), (what, int((x != y).sum()))
        assert listed <= full, what
        if used > 1:
            layered_frames += 1
            fewest = min(fewest, listed / max(1, full))
        if it % 10 == 0:
            print(what + " ok", flush=True)
    print(f"{iterations} iterations ({skipped} skipped for size), {layered_frames} of them layered: every image equals the one-list image byte for byte "
          f"(smallest listed / total {fewest:.3f})")


if __name__ == "__main__":
    main()
