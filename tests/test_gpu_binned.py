"""GPU tests of the renderer's binned pipeline (gs_bins.hip): depth-sorted splats written straight to their slot of the
per-tile lists.  Its value list, range table and image must be IDENTICAL to the two-level (sort-based) pipeline's --
both walk the splats with the same AABB and the same overlap arithmetic -- and through that to the reference-shaped
pipeline and the oracle (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

from tests.test_gpu_parity import _ply_scene, _procedural

pytestmark = pytest.mark.gpu


def _lists(gs, r, w, h):
    b = r.buffers()
    tx, ty = gs.tile_counts(w, h)
    values = r.download(b.values, b.tileListSize, np.int32)
    ranges = r.download(b.tileRange, 2 * tx * ty, np.int32).reshape(-1, 2)
    return b.tileListSize, values, ranges


def _render_both(gs, w, h, scene, cams, rows=None, band_cull=None, graph=False):
    out = []
    for pipeline in (gs.PIPELINE_TWO_LEVEL, gs.PIPELINE_BINNED):
        r = gs.Renderer(w, h, tile_rows=rows, pipeline=pipeline, band_cull=band_cull)
        r.set_graph(graph)
        r.set_scene(*scene)
        frames = []
        for cam in cams:
            r.render(cam)
            frames.append((_lists(gs, r, w, h), r.read_image()))
        r.close()
        out.append(frames)
    return out


def _assert_same(two_level, binned):
    for ((size_a, values_a, ranges_a), image_a), ((size_b, values_b, ranges_b), image_b) in zip(two_level, binned):
        assert size_a == size_b
        assert np.array_equal(ranges_a, ranges_b)
        assert np.array_equal(values_a, values_b)
        assert np.array_equal(image_a, image_b)


@pytest.mark.parametrize("w,h,n,scales", [(640, 360, 50000, (0.01, 0.3)), (1920, 1080, 200000, (0.005, 0.08)), (333, 177, 20000, (0.01, 1.0)),
                                          (1920, 1080, 3000, (0.5, 3.0)), (64, 48, 1, (0.1, 0.2))])
def test_binned_equals_two_level(gs, w, h, n, scales):
    """Procedural splats from tiny to screen-filling (AABBs of thousands of tiles), odd resolutions, a single splat."""
    pos, sr, col = _procedural(gs, n, seed=21, min_scale=scales[0], max_scale=scales[1])
    cam = gs.camera_from_bounds([-4] * 3, [4] * 3, w, h)
    a, b = _render_both(gs, w, h, (pos, sr, col), [cam])
    assert a[0][0][0] > 0
    _assert_same(a, b)


@pytest.mark.parametrize("graph", [False, True])
def test_binned_bands_cull_and_orbit(gs, tmp_path, graph):
    """SH scene over an orbit: full frame, full frame with the visible-splat compaction forced, and culled sub-bands;
    regular launches and captured frames."""
    from cudagaussianrenderer_amd import scenes

    w, h = 1280, 720
    ply = _ply_scene(gs, tmp_path, 80000, 3, log_scale_mean=-3.3)
    scene = (ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
    cams = [gs.camera_from_pose(p, q, w, h) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], 4)]
    for rows, cull in [(None, None), (None, 1), ((14, 31), None), ((0, 3), None), ((44, 45), 2)]:
        a, b = _render_both(gs, w, h, scene, cams + cams[:1], rows=rows, band_cull=cull, graph=graph)
        _assert_same(a, b)


def test_binned_every_row_length(gs):
    """GS_BIN_BATCHES picks the number of 64-splat batches per row of the count table: the result may not depend on it."""
    import os

    w, h = 800, 450
    pos, sr, col = _procedural(gs, 30011, seed=4, min_scale=0.01, max_scale=0.4)
    cam = gs.camera_from_bounds([-4] * 3, [4] * 3, w, h)
    want = None
    try:
        for batches in (1, 2, 3, 8, 16):
            os.environ["GS_BIN_BATCHES"] = str(batches)
            r = gs.Renderer(w, h, pipeline=gs.PIPELINE_BINNED)
            r.set_scene(pos, sr, col)
            r.render(cam)
            got = (_lists(gs, r, w, h), r.read_image())
            r.close()
            if want is None:
                want = got
            else:
                assert got[0][0] == want[0][0]
                assert np.array_equal(got[0][1], want[0][1]) and np.array_equal(got[0][2], want[0][2])
                assert np.array_equal(got[1], want[1])
    finally:
        os.environ.pop("GS_BIN_BATCHES", None)


@pytest.mark.parametrize("graph", [False, True])
def test_binned_overflow_then_recover(gs, graph):
    """Demand above the provisioned 8 pairs per splat: the regular path grows the list and emits again inside the frame;
    a captured frame is truncated and the next one runs with a doubled list (Demo.cpp:357-366,447)."""
    w = h = 512
    pos, sr, col = _procedural(gs, 3000, seed=9, min_scale=0.3, max_scale=1.5)
    cam = gs.camera_from_bounds([-4] * 3, [4] * 3, w, h)
    ref = gs.Renderer(w, h, pipeline=gs.PIPELINE_TWO_LEVEL)
    ref.set_scene(pos, sr, col)
    st = ref.render(cam, stats=True)
    want = (_lists(gs, ref, w, h), ref.read_image())
    assert st.tileListSize > 8 * 3000
    ref.close()
    r = gs.Renderer(w, h, pipeline=gs.PIPELINE_BINNED)
    r.set_graph(graph)
    r.set_scene(pos, sr, col)
    for _ in range(6 if graph else 1):
        r.render(cam)
    got = (_lists(gs, r, w, h), r.read_image())
    r.close()
    _assert_same([want], [got])


def test_binned_full_size_determinism_and_order(gs, tmp_path):
    """BASELINE.json configs[2] shape (1 M splats, SH 3, 1080p): lists ordered by (quantised depth, splat index) inside
    every tile, ranges tile the list exactly, two renders give the same bytes."""
    from cudagaussianrenderer_amd import scenes

    n, w, h = 1_000_000, 1920, 1080
    path = str(tmp_path / "c3.ply")
    scenes.write_synthetic_ply(path, n, sh_degree=3, seed=1234)
    ply = gs.parse_ply(path)
    cam = gs.camera_from_bounds(ply["boundsMin"], ply["boundsMax"], w, h)
    r = gs.Renderer(w, h, pipeline=gs.PIPELINE_BINNED)
    r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], 3, 48)
    st = r.render(cam, stats=True)
    size, values, ranges = _lists(gs, r, w, h)
    image = r.read_image()
    assert size == st.tileListSize > n
    counts = np.where(ranges[:, 0] >= 0, ranges[:, 1] - ranges[:, 0], 0)
    assert counts.sum() == size
    nonempty = counts > 0
    assert np.array_equal(ranges[nonempty, 0], (np.cumsum(counts) - counts)[nonempty])
    tiles = np.repeat(np.arange(len(counts)), counts)
    b = r.buffers()
    z = r.download(b.positionClipSpaceZ, n, np.float32)
    depth = np.floor(np.clip((z.astype(np.float64) + 1.0) * 0.5, 0.0, 1.0).astype(np.float32).astype(np.float64) * 2.0 ** 32)
    depth = np.minimum(depth, 2.0 ** 32 - 1).astype(np.uint64)
    full = (tiles.astype(np.uint64) << np.uint64(32)) | depth[values]
    assert np.all(full[1:] >= full[:-1])
    same = full[1:] == full[:-1]
    assert np.all(values[1:][same] > values[:-1][same])
    r.render(cam)
    assert np.array_equal(r.read_image(), image)
    r.close()
