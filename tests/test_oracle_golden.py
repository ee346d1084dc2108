"""The CPU oracle against the fixtures generated from the reference itself (tests/golden/make_golden.py)
and against analytic known answers."""
import json
import os

import numpy as np
import pytest


def _load_npz(path):
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", ["ply_l3_64", "ply_l0_32"])
def test_parse_ply_matches_reference(gsref, golden_dir, name):
    want = _load_npz(os.path.join(golden_dir, name + ".npz"))
    got = gsref.parse_ply(os.path.join(golden_dir, name + ".ply"))
    assert got["vertexCount"] == int(want["vertexCount"])
    assert got["shDegree"] == int(want["shDegree"])
    assert got["shCount"] == int(want["shCount"])
    for key in ["boundsMin", "boundsMax", "position", "scaleAndRotation", "color", "sh"]:
        # Bit-exact: same float ops (exp, sqrt, divide) on the same libm.
        assert np.array_equal(got[key].view(np.uint32), np.asarray(want[key], np.float32).view(np.uint32)), key


def test_parse_ply_errors_match_reference(gsref, golden_dir, tmp_path):
    cases = json.load(open(os.path.join(golden_dir, "ply_errors.json")))
    assert len(cases) >= 10
    for name, case in cases.items():
        p = tmp_path / (name + ".ply")
        p.write_bytes(case["header"].encode("ascii") + b"\0" * 512)
        with pytest.raises(gsref.PlyError) as e:
            gsref.parse_ply(str(p))
        assert str(e.value) == case["error"], name


def test_encode_vec4_matches_reference(gsref, golden_dir):
    for row in json.load(open(os.path.join(golden_dir, "encode_vec4.json"))):
        assert gsref.encode_vec4(row["v"]) == row["bits"]
    # identity quaternion (x,y,z,w) = (0,0,0,1) -> (0.5,0.5,0.5,1) -> 0x7F7F7FFF, decodes to ~(-0.0039.., 1)
    assert gsref.encode_vec4([0.5, 0.5, 0.5, 1.0]) == 0x7F7F7FFF
    d = gsref.decode_vec4(0x7F7F7FFF) * 2 - 1
    assert np.allclose(d, [-0.00392157, -0.00392157, -0.00392157, 1.0], atol=1e-7)


def test_camera_default_pose_matches_reference(gsref, golden_dir):
    """The oracle's default pose against the reference's own CameraControls.cpp: equal, field by field."""
    for case in json.load(open(os.path.join(golden_dir, "camera_default_pose.json"))):
        cam = gsref.camera_from_bounds(case["boundsMin"], case["boundsMax"], case["width"], case["height"])
        for name in ("position", "view", "projection", "viewProjection", "fovCotangent", "depthScaleBias"):
            assert np.array_equal(np.array(getattr(cam, name)[:], np.float32), np.array(case[name], np.float32)), name
        assert np.float32(cam.aspect) == np.float32(case["aspect"])


def test_sh_basis_matches_sh_gen(gsref, golden_dir):
    data = json.load(open(os.path.join(golden_dir, "sh_basis.json")))
    scale = np.array([0.1, 0.2, -0.1], np.float32)  # keeps result + 0.5 inside (0,1): no clamp
    for sample in data["samples"]:
        basis = np.array(sample["basis"])
        assert basis.shape == (25,)
        for degree, count in [(0, 1), (1, 4), (2, 9), (3, 16), (4, 25)]:
            for k in range(count):
                sh = np.zeros(75, np.float32)
                sh[3 * k:3 * k + 3] = scale
                rgb = gsref.spherical_harmonics(degree, sample["dir"], sh)
                assert np.allclose((rgb - 0.5) / scale, basis[k], atol=3e-5), (degree, k)
        # coefficients above the requested degree are ignored
        sh = np.zeros(75, np.float32)
        sh[3 * 4:3 * 4 + 3] = scale
        assert np.allclose(gsref.spherical_harmonics(1, sample["dir"], sh), 0.5)
    # clamp(result + 0.5, 0, 1) (GaussianRender.cu:154)
    sh = np.zeros(75, np.float32)
    sh[0:3] = [10.0, -10.0, 0.0]
    assert np.allclose(gsref.spherical_harmonics(0, [0, 0, 1], sh), [1.0, 0.0, 0.5])


def test_get_key_known_answers(gsref):
    assert gsref.get_key(5, 0.0) == 0x0000000580000000
    assert gsref.get_key(0, -1.0) == 0
    assert gsref.get_key(0, -5.0) == 0
    assert gsref.get_key(7, 1.0) == (7 << 32) | 0xFFFFFFFF  # 2^32 overflow saturates (SURVEY hard part 5)
    assert gsref.get_key(4095, 0.5) == (4095 << 32) | 0xC0000000


def test_realign_spherical_harmonics_layout(gsref):
    n, s = 300, 12
    src = np.arange(n * s, dtype=np.float32).reshape(n, s)
    dst = gsref.realign_spherical_harmonics(src, s, n)
    assert dst.shape[0] == 2 * 256 * s
    for splat in [0, 1, 255, 256, 299]:
        g, lane = divmod(splat, 256)
        for i in [0, 5, 11]:
            assert dst[g * 256 * s + i * 256 + lane] == src[splat, i]
    assert dst[1 * 256 * s + 0 * 256 + 44] == 0.0  # padded tail of the last group


def test_epanechnikov_footprint_known_answers(gsref):
    """GaussianRender.cu:8-9 EPANECHNIKOV_KERNEL as a switch: extent sqrt(7) sigma instead of 3 sigma (.cu:298-302), density
    1 - q/7 instead of exp(-q/2) (.cu:977-981).  One splat, so a pixel's colour is opacity * density * rgb, from float64."""
    w = h = 64
    pos = np.array([[0.1, -0.2, 0.3, 0.8]], np.float32)  # xyz + opacity
    sr = np.array([[0.9, 0.5, 0.7, 0.0]], np.float32)
    sr[0, 3] = np.array([gsref.encode_vec4([0.2, -0.3, 0.1, 0.9])], np.uint32).view(np.float32)[0]
    col = np.array([[0.9, 0.5, 0.25, 0.8]], np.float32)
    cam = gsref.camera_from_bounds([-2, -2, -2], [2, 2, 2], w, h)
    out = {}
    try:
        for epan in (False, True):
            gsref.set_footprint(epan)
            xy, z, ell, con = gsref.evaluate_splat_clip_data(pos, sr, cam, w, h)
            img, st, _ = gsref.render(pos, sr, col, None, 0, 0, cam, w, h)
            out[epan] = (xy, ell, con, img)
    finally:
        gsref.set_footprint(False)
    (xy_g, ell_g, con_g, img_g), (xy_e, ell_e, con_e, img_e) = out[False], out[True]
    assert np.array_equal(xy_g, xy_e) and np.array_equal(con_g, con_e)
    assert np.allclose(ell_e[:, 2:] / ell_g[:, 2:], np.sqrt(7.0) / 3.0, rtol=1e-6)  # extents only; same axis
    assert np.array_equal(ell_g[:, :2], ell_e[:, :2])
    px = np.arange(w, dtype=np.float64) * (2.0 / w) - 1.0
    py = np.arange(h, dtype=np.float64) * (2.0 / h) - 1.0
    dx = px[None, :] - float(xy_g[0, 0])
    dy = py[:, None] - float(xy_g[0, 1])
    a, b, c = [float(v) for v in con_g[0, :3]]
    q = a * dx * dx + c * dy * dy + 2.0 * b * dx * dy
    for img, density in ((img_g, np.exp(-0.5 * q)), (img_e, 1.0 - q / 7.0)):
        alpha = 0.8 * np.clip(density, 0.0, 1.0)
        touched = img[..., 3] == 255  # tiles that had work
        assert touched.any()
        for ch, rgb in enumerate([0.9, 0.5, 0.25]):
            want = np.floor(np.clip(rgb * alpha, 0, 1) * 255.0)
            diff = np.abs(img[..., ch].astype(np.int64) - want.astype(np.int64))[touched]
            assert diff.max() <= 1
    # compact support: outside q >= 7 the Epanechnikov splat contributes nothing, the Gaussian still does
    far = (q > 7.5) & (img_g[..., 3] == 255) & (img_e[..., 3] == 255)
    if far.any():
        assert img_e[..., 0][far].max() == 0


def test_glm_primitives_match_glm_itself_bit_for_bit(gsref, golden_dir):
    """The hand-written glm restatements of oracle/gsref.cpp (mat3_cast of un-normalised quaternions, mat3*mat3, transpose,
    mat4*vec4, (mat3) of a mat4, normalize incl. (0,0), atan(y,x), max/min/step/mix/clamp, length2/dot, sqrt/cos/sin/floor/ceil) against
    glm 1.0.1's own results on ~1 800 inputs (tests/golden/glm_ops.json, generated by tests/golden/make_golden.py from
    oracle/glm_ops_driver.cpp compiled against the reference's vendored glm).  Bit patterns, NaN payloads aside."""
    import ctypes as C

    fixture = json.load(open(os.path.join(golden_dir, "glm_ops.json")))
    lib = C.CDLL(os.path.join(os.path.dirname(golden_dir), os.pardir, "oracle", "libgsref.so"))
    fp = C.POINTER(C.c_float)
    assert len(fixture["cases"]) == len(fixture["signatures"]) >= 22
    checked = 0
    for name, (ins, out) in fixture["signatures"].items():
        fn = getattr(lib, "gsref_glm_" + name)
        fn.argtypes = [fp if isinstance(k, int) else (C.c_float if k == "f" else C.c_int) for k in ins] + ([fp] if isinstance(out, int) else [])
        fn.restype = None if isinstance(out, int) else (C.c_float if out == "f" else C.c_int)
        rows = fixture["cases"][name]
        assert len(rows) >= 10, name
        for row in rows:
            args, keep = [], []
            for kind, value in zip(ins, row["in"]):
                if kind == "i":
                    args.append(int(value))
                    continue
                arr = np.array(value, np.uint32).view(np.float32)
                keep.append(arr)
                args.append(arr.ctypes.data_as(fp) if isinstance(kind, int) else C.c_float(arr[0]))
            if isinstance(out, int):
                res = np.zeros(out, np.float32)
                fn(*args, res.ctypes.data_as(fp))
                got = res.view(np.uint32)
            elif out == "f":
                got = np.array([fn(*args)], np.float32).view(np.uint32)
            else:
                assert fn(*args) == row["out"], (name, row)
                checked += 1
                continue
            want = np.array(row["out"], np.uint32)
            nan_both = np.isnan(got.view(np.float32)) & np.isnan(want.view(np.float32))  # sign/payload of a NaN is not pinned
            assert np.array_equal(got[~nan_both], want[~nan_both]), (name, row, got.tolist())
            checked += 1
    assert checked >= 1800


@pytest.mark.parametrize("degree,threads", [(0, 3), (3, 8)])
def test_threaded_oracle_frame_is_the_serial_frame(gsref, tmp_path, degree, threads):
    """gsref_render_threads (the all-core CPU baseline of bench.py) only slices the serial stages' inputs: image, pair count, consumed
    entries and non-empty tiles are the serial frame's, bit for bit -- ragged splat count (not a multiple of the 256-splat SH group),
    1080p-shaped tiles (half-height last tile row)."""
    import cudagaussianrenderer_amd as gs
    from cudagaussianrenderer_amd import scenes

    w, h = 480, 270
    path = str(tmp_path / "s.ply")
    scenes.write_synthetic_ply(path, 20_011, sh_degree=degree, seed=5, log_scale_mean=-3.0)
    ply = gsref.parse_ply(path)
    sh = gs.realign_spherical_harmonics(ply["sh"], ply["shCount"], ply["vertexCount"]) if degree else None
    cam = gsref.camera_from_bounds(ply["boundsMin"], ply["boundsMax"], w, h)
    args = (ply["position"], ply["scaleAndRotation"], ply["color"], sh, ply["shDegree"], ply["shCount"], cam, w, h)
    want, st, col = gsref.render(*args)
    got, st_t, col_t = gsref.render(*args, threads=threads)
    assert want[..., :3].any() and st.tileListSize > 20_000
    assert np.array_equal(got, want) and np.array_equal(col_t, col)
    assert (st_t.tileListSize, st_t.consumedPairs, st_t.nonEmptyTiles) == (st.tileListSize, st.consumedPairs, st.nonEmptyTiles)


def test_camera_controls_blocks_match_the_reference_bit_for_bit(gsref, golden_dir):
    """SURVEY.md 8(f4).  CameraControls::update() polls GLFW and cannot run here; everything it is made of can, and did, through the
    reference's own code compiled where it lies (oracle/upstream_driver.cpp up_cc_*): view / projection / viewProjection at arbitrary
    poses (un-normalised rotations included), the private getWorldSpaceRay, removeRoll, rayIntersectsPlane (the exactly-parallel ray
    included), getPlane, projectOnPlane (the reference's vec - scalar), the whole state setBounds leaves.  The oracle's restatements
    give the same bits (tests/golden/camera_controls.json, written by tests/golden/make_golden.py)."""
    import ctypes as C

    fx = json.load(open(os.path.join(golden_dir, "camera_controls.json")))
    lib = gsref.lib()
    fp = C.POINTER(C.c_float)

    def f(bits):
        return np.array(bits, np.uint32).view(np.float32).copy()

    def same(got, want_bits, what):
        want = np.array(want_bits, np.uint32)
        g = np.atleast_1d(np.asarray(got, np.float32)).view(np.uint32)
        nan_both = np.isnan(g.view(np.float32)) & np.isnan(want.view(np.float32))
        assert np.array_equal(g[~nan_both], want[~nan_both]), (what, g.tolist(), want.tolist())

    P = lambda a: a.ctypes.data_as(fp)  # noqa: E731
    for row in fx["matrices"]:
        pos, q, scr = [f(b) for b in row["in"]]
        v, p, vp = np.zeros(16, np.float32), np.zeros(16, np.float32), np.zeros(16, np.float32)
        lib.gsref_cc_matrices(P(pos), P(q), P(scr), P(v), P(p), P(vp))
        for got, want in zip((v, p, vp), row["out"]):
            same(got, want, "matrices")
    for row in fx["world_ray"]:
        pos, q, scr, ptr = [f(b) for b in row["in"]]
        o, d = np.zeros(3, np.float32), np.zeros(3, np.float32)
        lib.gsref_cc_world_ray(P(pos), P(q), P(scr), P(ptr), P(o), P(d))
        same(o, row["out"][0], "ray origin")
        same(d, row["out"][1], "ray direction")
    for row in fx["remove_roll"]:
        q, r = f(row["in"][0]), np.zeros(4, np.float32)
        lib.gsref_cc_remove_roll(P(q), P(r))
        same(r, row["out"][0], "remove_roll")
    lib.gsref_cc_ray_plane.restype = C.c_int
    misses = 0
    for row in fx["ray_plane"]:
        o, d, plane = [f(b) for b in row["in"]]
        t = C.c_float(-123.0)
        hit = lib.gsref_cc_ray_plane(P(o), P(d), P(plane), C.byref(t))
        assert hit == row["out"][0]
        misses += hit == 0
        same(np.float32(t.value), row["out"][1], "ray_plane t")  # untouched (-123) on a miss, as in the reference
    assert misses >= 5
    for row in fx["get_plane"]:
        a, b, r = f(row["in"][0]), f(row["in"][1]), np.zeros(4, np.float32)
        lib.gsref_cc_get_plane(P(a), P(b), P(r))
        same(r, row["out"][0], "get_plane")
    for row in fx["project_on_plane"]:
        a, b, r = f(row["in"][0]), f(row["in"][1]), np.zeros(3, np.float32)
        lib.gsref_cc_project_on_plane(P(a), P(b), P(r))
        same(r, row["out"][0], "project_on_plane")
    for row in fx["set_bounds_state"]:
        a, b, scr = [f(x) for x in row["in"]]
        pos, q, anchor, floor, speed = np.zeros(3, np.float32), np.zeros(4, np.float32), np.zeros(3, np.float32), np.zeros(4, np.float32), C.c_float(0)
        lib.gsref_cc_set_bounds_state(P(a), P(b), P(scr), P(pos), P(q), P(anchor), P(floor), C.byref(speed))
        for got, want in zip((pos, q, anchor, floor, np.float32(speed.value)), row["out"]):
            same(got, want, "set_bounds")
    assert len(fx["matrices"]) >= 60 and len(fx["world_ray"]) >= 80 and len(fx["remove_roll"]) >= 80


def test_camera_controls_sessions_of_the_oracle_are_the_recorded_ones(gsref, golden_dir):
    """The recorded sessions (every mode, button changes incl. two buttons at once, all six keys, cursor outside the window) give the
    recorded states: a regression record of the oracle's update() restatement (CameraControls.cpp:148-253), whose blocks the test
    above pins.  Also sanity: orbit keeps the distance to the anchor, drag keeps the position, the rotation stays a unit quaternion
    with no roll (camera right axis horizontal)."""
    fx = json.load(open(os.path.join(golden_dir, "camera_controls.json")))
    assert len(fx["sessions"]) == 3
    for sess in fx["sessions"]:
        states = gsref.camera_controls_session(sess["width"], sess["height"], sess["boundsMin"], sess["boundsMax"], sess["frames"])
        want = np.array(sess["states"], np.uint32)
        assert np.array_equal(states.view(np.uint32), want)
        modes = states[1:, 15].astype(int)
        assert set(modes) == {0, 1, 2, 3}
        q = states[:, 3:7].astype(np.float64)
        assert np.abs(np.linalg.norm(q, axis=1) - 1.0).max() < 1e-5
        x, y, z, w = q.T
        right_y = 2 * (x * y + w * z)  # y component of the camera's right axis
        # (not exactly: the reference's projectOnPlane subtracts a scalar from every component, CameraControls.cpp:20-23, so the basis
        # removeRoll rebuilds is only nearly orthonormal; observed 6e-5)
        assert np.abs(right_y).max() < 1e-3
        frames = sess["frames"]
        for i, fr in enumerate(frames):
            before, after = states[i].astype(np.float64), states[i + 1].astype(np.float64)
            if after[15] == 2 and before[15] == 2 and fr[2] == 0:  # orbit, no keys: distance to the anchor kept
                assert abs(np.linalg.norm(after[:3] - after[7:10]) - np.linalg.norm(before[:3] - before[7:10])) < 1e-4
            if after[15] == 3 and fr[2] == 0:  # drag, no keys: position kept
                assert np.array_equal(after[:3], before[:3])
