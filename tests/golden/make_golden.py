#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REFERENCE itself (authoring container only).

Sources of truth used here, none of which is part of the repository's history (no reference source is copied):
  * oracle/_ref/libgsupstream.so -- the reference's own src/PlyParser.cpp and src/CameraControls.cpp
    compiled from /root/reference by oracle/Makefile (driver: oracle/upstream_driver.cpp).  The built .so is
    git-ignored; like every in-tree .so it is in the snapshot gpurun sends to a GPU box, where nothing loads it
    (the fixtures below are what the tests use);
  * /root/reference/sh_gen.py    -- the reference's SH code generator, imported and run; the closed forms
    it prints ("l=.., m=.. -> expr") are evaluated numerically with sympy.

Outputs (data only -- inputs and expected outputs):
  tests/golden/ply_l3_64.ply, ply_l0_32.ply            small synthetic 3DGS files (inputs)
  tests/golden/ply_l3_64.npz, ply_l0_32.npz            what the reference's parsePly returns for them
  tests/golden/ply_errors.json                          malformed headers -> the reference's exception text
  tests/golden/encode_vec4.json                         encodeVec4 known answers
  tests/golden/camera_default_pose.json                 CameraControls::setBounds poses + CameraData constants
  tests/golden/sh_basis.json                            Y_lm, l<=4, at sample directions
  tests/golden/glm_ops.json                             the glm 1.0.1 primitives the clip / tile-list / key code calls (GaussianRender.cu:215-320,
                                                        351-436,452,537-551), evaluated by glm itself (oracle/_ref/libglmops.so = oracle/glm_ops_driver.cpp
                                                        against the reference's vendored headers) on seeded inputs; float bit patterns in and out;
                                                        plus the glm calls CameraControls::update() adds (angleAxis, quat*quat, quat*vec3, quat(mat3), cross,
                                                        dot, mat4*mat4, inverse(mat4))
  tests/golden/camera_controls.json                     SURVEY.md 8(f4): what CameraControls::update() is made of, evaluated by the REFERENCE's own code
                                                        (up_cc_* of oracle/upstream_driver.cpp: view/projection at arbitrary poses, getWorldSpaceRay, removeRoll,
                                                        rayIntersectsPlane, getPlane, projectOnPlane, the state setBounds leaves), float bit patterns; and
                                                        recorded input sessions with the poses the ORACLE's restatement of update() gives for them (update()
                                                        itself polls GLFW and cannot run here: sessions are a regression record, not a reference output)

Run:  make -C oracle && python tests/golden/make_golden.py
"""
import contextlib
import ctypes as C
import io
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from cudagaussianrenderer_amd import scenes  # noqa: E402  (numpy-only input synthesis)

REF = "/root/reference"
UPSTREAM = os.path.join(ROOT, "oracle", "_ref", "libgsupstream.so")


class UpPly(C.Structure):
    _fields_ = [("vertexCount", C.c_int32), ("shDegree", C.c_int32), ("shCount", C.c_int32),
                ("boundsMin", C.c_float * 3), ("boundsMax", C.c_float * 3),
                ("position4", C.POINTER(C.c_float)), ("scaleAndRotation4", C.POINTER(C.c_float)),
                ("color4", C.POINTER(C.c_float)), ("sh", C.POINTER(C.c_float))]


def load_upstream():
    # RTLD_LAZY: CameraControls::update() references glfw input functions that are never called here.
    return C.CDLL(UPSTREAM, mode=os.RTLD_LAZY)


def upstream_parse(up, path):
    ply = UpPly()
    err = C.create_string_buffer(512)
    n = up.up_parse_ply(os.fsencode(path), C.byref(ply), err, C.c_size_t(512))
    if n < 0:
        return None, err.value.decode()
    def arr(ptr, count):
        return np.ctypeslib.as_array(ptr, shape=(count,)).copy() if count else np.zeros(0, np.float32)
    out = dict(vertexCount=n, shDegree=ply.shDegree, shCount=ply.shCount,
               boundsMin=np.array(ply.boundsMin[:], np.float32), boundsMax=np.array(ply.boundsMax[:], np.float32),
               position=arr(ply.position4, 4 * n).reshape(n, 4),
               scaleAndRotation=arr(ply.scaleAndRotation4, 4 * n).reshape(n, 4),
               color=arr(ply.color4, 4 * n).reshape(n, 4),
               sh=arr(ply.sh, ply.shCount * n).reshape(n, ply.shCount) if ply.shDegree else np.zeros((n, 0), np.float32))
    up.up_free_ply(C.byref(ply))
    return out, None


def make_ply_fixtures(up):
    # Degree 3, canonical 62-property order.
    names, table = scenes.synthetic_vertex_table(64, sh_degree=3, seed=7)
    p3 = os.path.join(HERE, "ply_l3_64.ply")
    scenes.write_ply(p3, names, table)
    out, err = upstream_parse(up, p3)
    assert err is None, err
    np.savez_compressed(os.path.join(HERE, "ply_l3_64.npz"), **out)

    # Degree 0, shuffled column order, extra unknown columns (nx..nz stay), some extreme raw values.
    names0, table0 = scenes.synthetic_vertex_table(32, sh_degree=0, seed=11)
    table0[0, names0.index("opacity")] = 40.0      # sigmoid -> 1
    table0[1, names0.index("opacity")] = -40.0     # sigmoid -> ~0
    table0[2, names0.index("rot_0"):names0.index("rot_0") + 4] = [0, 0, 0, 0]  # zero quaternion -> identity
    table0[3, names0.index("rot_0"):names0.index("rot_0") + 4] = [2, 0, 0, 0]  # un-normalised identity
    perm = np.random.default_rng(3).permutation(len(names0))
    names0p = [names0[i] for i in perm]
    p0 = os.path.join(HERE, "ply_l0_32.ply")
    scenes.write_ply(p0, names0p, table0[:, perm])
    out0, err = upstream_parse(up, p0)
    assert err is None, err
    np.savez_compressed(os.path.join(HERE, "ply_l0_32.npz"), **out0)

    # Malformed headers: every throw site of PlyParser.cpp that a header can reach.
    base_props = "".join(f"property float {n}\n" for n in scenes.ply_property_names(0))
    def hdr(body):
        return ("ply\n" + body).encode("ascii")
    cases = {
        "empty_line": hdr("format binary_little_endian 1.0\n\nelement vertex 1\n" + base_props + "end_header\n"),
        "element_not_vertex": hdr("format binary_little_endian 1.0\nelement face 1\n" + base_props + "end_header\n"),
        "element_no_count": hdr("format binary_little_endian 1.0\nelement vertex\n" + base_props + "end_header\n"),
        "property_not_float": hdr("format binary_little_endian 1.0\nelement vertex 1\nproperty double x\nend_header\n"),
        "property_no_name": hdr("format binary_little_endian 1.0\nelement vertex 1\nproperty float\nend_header\n"),
        "duplicate_property": hdr("format binary_little_endian 1.0\nelement vertex 1\nproperty float x\nproperty float x\nend_header\n"),
        "ascii_format": hdr("format ascii 1.0\nelement vertex 1\n" + base_props + "end_header\n"),
        "no_vertex_count": hdr("format binary_little_endian 1.0\n" + base_props + "end_header\n"),
        "header_too_long": hdr("format binary_little_endian 1.0\nelement vertex 1\n" +
                               "".join(f"property float p{i}\n" for i in range(300)) + "end_header\n"),
        "missing_opacity": hdr("format binary_little_endian 1.0\nelement vertex 1\n" +
                               base_props.replace("property float opacity\n", "") + "end_header\n"),
        "bad_sh_count": hdr("format binary_little_endian 1.0\nelement vertex 1\n" + base_props +
                            "".join(f"property float f_rest_{i}\n" for i in range(10)) + "end_header\n"),
    }
    errors = {}
    tmp = os.path.join(HERE, "_tmp_bad.ply")
    for name, header in cases.items():
        with open(tmp, "wb") as f:
            f.write(header + b"\0" * 4 * 128)
        out, err = upstream_parse(up, tmp)
        assert out is None, name
        errors[name] = {"header": header.decode("ascii"), "error": err}
    os.remove(tmp)
    with open(os.path.join(HERE, "ply_errors.json"), "w") as f:
        json.dump(errors, f, indent=1)


def make_encode_fixture(up):
    up.up_encode_vec4.restype = C.c_uint32
    rng = np.random.default_rng(5)
    vecs = [[0.5, 0.5, 0.5, 1.0], [0, 0, 0, 0], [1, 1, 1, 1], [-1, 2, 0.25, 0.999999], [0.99607843, 0.00392157, 0.5019608, 0.49803922]]
    vecs += rng.uniform(-0.1, 1.1, size=(40, 4)).round(6).tolist()
    rows = []
    for v in vecs:
        a = np.array(v, np.float32)
        rows.append({"v": [float(x) for x in a], "bits": int(up.up_encode_vec4(a.ctypes.data_as(C.POINTER(C.c_float))))})
    with open(os.path.join(HERE, "encode_vec4.json"), "w") as f:
        json.dump(rows, f, indent=1)


def make_camera_fixture(up):
    cases = []
    for (bmin, bmax) in [([-4, -4, -4], [4, 4, 4]), ([-1.5, 0.25, -3.0], [2.5, 1.75, 6.0]), ([10, 10, 10], [11, 12, 10.5])]:
        for (w, h) in [(1024, 1024), (1920, 1080), (256, 256), (3840, 2160)]:
            a = np.array(bmin, np.float32)
            b = np.array(bmax, np.float32)
            pos = np.zeros(3, np.float32)
            view = np.zeros(16, np.float32)
            proj = np.zeros(16, np.float32)
            vp = np.zeros(16, np.float32)
            params = np.zeros(4, np.float32)
            fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
            up.up_camera_default(fp(a), fp(b), C.c_float(w), C.c_float(h), fp(pos), fp(view), fp(proj), fp(vp), fp(params))
            aspect, fov, near, far = [float(x) for x in params]
            # Demo.cpp:383-392 (the frame loop, not CameraControls) -- evaluated here in float32 like the reference.
            cot_y = np.float32(1.0) / np.tan(np.float32(fov) * np.float32(0.5), dtype=np.float32)
            cot_x = cot_y / np.float32(aspect)
            scale_z = np.float32(-2.0) / (np.float32(far) - np.float32(near))
            bias_z = -(np.float32(far) + np.float32(near)) / (np.float32(far) - np.float32(near))
            cases.append({"boundsMin": bmin, "boundsMax": bmax, "width": w, "height": h,
                          "position": pos.tolist(), "view": view.tolist(), "projection": proj.tolist(),
                          "viewProjection": vp.tolist(), "aspect": aspect, "fov": fov, "near": near, "far": far,
                          "fovCotangent": [float(cot_x), float(cot_y)], "depthScaleBias": [float(scale_z), float(bias_z)]})
    with open(os.path.join(HERE, "camera_default_pose.json"), "w") as f:
        json.dump(cases, f, indent=1)


def make_sh_fixture():
    import sympy

    sys.path.insert(0, REF)
    import sh_gen  # the reference's generator (imported, not copied)

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sh_gen.main()
    x, y, z = sympy.symbols("x y z", real=True)
    r = sympy.symbols("r", real=True, positive=True)
    forms = {}
    for line in buf.getvalue().splitlines():
        m = re.match(r"l=(\d+), m=(-?\d+)\s*->\s*(.*)$", line)
        if m:
            forms[(int(m.group(1)), int(m.group(2)))] = sympy.sympify(m.group(3), locals={"x": x, "y": y, "z": z, "r": r})
    assert len(forms) == 25, len(forms)
    rng = np.random.default_rng(9)
    dirs = rng.standard_normal((24, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dirs = np.vstack([dirs, np.eye(3), -np.eye(3), [[0.6, 0.8, 0.0]]])
    rows = []
    for d in dirs:
        vals = []
        for l in range(5):
            for m in range(-l, l + 1):  # same order as the generated code walks its coefficients
                vals.append(float(forms[(l, m)].subs({x: d[0], y: d[1], z: d[2], r: 1}).evalf(20)))
        rows.append({"dir": [float(v) for v in d], "basis": vals})
    with open(os.path.join(HERE, "sh_basis.json"), "w") as f:
        json.dump({"order": "k = l*l + (m + l), l = 0..4", "samples": rows}, f, indent=1)


# name -> (input kinds, output kind); a kind is a float-array length, "f" (float scalar) or "i" (int scalar).  tests/test_oracle_golden.py
# binds the oracle's gsref_glm_<name> hooks with the same table (stored in the fixture).
GLM_OPS = {
    "mat3_cast": ([4], 9), "mat3_mul": ([9, 9], 9), "mat3_transpose": ([9], 9), "mat3_sandwich": ([9, 9], 9),
    "mat4_mul_vec4": ([16, 4], 4), "mat3_of_mat4": ([16], 9), "normalize2": ([2], 2), "normalize3": ([3], 3),
    "atan2": (["f", "f"], "f"), "max": (["f", "f"], "f"), "min": (["f", "f"], "f"), "step": (["f", "f"], "f"),
    "mix": (["f", "f", "f"], "f"), "clamp01": (["f"], "f"), "clampi": (["i", "i", "i"], "i"), "length2": ([2], "f"),
    "dot2": ([2, 2], "f"), "sqrt": (["f"], "f"), "cos": (["f"], "f"), "sin": (["f"], "f"), "floor": (["f"], "f"), "ceil": (["f"], "f"),
    # CameraControls::update(); quaternions as (x, y, z, w)
    "angle_axis": (["f", 3], 4), "quat_mul": ([4, 4], 4), "quat_rotate": ([4, 3], 3), "quat_cast": ([9], 4), "cross": ([3, 3], 3),
    "dot3": ([3, 3], "f"), "mat4_mul": ([16, 16], 16), "mat4_inverse": ([16], 16),
}


def bind_glm_ops(lib, prefix):
    """ctypes signatures of `<prefix><name>` for every entry of GLM_OPS; returns {name: callable(list of inputs) -> output}."""
    fp = C.POINTER(C.c_float)

    def make(name, ins, out):
        fn = getattr(lib, prefix + name)
        fn.argtypes = [fp if isinstance(k, int) else (C.c_float if k == "f" else C.c_int) for k in ins] + ([fp] if isinstance(out, int) else [])
        fn.restype = None if isinstance(out, int) else (C.c_float if out == "f" else C.c_int)

        def call(args):
            conv = []
            for k, a in zip(ins, args):
                if isinstance(k, int):
                    conv.append(np.ascontiguousarray(a, np.float32))
                else:
                    conv.append(C.c_float(np.float32(a)) if k == "f" else int(a))
            cargs = [a.ctypes.data_as(fp) if isinstance(a, np.ndarray) else a for a in conv]
            if isinstance(out, int):
                res = np.zeros(out, np.float32)
                fn(*cargs, res.ctypes.data_as(fp))
                return res
            r = fn(*cargs)
            return np.float32(r) if out == "f" else int(r)
        return call
    return {name: make(name, ins, out) for name, (ins, out) in GLM_OPS.items()}


def glm_op_inputs():
    """Seeded inputs per primitive: the value ranges the pipeline feeds them plus the edge cases it can reach."""
    rng = np.random.default_rng(20251004)
    f32 = np.float32

    def quats(n):
        q = [np.array(b, f32) / f32(255.0) * f32(2.0) - f32(1.0) for b in rng.integers(0, 256, size=(n, 4))]  # decodeVec4(...)*2-1: not normalised
        q += [np.array(v, f32) for v in ([0, 0, 0, 1], [1, 1, 1, -1], [-1, -1, -1, -1], [0, 0, 0, 0], [2, 0, 0, 0], [1e-3, 0, 0, 1e-3])]
        q += list(rng.standard_normal((20, 4)).astype(f32))
        return q

    def mats(n, k, scale=1.0):
        return [(rng.standard_normal(k) * scale).astype(f32) for _ in range(n)]
    diag = []
    for s in rng.uniform(0.005, 0.5, size=(30, 3)):
        m = np.zeros(9, f32)
        m[0], m[4], m[8] = s
        diag.append(m)
    cov = []
    for a in mats(40, 9, 0.3):
        cov.append((a.reshape(3, 3) @ a.reshape(3, 3).T).astype(f32).reshape(9))
    vecs2 = [np.array(v, f32) for v in ([0, 0], [0, 1e-20], [1e-30, 1e-30], [3e19, 4e19], [-0.0, 0.0], [1, 0], [0, -1], [1e-12, 0], [0, 2.5e-7])]
    vecs2 += list((rng.standard_normal((60, 2)) * 10.0 ** rng.uniform(-9, 2, size=(60, 1))).astype(f32))
    scal = [f32(v) for v in (0, -0.0, 1, -1, 0.5, 1e-12, -1e-12, 1e12, -1e12, np.nan, np.inf, -np.inf, 128, -128, 0.99999994, 1.0000001, -1.0000001)]
    scal += list(rng.standard_normal(40).astype(f32)) + list((rng.standard_normal(20) * 100).astype(f32))
    pairs = [(a, b) for a in scal[:17] for b in scal[:17]][::3] + list(zip(scal[17:], scal[:17:-1]))
    angles = [f32(v) for v in (0, -0.0, np.pi, -np.pi, np.pi / 2, -np.pi / 2, 3.1415927, 1e-8, 100.0, -1e4)] + list(rng.uniform(-np.pi, np.pi, 60).astype(f32))
    unit = [np.array([np.cos(t), np.sin(t)], f32) for t in rng.uniform(-np.pi, np.pi, 60)]
    ops = {
        "mat3_cast": [[q] for q in quats(120)],
        "mat3_mul": [[a, b] for a, b in zip(mats(60, 9), mats(60, 9))] + [[a, d] for a, d in zip(mats(30, 9), diag)],
        "mat3_transpose": [[a] for a in mats(20, 9)],
        "mat3_sandwich": [[a, c] for a, c in zip(mats(40, 9, 0.5), cov)],
        "mat4_mul_vec4": [[m, np.append(v, f32(1.0))] for m, v in zip(mats(60, 16), mats(60, 3, 4.0))],
        "mat3_of_mat4": [[m] for m in mats(10, 16)],
        "normalize2": [[v] for v in vecs2],
        "normalize3": [[v] for v in mats(40, 3, 5.0)] + [[np.zeros(3, f32)]],
        "atan2": [[v[1], v[0]] for v in unit] + [[a, b] for a, b in pairs[:60]],
        "max": [list(p) for p in pairs], "min": [list(p) for p in pairs], "step": [list(p) for p in pairs],
        "mix": [[f32(-128.0), x, a] for x in scal[:30] for a in (f32(0.0), f32(1.0))] + [[a, b, f32(t)] for (a, b), t in zip(pairs[:40], rng.uniform(0, 1, 40))],
        "clamp01": [[x] for x in scal] + [[f32(v)] for v in rng.uniform(-0.5, 1.5, 40)],
        "clampi": [[int(x), 0, int(hi)] for x in rng.integers(-300, 400, 60) for hi in (64, 120)][::2] + [[-2147483648, 0, 68], [2147483647, 0, 240]],
        "length2": [[v] for v in vecs2], "dot2": [[a, b] for a, b in zip(vecs2, vecs2[::-1])],
        "sqrt": [[f32(abs(float(x)))] for x in scal if np.isfinite(x)] + [[f32(1e-12)], [f32(7.0)]],
        "cos": [[a] for a in angles], "sin": [[a] for a in angles],
        "floor": [[f32(v)] for v in rng.uniform(-3, 130, 50)] + [[f32(-0.0)], [f32(63.999996)], [f32(np.nan)], [f32(64.0)]],
        "ceil": [[f32(v)] for v in rng.uniform(-3, 130, 50)] + [[f32(-0.5)], [f32(64.000004)], [f32(np.nan)], [f32(64.0)]],
    }
    # ---- CameraControls::update(): drawn AFTER everything above, so the cases above keep their values
    def unit_quats(n):
        q = rng.standard_normal((n, 4))
        return list((q / np.linalg.norm(q, axis=1, keepdims=True)).astype(f32))

    def rotations(n):  # rotation matrices (columns), as floats: what removeRoll hands to quat(mat3), all four branches of quat_cast
        out = []
        for q in unit_quats(n):
            x, y, z, w = [float(v) for v in q]
            out.append(np.array([1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y), 2 * (x * y - w * z), 1 - 2 * (x * x + z * z),
                                 2 * (y * z + w * x), 2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)], f32))
        return out
    axes = [np.array(v, f32) for v in ([0, 1, 0], [1, 0, 0], [0, 0, 1])] + [(v / np.linalg.norm(v)).astype(f32) for v in rng.standard_normal((40, 3))]
    small = [f32(v) for v in (0, -0.0, 1e-8, 0.5, -0.5, np.pi, -np.pi)] + list((rng.standard_normal(50) * 0.05).astype(f32))
    rigid = []
    for q, t in zip(unit_quats(40), rng.standard_normal((40, 3)) * 8):
        x, y, z, w = [float(v) for v in q]
        m = np.eye(4)
        m[:3, 0] = [1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)]
        m[:3, 1] = [2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)]
        m[:3, 2] = [2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)]
        m[:3, 3] = t
        rigid.append(m.T.astype(f32).reshape(16))  # column-major
    persp = []
    for aspect in (1.0, 16 / 9, 4 / 3, 0.5):
        t = np.tan(np.radians(60.0) / 2)
        m = np.zeros((4, 4))
        m[0, 0], m[1, 1], m[2, 2], m[3, 2], m[2, 3] = 1 / (aspect * t), 1 / t, -(100 + 0.1) / (100 - 0.1), -1.0, -(2 * 100 * 0.1) / (100 - 0.1)
        persp.append(m.T.astype(f32).reshape(16))
    vp = [(p.reshape(4, 4).T.astype(np.float64) @ np.linalg.inv(r.reshape(4, 4).T.astype(np.float64))).T.astype(f32).reshape(16) for r in rigid[:20] for p in persp[:2]]
    ops.update({
        "angle_axis": [[a, ax] for a, ax in zip(small, axes + axes)][:50],
        "quat_mul": [[a, b] for a, b in zip(unit_quats(50), unit_quats(50))] + [[a, b] for a, b in zip(mats(10, 4), mats(10, 4))],
        "quat_rotate": [[q, v] for q, v in zip(unit_quats(50), mats(50, 3, 6.0))] + [[q, np.array(v, f32)] for q in unit_quats(6) for v in ([1, 0, 0], [0, 0, 1])],
        "quat_cast": [[m] for m in rotations(80)] + [[np.array(v, f32)] for v in ([1, 0, 0, 0, 1, 0, 0, 0, 1], [1, 0, 0, 0, -1, 0, 0, 0, -1], [-1, 0, 0, 0, 1, 0, 0, 0, -1], [-1, 0, 0, 0, -1, 0, 0, 0, 1])],
        "cross": [[a, b] for a, b in zip(mats(40, 3, 3.0), mats(40, 3, 3.0))] + [[axes[0], axes[1]], [axes[1], axes[1]]],
        "dot3": [[a, b] for a, b in zip(mats(40, 3, 3.0), mats(40, 3, 3.0))] + [[axes[0], m] for m in mats(10, 3)],
        "mat4_mul": [[a, b] for a, b in zip(mats(20, 16), mats(20, 16))] + [[p, r] for p, r in zip(persp * 5, rigid[:20])] + [[r, q] for r, q in zip(rigid[:10], rigid[10:20])],
        "mat4_inverse": [[m] for m in rigid] + [[m] for m in vp] + [[m] for m in mats(20, 16)],
    })
    return ops


def make_glm_ops_fixture():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libglmops.so"))
    ops = bind_glm_ops(lib, "glmops_")

    def bits(v):
        if isinstance(v, (int, np.integer)) and not isinstance(v, np.floating):
            return int(v)
        return [int(b) for b in np.atleast_1d(np.asarray(v, np.float32)).view(np.uint32)]
    out = {"what": "glm 1.0.1 primitives evaluated by glm itself (oracle/glm_ops_driver.cpp, g++ -O2 -ffp-contract=off, scalar code path); "
                   "floats are IEEE-754 bit patterns (uint32), ints are plain", "signatures": {k: [v[0], v[1]] for k, v in GLM_OPS.items()}, "cases": {}}
    total = 0
    for name, cases in glm_op_inputs().items():
        rows = []
        for args in cases:
            rows.append({"in": [bits(a) for a in args], "out": bits(ops[name](args))})
        out["cases"][name] = rows
        total += len(rows)
    with open(os.path.join(HERE, "glm_ops.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"glm_ops.json: {total} cases over {len(out['cases'])} primitives")


def _bits(v):
    if isinstance(v, (int, np.integer)) and not isinstance(v, np.floating):
        return int(v)
    return [int(b) for b in np.atleast_1d(np.asarray(v, np.float32)).view(np.uint32)]


def camera_sessions():
    """Recorded inputs: (dt, mouseButtons, keys, cursorX, cursorY) per frame; every mode, button changes, keys, the first-frame delta."""
    rng = np.random.default_rng(424242)
    sessions = []
    for (w, h, bmin, bmax) in [(1920, 1080, [-4, -4, -4], [4, 4, 4]), (1024, 1024, [-1.5, 0.25, -3.0], [2.5, 1.75, 6.0]), (640, 480, [10, 10, 10], [11, 12, 10.5])]:
        frames = []
        x, y = w * 0.5, h * 0.5
        for seg in range(14):
            buttons = [0, 1, 2, 4, 0, 4, 2, 1, 3, 6, 0, 2, 4, 1][seg]
            keys = [0, 0, 1, 0, 8 | 16, 0, 2, 4 | 32, 0, 0, 63, 0, 1 | 8, 0][seg]
            vx, vy = rng.uniform(-14, 14, 2)
            for _ in range(int(rng.integers(3, 9))):
                x, y = x + vx + rng.uniform(-2, 2), y + vy + rng.uniform(-2, 2)
                frames.append([float(np.float32(rng.uniform(0.004, 0.03))), buttons, keys, float(np.round(x * 4) / 4), float(np.round(y * 4) / 4)])
        sessions.append({"width": w, "height": h, "boundsMin": bmin, "boundsMax": bmax, "frames": frames})
    return sessions


def make_camera_controls_fixture(up):
    sys.path.insert(0, ROOT)
    fp = C.POINTER(C.c_float)
    rng = np.random.default_rng(777)
    f32 = np.float32
    P = lambda a: a.ctypes.data_as(fp)  # noqa: E731

    def unit_quat():
        q = rng.standard_normal(4)
        return (q / np.linalg.norm(q)).astype(f32)
    screens = [np.array(v, f32) for v in ([1920, 1080], [1024, 1024], [640, 480], [3840, 2160])]
    out = {"what": "what CameraControls::update() is made of, evaluated by the reference's own code (oracle/upstream_driver.cpp up_cc_*, "
                   "CameraControls.cpp compiled where it lies); float bit patterns (uint32). 'sessions' are input recordings with the poses the "
                   "ORACLE's restatement of update() gives (oracle/gsref.cpp gsref_cc_update): a regression record, not a reference output"}
    rows = []
    for i in range(60):
        pos, q, scr = (rng.standard_normal(3) * 9).astype(f32), unit_quat() if i % 5 else rng.standard_normal(4).astype(f32), screens[i % 4]
        v, p, vp = np.zeros(16, f32), np.zeros(16, f32), np.zeros(16, f32)
        up.up_cc_matrices(P(pos), P(q), P(scr), P(v), P(p), P(vp))
        rows.append({"in": [_bits(pos), _bits(q), _bits(scr)], "out": [_bits(v), _bits(p), _bits(vp)]})
    out["matrices"] = rows
    rows = []
    for i in range(80):
        pos, q, scr = (rng.standard_normal(3) * 9).astype(f32), unit_quat(), screens[i % 4]
        ptr = (rng.uniform(-0.2, 1.2, 2) * scr).astype(f32)
        o, d = np.zeros(3, f32), np.zeros(3, f32)
        up.up_cc_world_ray(P(pos), P(q), P(scr), P(ptr), P(o), P(d))
        rows.append({"in": [_bits(pos), _bits(q), _bits(scr), _bits(ptr)], "out": [_bits(o), _bits(d)]})
    out["world_ray"] = rows
    rows = []
    for i in range(80):
        q = unit_quat() if i % 4 else rng.standard_normal(4).astype(f32)
        r = np.zeros(4, f32)
        up.up_cc_remove_roll(P(q), P(r))
        rows.append({"in": [_bits(q)], "out": [_bits(r)]})
    out["remove_roll"] = rows
    rows = []
    up.up_cc_ray_plane.restype = C.c_int
    for i in range(60):
        o, d = (rng.standard_normal(3) * 5).astype(f32), rng.standard_normal(3).astype(f32)
        d = (d / np.linalg.norm(d)).astype(f32)
        plane = np.append(rng.standard_normal(3), rng.standard_normal() * 4).astype(f32)
        if i % 10 == 0:  # a ray inside the plane: denominator exactly 0
            plane = np.array([0, 1, 0, -1.5], f32)
            d = np.array([1, 0, 0], f32)
        t = C.c_float(-123.0)
        hit = up.up_cc_ray_plane(P(o), P(d), P(plane), C.byref(t))
        rows.append({"in": [_bits(o), _bits(d), _bits(plane)], "out": [int(hit), _bits(f32(t.value))]})
    out["ray_plane"] = rows
    rows, rows2 = [], []
    for i in range(40):
        a, b = rng.standard_normal(3).astype(f32), (rng.standard_normal(3) * 4).astype(f32)
        r4, r3 = np.zeros(4, f32), np.zeros(3, f32)
        up.up_cc_get_plane(P(a), P(b), P(r4))
        up.up_cc_project_on_plane(P(b), P(a), P(r3))
        rows.append({"in": [_bits(a), _bits(b)], "out": [_bits(r4)]})
        rows2.append({"in": [_bits(b), _bits(a)], "out": [_bits(r3)]})
    out["get_plane"], out["project_on_plane"] = rows, rows2
    rows = []
    for (bmin, bmax) in [([-4, -4, -4], [4, 4, 4]), ([-1.5, 0.25, -3.0], [2.5, 1.75, 6.0]), ([10, 10, 10], [11, 12, 10.5]), ([-30, -2, -7], [15, 3, 8])]:
        for scr in screens[:2]:
            a, b = np.array(bmin, f32), np.array(bmax, f32)
            pos, q, anchor, floor, speed = np.zeros(3, f32), np.zeros(4, f32), np.zeros(3, f32), np.zeros(4, f32), C.c_float(0)
            up.up_cc_set_bounds(P(a), P(b), P(scr), P(pos), P(q), P(anchor), P(floor), C.byref(speed))
            rows.append({"in": [_bits(a), _bits(b), _bits(scr)], "out": [_bits(pos), _bits(q), _bits(anchor), _bits(floor), _bits(f32(speed.value))]})
    out["set_bounds_state"] = rows

    # sessions through the oracle's restatement of update()
    from oracle import gsref
    sessions = camera_sessions()
    for sess in sessions:
        sess["states"] = [_bits(st) for st in gsref.camera_controls_session(sess["width"], sess["height"], sess["boundsMin"], sess["boundsMax"], sess["frames"])]
    out["sessions"] = sessions
    with open(os.path.join(HERE, "camera_controls.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("camera_controls.json:", {k: len(v) for k, v in out.items() if isinstance(v, list)})


def main():
    if "--only-camera-controls" in sys.argv:
        make_camera_controls_fixture(load_upstream())
        return
    if "--only-glm" in sys.argv:
        make_glm_ops_fixture()
        return
    up = load_upstream()
    make_ply_fixtures(up)
    make_encode_fixture(up)
    make_camera_fixture(up)
    make_camera_controls_fixture(up)
    make_sh_fixture()
    make_glm_ops_fixture()
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
