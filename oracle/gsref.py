"""ctypes binding of the CPU oracle (oracle/libgsref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsref.so")


class CameraData(C.Structure):
    """Field-for-field GaussianRender.cuh:17-26 (224 bytes, glm column-major matrices)."""

    _fields_ = [
        ("viewProjection", C.c_float * 16),
        ("projection", C.c_float * 16),
        ("view", C.c_float * 16),
        ("fovCotangent", C.c_float * 2),
        ("depthScaleBias", C.c_float * 2),
        ("position", C.c_float * 3),
        ("aspect", C.c_float),
    ]


class FrameStats(C.Structure):
    _fields_ = [
        ("msSphericalHarmonics", C.c_double),
        ("msClipData", C.c_double),
        ("msBuildTileList", C.c_double),
        ("msSortTileList", C.c_double),
        ("msTileRanges", C.c_double),
        ("msRasterize", C.c_double),
        ("msTotal", C.c_double),
        ("tileListSize", C.c_int64),
        ("consumedPairs", C.c_int64),
        ("nonEmptyTiles", C.c_int64),
    ]


class _Ply(C.Structure):
    _fields_ = [
        ("vertexCount", C.c_int32),
        ("shDegree", C.c_int32),
        ("shCount", C.c_int32),
        ("boundsMin", C.c_float * 3),
        ("boundsMax", C.c_float * 3),
        ("position4", C.POINTER(C.c_float)),
        ("scaleAndRotation4", C.POINTER(C.c_float)),
        ("color4", C.POINTER(C.c_float)),
        ("sh", C.POINTER(C.c_float)),
    ]


def build(force=False):
    """Compile the oracle (and oracle/_ref when the reference checkout is present)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
        os.path.join(_HERE, "gsref.cpp")
    ):
        subprocess.check_call(["make", "-C", _HERE, "libgsref.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.gsref_encode_vec4.restype = C.c_uint32
        _lib.gsref_get_key.restype = C.c_uint64
        _lib.gsref_get_key.argtypes = [C.c_int32, C.c_float]
        _lib.gsref_realigned_sh_size.restype = C.c_size_t
        _lib.gsref_build_tile_list.restype = C.c_int64
        _lib.gsref_rasterize_tiles.restype = C.c_int64
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def encode_vec4(v):
    a, p = _f(v)
    return int(lib().gsref_encode_vec4(p))


def decode_vec4(bits):
    out = np.zeros(4, np.float32)
    lib().gsref_decode_vec4(C.c_uint32(bits), _p(out, C.c_float))
    return out


def get_key(tile, depth):
    return int(lib().gsref_get_key(int(tile), float(depth)))


def spherical_harmonics(l, direction, sh, stride=1):
    d, dp = _f(direction)
    s, sp = _f(sh)
    out = np.zeros(3, np.float32)
    lib().gsref_spherical_harmonics(int(l), dp, sp, int(stride), _p(out, C.c_float))
    return out


def realign_spherical_harmonics(sh, sh_count, splat_count, group_size=256):
    s, sp = _f(sh)
    n = lib().gsref_realigned_sh_size(group_size, sh_count, splat_count)
    out = np.zeros(n, np.float32)
    lib().gsref_realign_spherical_harmonics(sp, _p(out, C.c_float), group_size, sh_count, splat_count)
    return out


def evaluate_spherical_harmonics(position, sh_aligned, sh_degree, sh_count, camera_position):
    pos, pp = _f(position)
    s, sp = _f(sh_aligned)
    cam, cp = _f(camera_position)
    n = pos.shape[0]
    color = np.zeros((n, 4), np.float32)
    lib().gsref_evaluate_spherical_harmonics(n, pp, sp, int(sh_degree), int(sh_count), cp, _p(color, C.c_float))
    return color


def evaluate_splat_clip_data(position, scale_and_rotation, camera, width, height):
    pos, pp = _f(position)
    sr, srp = _f(scale_and_rotation)
    n = pos.shape[0]
    clip_xy = np.zeros((n, 2), np.float32)
    clip_z = np.zeros(n, np.float32)
    ellipse = np.zeros((n, 4), np.float32)
    conic = np.zeros((n, 4), np.float32)
    lib().gsref_evaluate_splat_clip_data(
        n, pp, srp, C.byref(camera), int(width), int(height), _p(clip_xy, C.c_float), _p(clip_z, C.c_float),
        _p(ellipse, C.c_float), _p(conic, C.c_float))
    return clip_xy, clip_z, ellipse, conic


def build_tile_list(clip_xy, clip_z, ellipse, width, height, rows=(0, 0), capacity=None):
    xy, xyp = _f(clip_xy)
    z, zp = _f(clip_z)
    e, ep = _f(ellipse)
    n = z.shape[0]
    total = lib().gsref_build_tile_list(n, xyp, zp, ep, int(width), int(height), int(rows[0]), int(rows[1]), None, None,
                                        C.c_int64(0))
    cap = total if capacity is None else int(capacity)
    keys = np.zeros(max(cap, 1), np.uint64)
    values = np.zeros(max(cap, 1), np.int32)
    lib().gsref_build_tile_list(n, xyp, zp, ep, int(width), int(height), int(rows[0]), int(rows[1]),
                                _p(keys, C.c_uint64), _p(values, C.c_int32), C.c_int64(cap))
    m = min(total, cap)
    return keys[:m], values[:m], int(total)


def sort_end_bit(width, height):
    return int(lib().gsref_sort_end_bit(int(width), int(height)))


def sort_tile_list(keys, values, end_bit):
    k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
    v = np.ascontiguousarray(values, dtype=np.int32).copy()
    lib().gsref_sort_tile_list(C.c_int64(k.shape[0]), _p(k, C.c_uint64), _p(v, C.c_int32), int(end_bit))
    return k, v


def evaluate_tile_ranges(keys, total_tiles):
    k = np.ascontiguousarray(keys, dtype=np.uint64)
    rng = np.full(2 * total_tiles, -1, np.int32)
    lib().gsref_evaluate_tile_ranges(C.c_int64(k.shape[0]), _p(k, C.c_uint64), _p(rng, C.c_int32))
    return rng


def rasterize_tiles(width, height, tile_range, values, clip_xy, conic, color, rows=(0, 0)):
    rng = np.ascontiguousarray(tile_range, dtype=np.int32)
    v = np.ascontiguousarray(values, dtype=np.int32)
    xy, xyp = _f(clip_xy)
    cn, cnp = _f(conic)
    col, colp = _f(color)
    img = np.zeros((height, width, 4), np.uint8)
    consumed = lib().gsref_rasterize_tiles(int(width), int(height), int(rows[0]), int(rows[1]), _p(rng, C.c_int32),
                                           _p(v, C.c_int32), xyp, cnp, colp, _p(img, C.c_uint8))
    return img, int(consumed)


def set_footprint(epanechnikov):
    """GaussianRender.cu:8-9: the EPANECHNIKOV_KERNEL build option as a switch (False = Gaussian, the default)."""
    lib().gsref_set_footprint(1 if epanechnikov else 0)


def render(position, scale_and_rotation, color, sh_aligned, sh_degree, sh_count, camera, width, height, threads=1):
    """Demo.cpp:394-479 on the CPU. Returns (image HxWx4 uint8, FrameStats, colour actually used).  threads > 1: the same frame, bit
    for bit, with every stage spread over that many host threads (the all-core baseline)."""
    pos, pp = _f(position)
    sr, srp = _f(scale_and_rotation)
    col = np.ascontiguousarray(color, dtype=np.float32).copy()
    if sh_degree != 0:
        s, sp = _f(sh_aligned)
    else:
        sp = None
    img = np.zeros((height, width, 4), np.uint8)
    st = FrameStats()
    if threads > 1:
        rc = lib().gsref_render_threads(pos.shape[0], pp, srp, _p(col, C.c_float), sp, int(sh_degree), int(sh_count),
                                        C.byref(camera), int(width), int(height), _p(img, C.c_uint8), C.byref(st), int(threads))
    else:
        rc = lib().gsref_render(pos.shape[0], pp, srp, _p(col, C.c_float), sp, int(sh_degree), int(sh_count),
                                C.byref(camera), int(width), int(height), _p(img, C.c_uint8), C.byref(st))
    assert rc == 0
    return img, st, col


def camera_from_bounds(bmin, bmax, width, height):
    a, ap = _f(bmin)
    b, bp = _f(bmax)
    cam = CameraData()
    lib().gsref_camera_from_bounds(ap, bp, int(width), int(height), C.byref(cam))
    return cam


def camera_from_pose(position, rotation_xyzw, width, height):
    a, ap = _f(position)
    b, bp = _f(rotation_xyzw)
    cam = CameraData()
    lib().gsref_camera_from_pose(ap, bp, int(width), int(height), C.byref(cam))
    return cam


def camera_controls_session(width, height, bounds_min, bounds_max, frames, with_camera_data=False):
    """The oracle's CameraControls restatement (gsref_cc_*, CameraControls.cpp:132-253) over a recorded session:
    frames = [(dt, mouseButtons, keys, cursorX, cursorY), ...] -> (len(frames) + 1, 16) float32 states (the state setBounds
    leaves, then the state after every update): position 3, rotation xyzw 4, anchor 3, floor plane 4, speed, mode."""
    L = lib()
    L.gsref_cc_create.restype = C.c_void_p
    L.gsref_cc_create.argtypes = [C.c_float, C.c_float]
    L.gsref_cc_destroy.argtypes = [C.c_void_p]
    L.gsref_cc_set_bounds.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.gsref_cc_update.argtypes = [C.c_void_p, C.c_float, C.c_uint32, C.c_uint32, C.c_double, C.c_double]
    L.gsref_cc_state.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.gsref_cc_camera_data.argtypes = [C.c_void_p, C.POINTER(CameraData)]
    c = L.gsref_cc_create(float(width), float(height))
    try:
        a, ap = _f(bounds_min)
        b, bp = _f(bounds_max)
        L.gsref_cc_set_bounds(c, ap, bp)
        states = np.zeros((len(frames) + 1, 16), np.float32)
        cams = []

        def snap(i):
            L.gsref_cc_state(c, _p(states[i], C.c_float))
            if with_camera_data:
                cam = CameraData()
                L.gsref_cc_camera_data(c, C.byref(cam))
                cams.append(cam)
        snap(0)
        for i, (dt, buttons, keys, x, y) in enumerate(frames):
            L.gsref_cc_update(c, float(dt), int(buttons), int(keys), float(x), float(y))
            snap(i + 1)
    finally:
        L.gsref_cc_destroy(c)
    return (states, cams) if with_camera_data else states


class PlyError(ValueError):
    pass


def parse_ply(path):
    ply = _Ply()
    err = C.create_string_buffer(512)
    n = lib().gsref_parse_ply(os.fsencode(path), C.byref(ply), err, C.c_size_t(512))
    if n < 0:
        raise PlyError(err.value.decode())
    try:
        def arr(ptr, count):
            return np.ctypeslib.as_array(ptr, shape=(count,)).copy() if count else np.zeros(0, np.float32)
        out = {
            "vertexCount": n,
            "shDegree": ply.shDegree,
            "shCount": ply.shCount,
            "boundsMin": np.array(ply.boundsMin[:], np.float32),
            "boundsMax": np.array(ply.boundsMax[:], np.float32),
            "position": arr(ply.position4, 4 * n).reshape(n, 4),
            "scaleAndRotation": arr(ply.scaleAndRotation4, 4 * n).reshape(n, 4),
            "color": arr(ply.color4, 4 * n).reshape(n, 4),
            "sh": arr(ply.sh, ply.shCount * n).reshape(n, ply.shCount) if ply.shDegree else np.zeros((n, 0), np.float32),
        }
    finally:
        lib().gsref_free_ply(C.byref(ply))
    return out


def hardware_concurrency():
    return int(lib().gsref_hardware_concurrency())
