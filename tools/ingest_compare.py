"""Authoring container only (needs oracle/_ref/libgsupstream.so = the reference's own PlyParser.cpp compiled where it lies): the parse
step of SURVEY.md section 8(f2) -- gs_parse_ply (block read + transpose) against the reference's parsePly (one ifstream::read per value,
PlyParser.cpp:132-140) on the SAME synthetic files, page-cached, best of `repeats`; checks that both return the same arrays.
usage: python tools/ingest_compare.py [splats ...]   (default 1000000 5000000)"""
import ctypes as C
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cudagaussianrenderer_amd import _capi, scenes  # noqa: E402


class UpPly(C.Structure):
    _fields_ = [("vertexCount", C.c_int32), ("shDegree", C.c_int32), ("shCount", C.c_int32), ("boundsMin", C.c_float * 3), ("boundsMax", C.c_float * 3),
                ("position4", C.POINTER(C.c_float)), ("scaleAndRotation4", C.POINTER(C.c_float)), ("color4", C.POINTER(C.c_float)), ("sh", C.POINTER(C.c_float))]


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1_000_000, 5_000_000]
    up = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libgsupstream.so"), mode=os.RTLD_LAZY)
    up.up_parse_ply.argtypes = [C.c_char_p, C.POINTER(UpPly), C.c_char_p, C.c_size_t]
    up.up_free_ply.argtypes = [C.POINTER(UpPly)]
    lib = _capi.load()
    repeats = 3
    print(f"host: {os.cpu_count()} hardware threads (authoring container); files in a tmpfs-backed temporary directory, page-cached; best of {repeats}")
    for n in sizes:
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "scene.ply")
            names, table = scenes.synthetic_vertex_table(n, sh_degree=3, seed=1234, log_scale_mean=-3.6)
            scenes.write_ply(path, names, table)
            del table
            size_mb = os.path.getsize(path) / 1e6
            err = C.create_string_buffer(256)
            ours, theirs = [], []
            same = None
            for _ in range(repeats):
                ply = _capi.Ply()
                t0 = time.perf_counter()
                got = lib.gs_parse_ply(os.fsencode(path), C.byref(ply), err, 256)
                ours.append(time.perf_counter() - t0)
                ref = UpPly()
                t0 = time.perf_counter()
                want = up.up_parse_ply(os.fsencode(path), C.byref(ref), err, 256)
                theirs.append(time.perf_counter() - t0)
                assert got == want == n, (got, want)
                if same is None:
                    cnt = ply.sphericalHarmonicsCount
                    # (compared as bit patterns: scaleAndRotation.w is four packed bytes, some of which read as NaN)
                    same = all(np.array_equal(np.ctypeslib.as_array(a, shape=(k,)).view(np.uint32), np.ctypeslib.as_array(b, shape=(k,)).view(np.uint32))
                               for a, b, k in ((ply.position, ref.position4, 4 * n), (ply.scaleAndRotation, ref.scaleAndRotation4, 4 * n),
                                               (ply.color, ref.color4, 4 * n), (ply.sphericalHarmonics, ref.sh, cnt * n)))
                lib.gs_free_ply(C.byref(ply))
                up.up_free_ply(C.byref(ref))
            a, b = min(ours), min(theirs)
            print(f"{n:>9d} splats, SH 3, {size_mb:7.1f} MB: gs_parse_ply {a:6.3f} s = {size_mb / a:7.0f} MB/s | reference parsePly {b:6.3f} s = "
                  f"{size_mb / b:6.0f} MB/s | {b / a:4.1f}x | same arrays: {same}")


if __name__ == "__main__":
    main()
