"""cudagaussianrenderer_amd -- Python host over libgsplat_hip.so (MI355X / gfx950).

Mirrors the reference's host interface for the per-frame path (src/GaussianRender.cuh:118-143,
src/Demo.cpp:394-479, src/PlyParser.h:13-21): the eight staged calls, `DeviceBuffer`, `DoubleBuffer`,
`parse_ply`, and a `Renderer` that is the frame-loop body as `render(scene, camera)`.
All arithmetic happens in the HIP library; this module only moves pointers and numpy arrays.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import CameraData, FrameStats, GlobalArgs, GsplatLibraryMissing, TileListArgs  # noqa: F401

TILE_SIZE = 16
SH_GROUP_SIZE = 256
FOOTPRINT_GAUSSIAN = 0      # exp(-q/2), 3 sigma extent
FOOTPRINT_EPANECHNIKOV = 1  # the reference's EPANECHNIKOV_KERNEL option: 1 - q/7, sqrt(7) sigma extent
PIPELINE_REFERENCE = 0  # the reference's stage order: one 64-bit tile|depth radix sort of all pairs
PIPELINE_TWO_LEVEL = 1  # default: depth-sort the splats, emit, stable-sort the pairs on the tile bits
TILE_START_ORDER_SCREEN = 0  # the raster's tiles start in the XCD-aware screen order
TILE_START_ORDER_PREVIOUS_FRAME = 1  # ... longest first by the previous frame's consumption (include/gsplat_hip.h)
STORAGE_ORDER_CALLER = 0  # splats stored in the caller's order (the reference's behaviour)
STAGED_BUILDER_REFERENCE_ORDER = 0  # gs_set_staged_builder: (splat, local tile) emission order + unconditional sort
DEPTH_LAYERS_CHOOSE = 0  # gs_renderer_set_depth_layers: the renderer chooses from the frames' pair totals (for one frame at a time: two layers from 16 M pairs, three from 40 M)
DEPTH_LAYERS_CHOOSE_PIPELINED = -1  # ... for a host with several frames in flight on the GPU (two layers from 4.5 M pairs)
STAGED_BUILDER_SORTED = 1  # default: buildTileList emits in the order the sort leaves, sortTileList looks before it sorts
STORAGE_ORDER_MORTON = 1  # splats stored along a Morton curve: same frames, faster (include/gsplat_hip.h)


def lib():
    return _capi.load()


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- DeviceBuffer<T> (src/DeviceBuffer.cu:8-127) ---------------------------------------------------
class DeviceBuffer:
    """RAII device allocation with upload/download, grow-only resize and byte memset."""

    def __init__(self, data=None, nbytes=None, dtype=None):
        self.ptr = None
        self.nbytes = 0
        self.allocated = 0
        self.dtype = np.dtype(dtype) if dtype is not None else None
        if data is not None:
            self.copy_from(data)
        elif nbytes is not None:
            self.resize_if_needed(nbytes)

    def resize_if_needed(self, nbytes):
        """DeviceBuffer.cu:47-64. Returns True when a new allocation was made."""
        if nbytes > self.allocated:
            self.release()
            self.ptr = lib().gs_device_alloc(nbytes)
            self.allocated = nbytes
            self.nbytes = nbytes
            return True
        self.nbytes = nbytes
        return False

    def copy_from(self, array):
        a = np.ascontiguousarray(array)
        self.dtype = a.dtype
        self.resize_if_needed(a.nbytes)
        if a.nbytes:
            lib().gs_device_upload(self.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes)

    def copy_to(self, count=None, dtype=None, shape=None):
        dt = np.dtype(dtype) if dtype is not None else self.dtype
        n = (self.nbytes // dt.itemsize) if count is None else count
        out = np.empty(n, dt)
        if n:
            lib().gs_device_download(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes)
        return out.reshape(shape) if shape is not None else out

    def clear_memory(self, value):
        """DeviceBuffer.cu:40-44: byte-wise memset (value 255 -> int32 -1)."""
        if self.nbytes:
            lib().gs_device_memset(self.ptr, value, self.nbytes)

    def release(self):
        if self.ptr:
            lib().gs_device_free(self.ptr)
        self.ptr = None
        self.nbytes = 0
        self.allocated = 0

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class DoubleBuffer:
    """GaussianRender.cuh:87-116."""

    def __init__(self, current, alternate):
        self.buffers = [current, alternate]
        self.selector = 0

    def current(self):
        return self.buffers[self.selector]

    def alternate(self):
        return self.buffers[self.selector ^ 1]


class GpuTimer:
    """Utilities.h:155-187 CudaTimer."""

    def __init__(self):
        self.handle = lib().gs_timer_create()

    def elapsed_ms(self):
        return float(lib().gs_timer_elapsed_ms(self.handle))

    def __del__(self):
        try:
            lib().gs_timer_destroy(self.handle)
        except Exception:
            pass


# ---- the eight staged functions, reference names ---------------------------------------------------
def set_global_args(args):
    lib().gs_set_global_args(C.byref(args))


def set_tile_list_args(args):
    lib().gs_set_tile_list_args(C.byref(args))


def _t(timer):
    return timer.handle if timer is not None else None


def evaluate_spherical_harmonics(timer, count):
    lib().gs_evaluate_spherical_harmonics(_t(timer), count)


def evaluate_splat_clip_data(timer, count):
    lib().gs_evaluate_splat_clip_data(_t(timer), count)


def build_tile_list(timer, num_blocks, capacity):
    return int(lib().gs_build_tile_list(_t(timer), num_blocks, capacity))


def sort_tile_list(timer, size, temp, keys, values):
    """`temp` is a two-element list [ptr, nbytes] owned by the caller, updated in place (void*&, size_t&)."""
    k = _capi.DoubleBufferU64((C.c_void_p * 2)(keys.buffers[0].ptr, keys.buffers[1].ptr), keys.selector)
    v = _capi.DoubleBufferI32((C.c_void_p * 2)(values.buffers[0].ptr, values.buffers[1].ptr), values.selector)
    tp = C.c_void_p(temp[0])
    tb = C.c_size_t(temp[1])
    lib().gs_sort_tile_list(_t(timer), size, C.byref(tp), C.byref(tb), C.byref(k), C.byref(v))
    temp[0], temp[1] = tp.value, tb.value
    keys.selector = k.selector
    values.selector = v.selector


def evaluate_tile_range(timer, size):
    lib().gs_evaluate_tile_range(_t(timer), size)


def rasterize_tile(timer):
    lib().gs_rasterize_tile(_t(timer))


def set_footprint(footprint):
    """GaussianRender.cu:8-9 (EPANECHNIKOV_KERNEL) as a process-wide run-time switch."""
    lib().gs_set_footprint(int(footprint))


def synchronize():
    lib().gs_synchronize()


# ---- host helpers -----------------------------------------------------------------------------------
def encode_vec4(v):
    return int(lib().gs_encode_vec4(_fp(_f32(v))))


def sort_end_bit(width, height):
    return int(lib().gs_sort_end_bit(width, height))


def tile_counts(width, height):
    tx, ty = C.c_int(), C.c_int()
    lib().gs_tile_counts(width, height, C.byref(tx), C.byref(ty))
    return tx.value, ty.value


def realign_spherical_harmonics(sh, sh_count, splat_count, group_size=SH_GROUP_SIZE):
    src = _f32(sh).reshape(-1)
    n = lib().gs_realigned_sh_size(group_size, sh_count, splat_count)
    dst = np.zeros(n, np.float32)
    lib().gs_realign_spherical_harmonics(_fp(src), _fp(dst), group_size, sh_count, splat_count)
    return dst


def camera_from_bounds(bounds_min, bounds_max, width, height):
    cam = CameraData()
    lib().gs_camera_from_bounds(_fp(_f32(bounds_min)), _fp(_f32(bounds_max)), width, height, C.byref(cam))
    return cam


def camera_from_pose(position, rotation_xyzw, width, height):
    cam = CameraData()
    lib().gs_camera_from_pose(_fp(_f32(position)), _fp(_f32(rotation_xyzw)), width, height, C.byref(cam))
    return cam


MOUSE_LEFT, MOUSE_MIDDLE, MOUSE_RIGHT = 1, 2, 4
KEY_W, KEY_A, KEY_S, KEY_D, KEY_Q, KEY_E = 1, 2, 4, 8, 16, 32


class CameraControls:
    """The reference's interactive camera (CameraControls.h:7-76) without a window: `update` takes the frame's input state
    (mouse buttons, cursor, keys) instead of polling GLFW.  Left button = look around, middle = orbit the anchor, right = pan,
    W A S D Q E = move (CameraControls.cpp:148-253); same poses as the reference, bit for bit (csrc/gs_camera.cpp)."""

    def __init__(self, width, height):
        self._h = lib().gs_camera_controls_create(float(width), float(height))
        if not self._h:
            raise MemoryError("gs_camera_controls_create")

    def close(self):
        if self._h:
            lib().gs_camera_controls_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_bounds(self, bounds_min, bounds_max):
        lib().gs_camera_controls_set_bounds(self._h, _fp(_f32(bounds_min)), _fp(_f32(bounds_max)))

    def set_pose(self, position, rotation_xyzw):
        lib().gs_camera_controls_set_pose(self._h, _fp(_f32(position)), _fp(_f32(rotation_xyzw)))

    def update(self, dt, mouse_buttons=0, keys=0, cursor=(0.0, 0.0)):
        inp = _capi.CameraInput(int(mouse_buttons), int(keys), float(cursor[0]), float(cursor[1]))
        lib().gs_camera_controls_update(self._h, C.byref(inp), float(dt))

    def state(self):
        st = _capi.CameraControlsState()
        lib().gs_camera_controls_get_state(self._h, C.byref(st))
        return st

    def camera_data(self):
        cam = CameraData()
        lib().gs_camera_controls_camera_data(self._h, C.byref(cam))
        return cam


def morton_order(position):
    """gs_morton_order: storage_to_splat for (N, 4) positions."""
    pos = _f32(position)
    out = np.zeros(max(pos.shape[0], 1), np.int32)
    lib().gs_morton_order(_fp(pos), pos.shape[0], out.ctypes.data_as(C.POINTER(C.c_int32)))
    return out[:pos.shape[0]]


def generate_random_gaussians(count, seed=1234, min_scale=0.01, max_scale=0.5, extent=4.0):
    """Demo.cpp:119-143,258-269 with a seeded PRNG. Returns position, scaleAndRotation, color (N,4) float32."""
    pos = np.zeros((count, 4), np.float32)
    sr = np.zeros((count, 4), np.float32)
    col = np.zeros((count, 4), np.float32)
    lo = _f32([-extent] * 3)
    hi = _f32([extent] * 3)
    lib().gs_generate_random_gaussians(count, seed, min_scale, max_scale, _fp(lo), _fp(hi), _fp(pos), _fp(sr), _fp(col))
    return pos, sr, col


class PlyError(ValueError):
    """The reference throws std::invalid_argument (PlyParser.cpp:39-237); same message text."""


def parse_ply(path):
    """parsePly (PlyParser.h:13-21). Returns a dict of numpy arrays."""
    import os

    ply = _capi.Ply()
    err = C.create_string_buffer(512)
    n = lib().gs_parse_ply(os.fsencode(path), C.byref(ply), err, 512)
    if n < 0:
        raise PlyError(err.value.decode())
    try:
        def arr(ptr, count):
            return np.ctypeslib.as_array(ptr, shape=(count,)).copy() if count else np.zeros(0, np.float32)
        deg, cnt = ply.sphericalHarmonicsDegree, ply.sphericalHarmonicsCount
        return {
            "vertexCount": n,
            "shDegree": deg,
            "shCount": cnt,
            "boundsMin": np.array(ply.boundsMin[:], np.float32),
            "boundsMax": np.array(ply.boundsMax[:], np.float32),
            "position": arr(ply.position, 4 * n).reshape(n, 4),
            "scaleAndRotation": arr(ply.scaleAndRotation, 4 * n).reshape(n, 4),
            "color": arr(ply.color, 4 * n).reshape(n, 4),
            "sh": arr(ply.sphericalHarmonics, cnt * n).reshape(n, cnt) if deg else np.zeros((n, 0), np.float32),
        }
    finally:
        lib().gs_free_ply(C.byref(ply))


def render(position, scale_and_rotation, color, camera, width, height, sh=None, sh_degree=0, sh_count=0):
    """gs_render: the scene (host arrays, as parse_ply returns them) and a camera in, the RGBA8 image (H, W, 4) out -- one frame; keep a
    Renderer for more than one."""
    pos, sr, col = _f32(position), _f32(scale_and_rotation), _f32(color)
    rows = _f32(sh).reshape(-1) if sh_degree else None
    scene = _capi.Scene(pos.shape[0], _fp(pos), _fp(sr), _fp(col), _fp(rows) if rows is not None else None, int(sh_degree), int(sh_count))
    img = np.empty((height, width, 4), np.uint8)
    rc = lib().gs_render(C.byref(scene), C.byref(camera), width, height, img.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError(f"gs_render failed ({rc})")
    return img


# ---- render(scene, camera) ----------------------------------------------------------------------------
class Renderer:
    """The frame loop body of src/Demo.cpp:352-479 behind one call."""

    def __init__(self, width, height, tile_rows=None, pipeline=None, band_cull=None, views=1, storage_order=None):
        self.width, self.height = width, height
        self.handle = lib().gs_renderer_create(width, height)
        if not self.handle:
            raise ValueError(f"gs_renderer_create({width}, {height}) refused: at most 65536 pixels per axis")
        self.views = int(views)
        if self.views != 1 and lib().gs_renderer_set_views(self.handle, self.views) != 0:
            raise ValueError(f"views must be 1..4 (got {views})")
        if tile_rows is not None:
            lib().gs_renderer_set_tile_rows(self.handle, tile_rows[0], tile_rows[1])
        if pipeline is not None:
            self.set_pipeline(pipeline)
        if band_cull is not None:  # 0 auto (on for a strict sub-band of tile rows), 1 on, 2 off
            lib().gs_renderer_set_band_cull(self.handle, band_cull)
        if storage_order is not None:  # STORAGE_ORDER_CALLER / STORAGE_ORDER_MORTON, or "caller" / "morton"
            order = {"caller": STORAGE_ORDER_CALLER, "file": STORAGE_ORDER_CALLER, "morton": STORAGE_ORDER_MORTON}.get(storage_order, storage_order)
            lib().gs_renderer_set_storage_order(self.handle, int(order))
        self.splat_count = 0

    def set_pipeline(self, pipeline):
        """PIPELINE_REFERENCE is refused while the renderer holds a scene stored in Morton order (it breaks depth ties by storage order)."""
        if lib().gs_renderer_set_pipeline(self.handle, int(pipeline)) != 0:
            raise ValueError("gs_renderer_set_pipeline: the reference-shaped pipeline cannot run on a scene stored in Morton order")

    def set_scene(self, position, scale_and_rotation, color, sh=None, sh_degree=0, sh_count=0):
        pos, sr, col = _f32(position), _f32(scale_and_rotation), _f32(color)
        n = pos.shape[0]
        shp = _fp(_f32(sh).reshape(-1)) if sh_degree else None
        rc = lib().gs_renderer_set_scene(self.handle, n, _fp(pos), _fp(sr), _fp(col), shp, sh_degree, sh_count)
        if rc != 0:
            raise ValueError("gs_renderer_set_scene failed")
        self.splat_count = n

    def share_scene(self, owner):
        """Use `owner`'s read-only scene buffers (several renderers of one scene = frames in flight); `owner` must outlive self."""
        if lib().gs_renderer_share_scene(self.handle, owner.handle) != 0:
            raise ValueError("gs_renderer_share_scene failed")
        self.splat_count = owner.splat_count
        self._scene_owner = owner  # keep it alive

    def storage_order(self):
        """storage_to_splat[k] = the caller's index of the splat stored in slot k: per-splat buffers and the pair list's values are
        in storage order (identity unless the renderer was created with storage_order="morton")."""
        out = np.zeros(max(self.splat_count, 1), np.int32)
        n = lib().gs_renderer_storage_order(self.handle, out.ctypes.data_as(C.POINTER(C.c_int32)))
        return out[:n]

    def set_back_buffer(self, device_ptr):
        lib().gs_renderer_set_back_buffer(self.handle, device_ptr)

    def render(self, camera, timed=False, stats=False):
        """timed: False/0 none, True/1 blocking per-stage timers (reference style), 2 non-blocking stage events."""
        mode = int(timed)
        st = FrameStats() if (stats or mode == 1) else None
        rc = lib().gs_renderer_render(self.handle, C.byref(camera), mode, C.byref(st) if st is not None else None)
        if rc != 0:
            raise RuntimeError(f"gs_renderer_render failed ({rc})")
        return st

    def render_views(self, cameras):
        """`views` frames of the scene by ONE replay of the captured chain (set_graph(True) first); image v: read_image(view=v)."""
        if len(cameras) != self.views:
            raise ValueError(f"{self.views} cameras expected, got {len(cameras)}")
        arr = (_capi.CameraData * self.views)(*cameras)
        rc = lib().gs_renderer_render_views(self.handle, arr)
        if rc != 0:
            raise RuntimeError(f"gs_renderer_render_views failed ({rc})")

    def set_depth_layers(self, layers, fractions=None):
        """gs_renderer_set_depth_layers: list / sort / blend the depth-sorted splats in `layers` layers (captured frames; same image bytes)."""
        arr = (C.c_float * len(fractions))(*fractions) if fractions else None
        if lib().gs_renderer_set_depth_layers(self.handle, int(layers), arr) != 0:
            raise ValueError("gs_renderer_set_depth_layers: 0 / -1 (the renderer chooses; -1: for several frames in flight) or 1..4 layers, ascending fractions in (0, 1)")

    def depth_layers_in_use(self):
        """Layers the last captured frame was made of."""
        return int(lib().gs_renderer_depth_layers_in_use(self.handle))

    def listed_pairs(self):
        """Pairs the last captured frame listed, summed over its depth layers."""
        return int(lib().gs_renderer_listed_pairs(self.handle))

    def set_tile_start_order(self, mode):
        """TILE_START_ORDER_SCREEN (default) or TILE_START_ORDER_PREVIOUS_FRAME: the raster's tiles started longest-first by the previous
        frame's consumption (interactive cameras; same image bytes)."""
        if lib().gs_renderer_set_tile_start_order(self.handle, int(mode)) != 0:
            raise ValueError("gs_renderer_set_tile_start_order: unknown mode")

    def tile_order_state(self):
        """(table in force for the next frame, sum |consumed now - before|, sum consumed now, frames seen) or None."""
        out = (C.c_int32 * 4)()
        if lib().gs_renderer_debug_tile_order_state(self.handle, out) != 0:
            return None
        return tuple(int(v) for v in out)

    def set_footprint(self, footprint):
        """This renderer's footprint (FOOTPRINT_GAUSSIAN / FOOTPRINT_EPANECHNIKOV; -1 = the process-wide gs_set_footprint at frame time)."""
        lib().gs_renderer_set_footprint(self.handle, int(footprint))

    def set_block_cull(self, enabled=True):
        """A/B: the band cull of a Morton-stored scene by blocks of 256 neighbours (default) or by projecting every splat (round 5)."""
        lib().gs_renderer_debug_set_block_cull(self.handle, 1 if enabled else 0)

    def block_cull_kept(self):
        """(blocks the last band-culled frame's block test kept, blocks x views), or (-1, total) without block bounds."""
        total = C.c_int64(0)
        kept = lib().gs_renderer_debug_block_cull_kept(self.handle, C.byref(total))
        return int(kept), int(total.value)

    def band_kept_splats(self):
        """Splats the last band-culled frame kept (the culled list's length, all views)."""
        return int(lib().gs_renderer_debug_band_kept_splats(self.handle))

    def layered_raster_report(self, camera):
        """One frame of `camera` as the captured layered frame's kernels, launched plainly with an event pair around every layer's raster
        launch: list of per-layer dicts (ms, consumed entries, tiles run, tiles that loaded / stored their state), or None when the
        renderer's last captured frame was not layered."""
        rep = _capi.LayeredRasterReport()
        if lib().gs_renderer_debug_layered_raster_report(self.handle, C.byref(camera), C.byref(rep)) != 0:
            return None
        return [{"raster_ms": float(rep.rasterMs[l]), "consumed_pairs": int(rep.consumedPairs[l]), "tiles_run": int(rep.tilesRun[l]),
                 "state_tile_loads": int(rep.stateTileLoads[l]), "state_tile_stores": int(rep.stateTileStores[l])} for l in range(rep.layers)]

    def colours_on_demand(self):
        """True when this renderer's frames evaluate SH colours inside the raster kernel (no SH pass)."""
        return bool(lib().gs_renderer_colours_on_demand(self.handle))

    def set_graph(self, enabled=True):
        """Replay plain frames (render(camera) without timers/stats) as one captured hipGraph launch."""
        lib().gs_renderer_set_graph(self.handle, 1 if enabled else 0)

    def captured_retries(self):
        """Captured frames rendered a second time because their pair demand exceeded the list capacity (never handed out truncated)."""
        return int(lib().gs_renderer_captured_retries(self.handle))

    def set_stream(self, hip_stream):
        """Give this renderer its own stream (e.g. torch.cuda.Stream().cuda_stream) so frames can overlap."""
        lib().gs_renderer_set_stream(self.handle, hip_stream)

    def begin(self, camera, timed=0):
        """First half of a frame (up to the pair emission); returns immediately."""
        rc = lib().gs_renderer_begin(self.handle, C.byref(camera), int(timed))
        if rc != 0:
            raise RuntimeError(f"gs_renderer_begin failed ({rc})")

    def finish(self, stats=False):
        """Second half (tile sort, ranges, raster); picks up the pair count of begin()."""
        st = FrameStats() if stats else None
        rc = lib().gs_renderer_finish(self.handle, C.byref(st) if st is not None else None)
        if rc != 0:
            raise RuntimeError(f"gs_renderer_finish failed ({rc})")
        return st

    def collect_stage_times(self):
        """(frames, FrameStats of sums) over the frames rendered with timed=2 since the last call."""
        st = FrameStats()
        frames = lib().gs_renderer_collect_stage_times(self.handle, C.byref(st))
        return frames, st

    def read_image(self, view=0):
        img = np.empty((self.height, self.width, 4), np.uint8)
        if lib().gs_renderer_read_back_buffer_view(self.handle, int(view), img.ctypes.data_as(C.c_void_p)) != 0:
            raise ValueError(f"no view {view}")
        return img

    def back_buffer_ptr(self, view=0):
        return lib().gs_renderer_back_buffer_view(self.handle, int(view))

    def buffers(self):
        b = _capi.RendererBuffers()
        lib().gs_renderer_buffers(self.handle, C.byref(b))
        return b

    def download(self, ptr, count, dtype):
        out = np.empty(count, dtype)
        if count:
            lib().gs_device_download(out.ctypes.data_as(C.c_void_p), ptr, out.nbytes)
        return out

    def close(self):
        if self.handle:
            lib().gs_renderer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
