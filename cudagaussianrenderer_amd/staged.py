"""The reference's frame loop over the EIGHT STAGED CALLS (src/Demo.cpp:394-479) for Python hosts: caller-owned device buffers
(DeviceBuffer, as Demo.cpp:314-331 allocates them), GlobalArgs / TileListArgs filled as Demo.cpp:406-430 fills them, one method per
stage -- each brings its stage's outputs back as numpy arrays, which is what the parity tests compare with the oracle.

    f = StagedFrame(position, scaleAndRotation, color, camera, 1920, 1080, sh_aligned, 3, 48)
    f.begin(); f.run_sh(); f.run_clip(); f.run_build(); f.run_sort(); f.run_ranges(); image = f.run_raster()

(the one-call renderer, cudagaussianrenderer_amd.Renderer, is the same frame without the eight round trips)."""
import ctypes as C

import numpy as np

import cudagaussianrenderer_amd as gs
from cudagaussianrenderer_amd import _capi


class StagedFrame:
    """Caller-owned device buffers + the staged calls, reference style."""

    def __init__(self, position, scale_and_rotation, color, camera, width, height, sh_aligned=None, sh_degree=0,
                 sh_count=0, capacity=None, tile_rows=(0, 0)):
        self.n = int(position.shape[0])
        self.width, self.height = width, height
        self.tiles_x, self.tiles_y = gs.tile_counts(width, height)
        self.total_tiles = self.tiles_x * self.tiles_y
        n = max(self.n, 1)
        f32 = np.float32
        self.position = gs.DeviceBuffer(np.ascontiguousarray(position, f32))
        self.scale_rot = gs.DeviceBuffer(np.ascontiguousarray(scale_and_rotation, f32))
        self.color = gs.DeviceBuffer(np.ascontiguousarray(color, f32))
        self.sh = gs.DeviceBuffer(np.ascontiguousarray(sh_aligned, f32)) if sh_degree else gs.DeviceBuffer(nbytes=16)
        self.clip_xy = gs.DeviceBuffer(nbytes=8 * n, dtype=f32)
        self.clip_z = gs.DeviceBuffer(nbytes=4 * n, dtype=f32)
        self.conic = gs.DeviceBuffer(nbytes=16 * n, dtype=f32)
        self.ellipse = gs.DeviceBuffer(nbytes=16 * n, dtype=f32)
        self.tile_range = gs.DeviceBuffer(nbytes=8 * self.total_tiles, dtype=np.int32)
        self.back_buffer = gs.DeviceBuffer(nbytes=4 * width * height, dtype=np.uint8)
        self.capacity = int(capacity if capacity is not None else max(8 * self.n, 64))  # Demo.cpp:325
        self.keys = [gs.DeviceBuffer(nbytes=8 * self.capacity, dtype=np.uint64) for _ in range(2)]
        self.values = [gs.DeviceBuffer(nbytes=4 * self.capacity, dtype=np.int32) for _ in range(2)]
        self.sort_temp = [None, 0]
        self.timer = gs.GpuTimer()
        self.size = 0
        self.sorted_sel = 0

        a = _capi.GlobalArgs()
        a.cameraData = camera
        a.position = self.position.ptr
        a.scaleAndRotation = self.scale_rot.ptr
        a.color = self.color.ptr
        a.sphericalHarmonicsDegree = sh_degree
        a.sphericalHarmonicsCount = sh_count
        a.sphericalHarmonics = self.sh.ptr
        a.positionClipSpaceXY = self.clip_xy.ptr
        a.positionClipSpaceZ = self.clip_z.ptr
        a.conic = self.conic.ptr
        a.screenEllipse = self.ellipse.ptr
        a.splatCount = self.n
        a.backBuffer = self.back_buffer.ptr
        a.tileRange = self.tile_range.ptr
        a.width, a.height = width, height
        a.tileRowBegin, a.tileRowEnd = tile_rows
        self.args = a
        self.sh_degree = sh_degree

    # -- Demo.cpp:399-430
    def begin(self):
        self.back_buffer.clear_memory(0)
        self.tile_range.clear_memory(255)
        gs.set_global_args(self.args)
        self._set_list(self.keys[0], self.values[0], 0)

    def _set_list(self, keys, values, size):
        t = _capi.TileListArgs(keys.ptr, values.ptr, size, self.capacity)
        gs.set_tile_list_args(t)

    def run_sh(self):
        gs.evaluate_spherical_harmonics(self.timer, self.n)
        return self.color.copy_to(dtype=np.float32, shape=(-1, 4))[: self.n]

    def run_clip(self):
        gs.evaluate_splat_clip_data(self.timer, self.n)
        n = self.n
        return (self.clip_xy.copy_to(dtype=np.float32).reshape(-1, 2)[:n], self.clip_z.copy_to(dtype=np.float32)[:n],
                self.ellipse.copy_to(dtype=np.float32).reshape(-1, 4)[:n], self.conic.copy_to(dtype=np.float32).reshape(-1, 4)[:n])

    def run_build(self):
        self.size = gs.build_tile_list(self.timer, 256 * 8, self.capacity)
        return (self.keys[0].copy_to(count=self.size, dtype=np.uint64), self.values[0].copy_to(count=self.size, dtype=np.int32),
                self.size)

    def run_sort(self):
        k = gs.DoubleBuffer(self.keys[0], self.keys[1])
        v = gs.DoubleBuffer(self.values[0], self.values[1])
        gs.sort_tile_list(self.timer, self.size, self.sort_temp, k, v)
        self.sorted_sel = k.selector
        assert v.selector == k.selector
        self._set_list(k.current(), v.current(), self.size)
        return k.current().copy_to(count=self.size, dtype=np.uint64), v.current().copy_to(count=self.size, dtype=np.int32)

    def run_ranges(self):
        if self.size != 0:  # Demo.cpp:469
            gs.evaluate_tile_range(self.timer, self.size)
        return self.tile_range.copy_to(dtype=np.int32)

    def run_raster(self):
        gs.rasterize_tile(self.timer)
        gs.synchronize()
        return self.back_buffer.copy_to(dtype=np.uint8).reshape(self.height, self.width, 4)

    def upload_list(self, keys, values):
        """Replace the current list with host data (for stage-isolated tests)."""
        self.size = int(keys.shape[0])
        assert self.size <= self.capacity
        if self.size:
            gs.lib().gs_device_upload(self.keys[0].ptr, np.ascontiguousarray(keys, np.uint64).ctypes.data_as(C.c_void_p), 8 * self.size)
            gs.lib().gs_device_upload(self.values[0].ptr, np.ascontiguousarray(values, np.int32).ctypes.data_as(C.c_void_p), 4 * self.size)
        self._set_list(self.keys[0], self.values[0], self.size)

    def upload(self, buf, array):
        a = np.ascontiguousarray(array)
        gs.lib().gs_device_upload(buf.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes)

    def free_sort_temp(self):
        if self.sort_temp[0]:
            gs.lib().gs_device_free(self.sort_temp[0])
            self.sort_temp = [None, 0]


def image_diff_stats(a, b):
    """Per-channel absolute difference statistics of two RGBA8 images."""
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return {"max": int(d.max()), "frac_gt1": float((d > 1).mean()), "frac_gt0": float((d > 0).mean())}
