"""ctypes declarations for libgsplat_hip.so (include/gsplat_hip.h).

The library is the product: there is no fallback.  If it is missing, loading fails loudly
(`GsplatLibraryMissing`) instead of silently computing something else.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSPLAT_HIP_LIBRARY: another build of the same library (A/B measurements of kernel variants, tools/ab_bench.sh)
LIB_PATH = os.environ.get("GSPLAT_HIP_LIBRARY") or os.path.join(_HERE, "libgsplat_hip.so")


class GsplatLibraryMissing(RuntimeError):
    pass


class CameraData(C.Structure):
    """GsCameraData == the reference's CameraData (GaussianRender.cuh:17-26)."""

    _fields_ = [
        ("viewProjection", C.c_float * 16),
        ("projection", C.c_float * 16),
        ("view", C.c_float * 16),
        ("fovCotangent", C.c_float * 2),
        ("depthScaleBias", C.c_float * 2),
        ("position", C.c_float * 3),
        ("aspect", C.c_float),
    ]


class GlobalArgs(C.Structure):
    """GsGlobalArgs == the reference's GlobalArgs (GaussianRender.cuh:56-74) + width/height/band."""

    _fields_ = [
        ("cameraData", CameraData),
        ("position", C.c_void_p),
        ("scaleAndRotation", C.c_void_p),
        ("color", C.c_void_p),
        ("sphericalHarmonicsDegree", C.c_int32),
        ("sphericalHarmonicsCount", C.c_int32),
        ("sphericalHarmonics", C.c_void_p),
        ("positionClipSpaceXY", C.c_void_p),
        ("positionClipSpaceZ", C.c_void_p),
        ("conic", C.c_void_p),
        ("screenEllipse", C.c_void_p),
        ("splatCount", C.c_int32),
        ("backBuffer", C.c_void_p),
        ("tileRange", C.c_void_p),
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("tileRowBegin", C.c_int32),
        ("tileRowEnd", C.c_int32),
    ]


class CameraInput(C.Structure):
    """GsCameraInput: what CameraControls::update polls from GLFW, handed in by the caller."""

    _fields_ = [("mouseButtons", C.c_uint32), ("keys", C.c_uint32), ("cursorX", C.c_double), ("cursorY", C.c_double)]


class CameraControlsState(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("rotationXYZW", C.c_float * 4), ("anchorPosition", C.c_float * 3), ("speed", C.c_float),
                ("mode", C.c_int32)]


class TileListArgs(C.Structure):
    _fields_ = [("keys", C.c_void_p), ("values", C.c_void_p), ("size", C.c_int32), ("capacity", C.c_int32)]


class DoubleBufferU64(C.Structure):
    _fields_ = [("buffers", C.c_void_p * 2), ("selector", C.c_int32)]


class DoubleBufferI32(C.Structure):
    _fields_ = [("buffers", C.c_void_p * 2), ("selector", C.c_int32)]


class Scene(C.Structure):
    """GsScene (gs_render)."""
    _fields_ = [("splatCount", C.c_int32), ("position", C.POINTER(C.c_float)), ("scaleAndRotation", C.POINTER(C.c_float)),
                ("color", C.POINTER(C.c_float)), ("sphericalHarmonics", C.POINTER(C.c_float)), ("sphericalHarmonicsDegree", C.c_int32),
                ("sphericalHarmonicsCount", C.c_int32)]


class Ply(C.Structure):
    _fields_ = [
        ("vertexCount", C.c_int32),
        ("sphericalHarmonicsDegree", C.c_int32),
        ("sphericalHarmonicsCount", C.c_int32),
        ("boundsMin", C.c_float * 3),
        ("boundsMax", C.c_float * 3),
        ("position", C.POINTER(C.c_float)),
        ("scaleAndRotation", C.POINTER(C.c_float)),
        ("color", C.POINTER(C.c_float)),
        ("sphericalHarmonics", C.POINTER(C.c_float)),
    ]


class FrameStats(C.Structure):
    _fields_ = [
        ("msSphericalHarmonics", C.c_float),
        ("msClipData", C.c_float),
        ("msBuildTileList", C.c_float),
        ("msSortTileList", C.c_float),
        ("msTileRanges", C.c_float),
        ("msRasterize", C.c_float),
        ("msFrame", C.c_float),
        ("tileListSize", C.c_int64),
        ("consumedPairs", C.c_int64),
        ("tileListCapacity", C.c_int32),
        ("grewTileList", C.c_int32),
    ]


class LayeredRasterReport(C.Structure):
    _fields_ = [("layers", C.c_int32), ("rasterMs", C.c_float * 4), ("consumedPairs", C.c_int64 * 4), ("tilesRun", C.c_int64 * 4),
                ("stateTileLoads", C.c_int64 * 4), ("stateTileStores", C.c_int64 * 4)]


class RendererBuffers(C.Structure):
    _fields_ = [
        ("position", C.c_void_p),
        ("scaleAndRotation", C.c_void_p),
        ("color", C.c_void_p),
        ("sphericalHarmonics", C.c_void_p),
        ("positionClipSpaceXY", C.c_void_p),
        ("positionClipSpaceZ", C.c_void_p),
        ("conic", C.c_void_p),
        ("screenEllipse", C.c_void_p),
        ("tileRange", C.c_void_p),
        ("keys", C.c_void_p),
        ("values", C.c_void_p),
        ("splatCount", C.c_int32),
        ("tileListSize", C.c_int32),
        ("keyBytes", C.c_int32),
    ]


# name -> (restype, argtypes); exactly the symbols include/gsplat_hip.h (the boundary) and include/gsplat_hip_debug.h (diagnostics,
# A/B switches, counters: DEBUG_SYMBOLS below names those) declare.
SIGNATURES = {
    "gs_timer_create": (C.c_void_p, []),
    "gs_timer_destroy": (None, [C.c_void_p]),
    "gs_timer_start": (None, [C.c_void_p]),
    "gs_timer_stop": (None, [C.c_void_p]),
    "gs_timer_elapsed_ms": (C.c_float, [C.c_void_p]),
    "gs_timer_set_blocking": (None, [C.c_void_p, C.c_int]),
    "gs_set_global_args": (None, [C.POINTER(GlobalArgs)]),
    "gs_set_tile_list_args": (None, [C.POINTER(TileListArgs)]),
    "gs_evaluate_spherical_harmonics": (None, [C.c_void_p, C.c_int32]),
    "gs_evaluate_splat_clip_data": (None, [C.c_void_p, C.c_int32]),
    "gs_build_tile_list": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32]),
    "gs_last_tile_list_demand": (C.c_int64, []),
    "gs_sort_tile_list": (None, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                 C.POINTER(DoubleBufferU64), C.POINTER(DoubleBufferI32)]),
    "gs_evaluate_tile_range": (None, [C.c_void_p, C.c_int32]),
    "gs_rasterize_tile": (None, [C.c_void_p]),
    "gs_set_collect_raster_stats": (None, [C.c_int]),
    "gs_last_raster_consumed_pairs": (C.c_int64, []),
    "gs_set_sort_backend": (None, [C.c_int]),
    "gs_set_staged_builder": (None, [C.c_int]),
    "gs_get_staged_builder": (C.c_int, []),
    "gs_set_staged_tile_start_order": (None, [C.c_int]),
    "gs_get_staged_tile_start_order": (C.c_int, []),
    "gs_debug_staged_tile_order_state": (C.c_int, [C.POINTER(C.c_int32)]),
    "gs_release_thread_scratch": (None, []),
    "gs_set_list_fusion": (None, [C.c_int]),
    "gs_get_list_fusion": (C.c_int, []),
    "gs_renderer_colours_on_demand": (C.c_int, [C.c_void_p]),
    "gs_renderer_set_depth_layers": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "gs_renderer_depth_layers": (C.c_int, [C.c_void_p]),
    "gs_renderer_listed_pairs": (C.c_int64, [C.c_void_p]),
    "gs_renderer_depth_layers_in_use": (C.c_int, [C.c_void_p]),
    "gs_renderer_depth_layer_changes": (C.c_int, [C.c_void_p]),
    "gs_renderer_debug_set_layer_compaction": (None, [C.c_void_p, C.c_int]),
    "gs_renderer_debug_set_block_cull": (None, [C.c_void_p, C.c_int]),
    "gs_renderer_debug_set_layer_compaction_reuse": (None, [C.c_void_p, C.c_int]),
    "gs_renderer_debug_block_cull_kept": (C.c_int64, [C.c_void_p, C.POINTER(C.c_int64)]),
    "gs_renderer_debug_band_kept_splats": (C.c_int64, [C.c_void_p]),
    "gs_renderer_debug_layered_raster_report": (C.c_int, [C.c_void_p, C.POINTER(CameraData), C.POINTER(LayeredRasterReport)]),
    "gs_renderer_debug_set_list_fusion": (None, [C.c_void_p, C.c_int]),
    "gs_renderer_set_tile_start_order": (C.c_int, [C.c_void_p, C.c_int]),
    "gs_renderer_tile_start_order": (C.c_int, [C.c_void_p]),
    "gs_renderer_debug_tile_order_state": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "gs_renderer_debug_set_tile_order_max_mismatch": (None, [C.c_void_p, C.c_float]),
    "gs_renderer_debug_raster_grid_slots": (C.c_int, [C.c_void_p]),
    "gs_renderer_debug_set_tile_order": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int]),
    "gs_renderer_debug_tile_consumed": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "gs_renderer_debug_set_raster_pixels_per_lane": (None, [C.c_void_p, C.c_int]),
    "gs_renderer_set_footprint": (None, [C.c_void_p, C.c_int]),
    "gs_renderer_footprint": (C.c_int, [C.c_void_p]),
    "gs_set_footprint": (None, [C.c_int]),
    "gs_get_footprint": (C.c_int, []),
    "gs_get_sort_backend": (C.c_int, []),
    "gs_radix_sort_pairs": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_int]),
    "gs_debug_set_raster_trace": (None, [C.c_void_p]),
    "gs_set_raster_pixels_per_lane": (None, [C.c_int]),
    "gs_get_raster_pixels_per_lane": (C.c_int, []),
    "gs_set_device": (None, [C.c_int]),
    "gs_set_stream": (None, [C.c_void_p]),
    "gs_get_stream": (C.c_void_p, []),
    "gs_synchronize": (None, []),
    "gs_device_info": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "gs_device_alloc": (C.c_void_p, [C.c_size_t]),
    "gs_device_free": (None, [C.c_void_p]),
    "gs_device_upload": (None, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "gs_device_download": (None, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "gs_device_memset": (None, [C.c_void_p, C.c_int, C.c_size_t]),
    "gs_encode_vec4": (C.c_uint32, [C.POINTER(C.c_float)]),
    "gs_sort_end_bit": (C.c_int, [C.c_int, C.c_int]),
    "gs_tile_counts": (None, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gs_realigned_sh_size": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "gs_realign_spherical_harmonics": (None, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int]),
    "gs_camera_from_bounds": (None, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(CameraData)]),
    "gs_camera_from_pose": (None, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(CameraData)]),
    "gs_camera_controls_create": (C.c_void_p, [C.c_float, C.c_float]),
    "gs_camera_controls_destroy": (None, [C.c_void_p]),
    "gs_camera_controls_set_bounds": (None, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "gs_camera_controls_update": (None, [C.c_void_p, C.POINTER(CameraInput), C.c_float]),
    "gs_camera_controls_camera_data": (None, [C.c_void_p, C.POINTER(CameraData)]),
    "gs_camera_controls_get_state": (None, [C.c_void_p, C.POINTER(CameraControlsState)]),
    "gs_camera_controls_set_pose": (None, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "gs_graphics_gl_register_buffer": (C.c_void_p, [C.c_uint32, C.c_uint32]),
    "gs_graphics_map": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "gs_graphics_unmap": (None, [C.c_void_p]),
    "gs_graphics_unregister": (None, [C.c_void_p]),
    "gs_generate_random_gaussians": (None, [C.c_int32, C.c_uint64, C.c_float, C.c_float, C.POINTER(C.c_float),
                                            C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                            C.POINTER(C.c_float)]),
    "gs_parse_ply": (C.c_int, [C.c_char_p, C.POINTER(Ply), C.c_char_p, C.c_size_t]),
    "gs_free_ply": (None, [C.POINTER(Ply)]),
    "gs_render": (C.c_int, [C.POINTER(Scene), C.POINTER(CameraData), C.c_int, C.c_int, C.c_void_p]),
    "gs_renderer_create": (C.c_void_p, [C.c_int, C.c_int]),
    "gs_renderer_destroy": (None, [C.c_void_p]),
    "gs_renderer_set_tile_rows": (None, [C.c_void_p, C.c_int, C.c_int]),
    "gs_renderer_set_pipeline": (C.c_int, [C.c_void_p, C.c_int]),
    "gs_renderer_set_band_cull": (None, [C.c_void_p, C.c_int]),
    "gs_renderer_set_scene": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                        C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int]),
    "gs_renderer_set_storage_order": (None, [C.c_void_p, C.c_int]),
    "gs_renderer_storage_order": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32)]),
    "gs_morton_order": (None, [C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_int32)]),
    "gs_renderer_share_scene": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gs_renderer_render": (C.c_int, [C.c_void_p, C.POINTER(CameraData), C.c_int, C.POINTER(FrameStats)]),
    "gs_renderer_set_graph": (None, [C.c_void_p, C.c_int]),
    "gs_renderer_set_views": (C.c_int, [C.c_void_p, C.c_int]),
    "gs_renderer_views": (C.c_int, [C.c_void_p]),
    "gs_renderer_render_views": (C.c_int, [C.c_void_p, C.POINTER(CameraData)]),
    "gs_renderer_back_buffer_view": (C.c_void_p, [C.c_void_p, C.c_int]),
    "gs_renderer_read_back_buffer_view": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "gs_renderer_captured_retries": (C.c_int, [C.c_void_p]),
    "gs_debug_set_ablation": (None, [C.c_int]),
    "gs_debug_set_clip_scan_fusion": (None, [C.c_int]),
    "gs_debug_get_clip_scan_fusion": (C.c_int, []),
    "gs_debug_set_await_timeout": (None, [C.c_double]),
    "gs_debug_set_staged_scratch_limit": (None, [C.c_int64]),
    "gs_debug_host_times": (C.c_longlong, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]),
    "gs_renderer_set_stream": (None, [C.c_void_p, C.c_void_p]),
    "gs_renderer_begin": (C.c_int, [C.c_void_p, C.POINTER(CameraData), C.c_int]),
    "gs_renderer_finish": (C.c_int, [C.c_void_p, C.POINTER(FrameStats)]),
    "gs_renderer_collect_stage_times": (C.c_int, [C.c_void_p, C.POINTER(FrameStats)]),
    "gs_renderer_back_buffer": (C.c_void_p, [C.c_void_p]),
    "gs_renderer_set_back_buffer": (None, [C.c_void_p, C.c_void_p]),
    "gs_renderer_read_back_buffer": (None, [C.c_void_p, C.c_void_p]),
    "gs_renderer_buffers": (None, [C.c_void_p, C.POINTER(RendererBuffers)]),
    "gs_comm_unique_id": (C.c_int, [C.c_char_p]),
    "gs_comm_create": (C.c_void_p, [C.c_int, C.c_int, C.c_char_p]),
    "gs_comm_destroy": (None, [C.c_void_p]),
    "gs_comm_rank": (C.c_int, [C.c_void_p]),
    "gs_comm_world": (C.c_int, [C.c_void_p]),
    "gs_band_pixel_rows": (None, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gs_gather_bands": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_void_p]),
    "gs_debug_rccl_self_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gs_debug_rccl_self_exchange_parts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "gs_debug_rccl_self_exchange_streams": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "gs_gather_bands_views": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_void_p]),
    "gs_version": (C.c_char_p, []),
}

# the symbols of include/gsplat_hip_debug.h (everything else above is include/gsplat_hip.h)
DEBUG_SYMBOLS = frozenset([
    "gs_set_collect_raster_stats", "gs_last_raster_consumed_pairs", "gs_renderer_captured_retries", "gs_renderer_depth_layer_changes",
    "gs_renderer_debug_set_layer_compaction", "gs_renderer_debug_set_block_cull", "gs_renderer_debug_set_layer_compaction_reuse", "gs_renderer_debug_block_cull_kept", "gs_renderer_debug_band_kept_splats", "gs_renderer_debug_layered_raster_report",
    "gs_renderer_debug_set_list_fusion", "gs_renderer_debug_raster_grid_slots", "gs_renderer_debug_tile_order_state", "gs_debug_staged_tile_order_state", "gs_renderer_debug_set_tile_order_max_mismatch", "gs_renderer_debug_set_tile_order", "gs_renderer_debug_tile_consumed", "gs_renderer_debug_set_raster_pixels_per_lane", "gs_set_sort_backend", "gs_get_sort_backend",
    "gs_set_list_fusion", "gs_get_list_fusion", "gs_radix_sort_pairs", "gs_set_raster_pixels_per_lane", "gs_get_raster_pixels_per_lane",
    "gs_debug_set_raster_trace", "gs_debug_set_ablation", "gs_debug_set_clip_scan_fusion", "gs_debug_get_clip_scan_fusion",
    "gs_debug_set_await_timeout", "gs_debug_set_staged_scratch_limit", "gs_debug_host_times", "gs_debug_rccl_self_exchange", "gs_debug_rccl_self_exchange_parts",
    "gs_debug_rccl_self_exchange_streams"])

_lib = None


def load():
    """Load libgsplat_hip.so and bind every declared symbol. Raises if the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GsplatLibraryMissing(
            f"{LIB_PATH} not found: build it with `make -C cudagaussianrenderer_amd/csrc` "
            "(or __graft_entry__.build()). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
