"""ctypes mirror of include/mirt.h — the C ABI of the MI355X ray-trace path.

Every struct here is byte-identical to the header (and therefore to the reference's
`#[repr(C)]` Rust types cited there).  tests/test_abi.py checks sizes and offsets.
"""
from __future__ import annotations

import ctypes as C

MIRT_MODE_PARITY = 0
MIRT_MODE_PT = 1
MIRT_MAX_SPP_PER_CALL = 1 << 24

MIRT_FLAG_SKY_HOSEK = 1 << 0
MIRT_FLAG_NO_TONEMAP = 1 << 1
MIRT_FLAG_NO_SRGB = 1 << 2
MIRT_FLAG_COUNT_WORK = 1 << 3
MIRT_FLAG_KERNEL_STRIP = 1 << 4
MIRT_FLAG_KERNEL_POOL = 1 << 5
MIRT_FLAG_NO_GRID = 1 << 6
MIRT_FLAG_COUNT_GRID = 1 << 7
MIRT_FLAG_FAST_MATH = 1 << 8
MIRT_FLAG_TEXEL_TILES = 1 << 9

MIRT_OK = 0
STATUS = {
    0: "MIRT_OK",
    -1: "MIRT_ERR_MAX_SAMPLES_MULTIPLE",
    -2: "MIRT_ERR_VIEWPORT_SIZE",
    -3: "MIRT_ERR_VFOV_RANGE",
    -4: "MIRT_ERR_APERTURE_RANGE",
    -5: "MIRT_ERR_FOCUS_DISTANCE",
    -6: "MIRT_ERR_SKY",
    -10: "MIRT_ERR_NULL_POINTER",
    -11: "MIRT_ERR_SPP_ZERO",
    -12: "MIRT_ERR_BAD_MODE",
    -13: "MIRT_ERR_BAD_ROWS",
    -14: "MIRT_ERR_MATERIAL_INDEX",
    -15: "MIRT_ERR_TEXEL_RANGE",
    -16: "MIRT_ERR_OUT_BUFFER",
    -17: "MIRT_ERR_NO_SCENE",
    -18: "MIRT_ERR_SCENE_TOO_LARGE",
    -19: "MIRT_ERR_FRAME_SPP",
    -20: "MIRT_ERR_NO_DEVICE",
    -21: "MIRT_ERR_HIP",
    -22: "MIRT_ERR_ALLOC",
    -23: "MIRT_ERR_IMAGE_DECODE",
    -24: "MIRT_ERR_SPP_RANGE",
}
for _code, _name in STATUS.items():
    globals()[_name] = _code


class MirtSphere(C.Structure):
    _fields_ = [("center", C.c_float * 4), ("radius", C.c_float),
                ("material_idx", C.c_uint32), ("_pad", C.c_uint32 * 2)]


class MirtTextureDescriptor(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("offset", C.c_uint32)]


class MirtMaterial(C.Structure):
    _fields_ = [("id", C.c_uint32), ("desc1", MirtTextureDescriptor),
                ("desc2", MirtTextureDescriptor), ("x", C.c_float)]


class MirtGpuCamera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("_padding1", C.c_float),
                ("horizontal", C.c_float * 3), ("_padding2", C.c_float),
                ("vertical", C.c_float * 3), ("_padding3", C.c_float),
                ("u", C.c_float * 3), ("_padding4", C.c_float),
                ("v", C.c_float * 3), ("lens_radius", C.c_float),
                ("lower_left_corner", C.c_float * 3), ("_padding5", C.c_float)]


class MirtSkyState(C.Structure):
    _fields_ = [("params", C.c_float * 27), ("radiances", C.c_float * 3),
                ("_padding", C.c_uint32 * 2), ("sun_direction", C.c_float * 4)]


class MirtCamera(C.Structure):
    _fields_ = [("eye_pos", C.c_float * 3), ("eye_dir", C.c_float * 3), ("up", C.c_float * 3),
                ("vfov_radians", C.c_float), ("aperture", C.c_float), ("focus_distance", C.c_float)]


class MirtSamplingParams(C.Structure):
    _fields_ = [("max_samples_per_pixel", C.c_uint32), ("num_samples_per_pixel", C.c_uint32),
                ("num_bounces", C.c_uint32)]


class MirtScene(C.Structure):
    _fields_ = [("camera", C.POINTER(MirtGpuCamera)),
                ("spheres", C.POINTER(MirtSphere)), ("n_spheres", C.c_uint32),
                ("materials", C.POINTER(MirtMaterial)), ("n_materials", C.c_uint32),
                ("texels", C.POINTER(C.c_float)), ("n_texels", C.c_uint64),
                ("sky", C.POINTER(MirtSkyState))]


class MirtParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32),
                ("num_bounces", C.c_uint32), ("mode", C.c_uint32), ("flags", C.c_uint32),
                ("seed", C.c_uint64), ("row_begin", C.c_uint32), ("row_end", C.c_uint32),
                ("tile_rows", C.c_uint32), ("n_parts", C.c_uint32), ("part", C.c_uint32),
                ("sample_begin", C.c_uint32), ("frame_spp", C.c_uint32), ("frame_begin", C.c_uint32)]


class MirtGridPlan(C.Structure):
    _fields_ = [("cell_factor", C.c_float), ("blob_bytes", C.c_uint32), ("n_cells", C.c_uint32), ("n_entries", C.c_uint32),
                ("n_big", C.c_uint32), ("pool_slots", C.c_uint32)]


class MirtStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("kernel_ms_total", C.c_double), ("launches", C.c_uint64),
                ("samples", C.c_uint64), ("rays", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("roots", C.c_uint64), ("hits", C.c_uint64),
                ("scatter", C.c_uint64 * 5), ("sky_misses", C.c_uint64),
                ("lane_iterations", C.c_uint64), ("wave_iterations", C.c_uint64),
                ("grid_cells", C.c_uint64), ("grid_wave_cells", C.c_uint64),
                ("texel_fetches", C.c_uint64 * 2), ("texel_tile_hits", C.c_uint64 * 2)]

    def as_dict(self) -> dict:
        return {"kernel_ms": self.kernel_ms, "kernel_ms_total": self.kernel_ms_total, "launches": self.launches,
                "samples": self.samples, "rays": self.rays,
                "sphere_tests": self.sphere_tests, "roots": self.roots, "hits": self.hits,
                "scatter": list(self.scatter), "sky_misses": self.sky_misses,
                "lane_iterations": self.lane_iterations, "wave_iterations": self.wave_iterations,
                "grid_cells": self.grid_cells, "grid_wave_cells": self.grid_wave_cells,
                "texel_fetches": list(self.texel_fetches), "texel_tile_hits": list(self.texel_tile_hits)}


# every symbol include/mirt.h declares: name -> (restype, argtypes)
_P = C.POINTER
SYMBOLS = {
    "mirt_version": (C.c_uint32, []),
    "mirt_last_error": (C.c_char_p, []),
    "mirt_status_string": (C.c_char_p, [C.c_int]),
    "mirt_validate_render_params": (C.c_int, [_P(MirtCamera), _P(MirtSamplingParams), C.c_uint32, C.c_uint32]),
    "mirt_camera_new": (C.c_int, [_P(MirtCamera), C.c_uint32, C.c_uint32, _P(MirtGpuCamera)]),
    "mirt_camera_from_fly_pose": (C.c_int, [_P(C.c_float), C.c_float, C.c_float, C.c_float, C.c_float,
                                            C.c_float, _P(MirtCamera)]),
    "mirt_degrees_to_radians": (C.c_float, [C.c_float]),
    "mirt_radians_to_degrees": (C.c_float, [C.c_float]),
    "mirt_params_out_rows": (C.c_uint32, [_P(MirtParams)]),
    "mirt_params_out_row_index": (C.c_uint32, [_P(MirtParams), C.c_uint32]),
    "mirt_grid_plan": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, _P(MirtGridPlan)]),
    "mirt_ctx_create": (C.c_int, [C.c_int, _P(C.c_void_p)]),
    "mirt_ctx_destroy": (None, [C.c_void_p]),
    "mirt_ctx_set_scene": (C.c_int, [C.c_void_p, _P(MirtScene)]),
    "mirt_ctx_set_camera": (C.c_int, [C.c_void_p, _P(MirtGpuCamera)]),
    "mirt_ctx_render": (C.c_int, [C.c_void_p, _P(MirtParams), C.c_void_p, C.c_size_t]),
    "mirt_ctx_render_device": (C.c_int, [C.c_void_p, _P(MirtParams), C.c_void_p, C.c_size_t, C.c_void_p]),
    "mirt_ctx_last_kernel": (C.c_char_p, [C.c_void_p]),
    "mirt_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "mirt_ctx_get_stats": (C.c_int, [C.c_void_p, _P(MirtStats)]),
    "mirt_ctx_accum_reset": (C.c_int, [C.c_void_p, _P(MirtParams)]),
    "mirt_ctx_accum_add": (C.c_int, [C.c_void_p, _P(MirtParams), C.c_void_p]),
    "mirt_ctx_accum_samples": (C.c_uint32, [C.c_void_p]),
    "mirt_ctx_accum_resolve": (C.c_int, [C.c_void_p, _P(MirtParams), C.c_void_p, C.c_size_t]),
    "mirt_ctx_accum_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mirt_ctx_selftest_math": (C.c_int, [C.c_void_p, _P(C.c_uint64)]),
    "mirt_ctx_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "mirt_ctx_frame_stream": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "mirt_render": (C.c_int, [_P(MirtScene), _P(MirtParams), C.c_int, C.c_void_p, C.c_size_t]),
    "mirt_rgba8_to_rgb8": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "mirt_jpeg_info": (C.c_int, [C.c_void_p, C.c_size_t, _P(C.c_uint32), _P(C.c_uint32)]),
    "mirt_jpeg_decode_rgb8": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "mirt_rgb8_to_texels": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "mirt_jpeg_last_error": (C.c_char_p, []),
    "mirt_ctx_deinterleave_device": (C.c_int, [C.c_void_p, _P(MirtParams), C.c_void_p, C.c_size_t,
                                               C.c_void_p, C.c_size_t, C.c_void_p]),
}


def bind(lib: C.CDLL, symbols: dict = SYMBOLS) -> None:
    """Attach restype/argtypes; raises AttributeError if the library lacks a declared symbol."""
    for name, (restype, argtypes) in symbols.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
