"""weekend-raytracer-wgpu_amd — MI355X-native per-pixel sphere ray tracer.

Drop-in for ONE hot path of linuxing3/weekend-raytracer-wgpu: `Layer::set_data` and its callees
(reference src/raytracer/layer.rs:264-444), plus the path-traced mode whose behaviours the
reference implements in WGSL (src/raytracer/raytracer.wgsl).  The compute lives in csrc/ (HIP for
gfx950 behind the C ABI of include/mirt.h); this package is the host-side mirror of the
reference's Rust interface.  Import it as `weekend_raytracer_wgpu_amd` (repo-root shim).
"""
from . import _abi
from ._abi import (MIRT_FLAG_COUNT_GRID, MIRT_FLAG_COUNT_WORK, MIRT_FLAG_FAST_MATH, MIRT_FLAG_KERNEL_POOL, MIRT_FLAG_KERNEL_STRIP, MIRT_FLAG_NO_GRID, MIRT_FLAG_NO_SRGB, MIRT_FLAG_NO_TONEMAP, MIRT_FLAG_SKY_HOSEK, MIRT_FLAG_TEXEL_TILES,
                   MIRT_MODE_PARITY, MIRT_MODE_PT)
from ._lib import LIB_PATH, MirtError, lib
from .context import Context, SceneData, make_params, params_out_row_index, params_out_rows
from .raytracer import *  # noqa: F401,F403
from . import scenes

__version__ = "0.3.0"
from . import multi_gpu
