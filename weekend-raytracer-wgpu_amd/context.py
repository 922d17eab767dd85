"""Thin object wrappers over the C ABI: SceneData (the bytes a `Layer` holds) and Context
(one per GPU: resident scene + render calls).  All compute happens behind libmirt.so."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _abi
from ._lib import check, lib


@dataclass
class SceneData:
    """What `Layer` holds when `set_data` runs (reference src/raytracer/layer.rs:37-46), flattened
    to the wire format: GpuCamera, contiguous spheres, GpuMaterial table, [f32;3] texel table."""
    camera: _abi.MirtGpuCamera
    spheres: Sequence[_abi.MirtSphere]
    materials: Sequence[_abi.MirtMaterial]
    texels: np.ndarray                       # float32 [n_texels, 3]
    sky: Optional[_abi.MirtSkyState] = None

    def __post_init__(self) -> None:
        self.texels = np.ascontiguousarray(self.texels, dtype=np.float32).reshape(-1, 3)
        self._c_spheres = (_abi.MirtSphere * max(1, len(self.spheres)))(*self.spheres)
        self._c_mats = (_abi.MirtMaterial * max(1, len(self.materials)))(*self.materials)

    def as_c(self) -> _abi.MirtScene:
        s = _abi.MirtScene()
        s.camera = C.pointer(self.camera)
        s.spheres = C.cast(self._c_spheres, C.POINTER(_abi.MirtSphere))
        s.n_spheres = len(self.spheres)
        s.materials = C.cast(self._c_mats, C.POINTER(_abi.MirtMaterial))
        s.n_materials = len(self.materials)
        s.texels = self.texels.ctypes.data_as(C.POINTER(C.c_float)) if self.texels.size else None
        s.n_texels = self.texels.shape[0]
        s.sky = C.pointer(self.sky) if self.sky is not None else None
        return s


def make_params(width: int, height: int, spp: int, *, mode: int = _abi.MIRT_MODE_PARITY, num_bounces: int = 8,
                flags: int = 0, seed: int = 0, row_begin: int = 0, row_end: int = 0, tile_rows: int = 0,
                n_parts: int = 0, part: int = 0, sample_begin: int = 0, frame_spp: int = 0, frame_begin: int = 0) -> _abi.MirtParams:
    p = _abi.MirtParams()
    p.width, p.height, p.spp, p.num_bounces = width, height, spp, num_bounces
    p.mode, p.flags, p.seed = mode, flags, seed
    p.row_begin, p.row_end = row_begin, row_end
    p.tile_rows, p.n_parts, p.part, p.sample_begin = tile_rows, n_parts, part, sample_begin
    p.frame_spp = frame_spp
    p.frame_begin = frame_begin
    return p


def params_out_rows(params: _abi.MirtParams) -> int:
    return int(lib().mirt_params_out_rows(C.byref(params)))


def params_out_row_index(params: _abi.MirtParams, i: int) -> int:
    return int(lib().mirt_params_out_row_index(C.byref(params), i))


HIP_STREAM_LEGACY = 1      # hipStreamLegacy ((hipStream_t)1, hip_runtime_api.h): the default ("null") stream by handle


def _stream_arg(stream: Optional[int]):
    """`stream` of the *_device / accum_add wrappers -> the C ABI's void* hip_stream.

    None  -> NULL: the context's own (non-blocking) stream.
    int h -> that hipStream_t; h == 0 is the DEFAULT stream (what `torch.cuda.current_stream().cuda_stream`
             returns unless a side stream is current) and is passed as hipStreamLegacy, because a NULL pointer
             already means "the context's own stream" in the C ABI -- sending 0 through would silently move the
             work to a stream that nothing the caller queues afterwards (an RCCL collective, a D2H copy) waits for."""
    if stream is None:
        return None
    return C.c_void_p(int(stream) if int(stream) != 0 else HIP_STREAM_LEGACY)


class Context:
    """mirt_ctx_* : one per HIP device; owns the device-resident scene."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        check(lib().mirt_ctx_create(device, C.byref(self._h)))
        self.device = device
        self._scene: Optional[SceneData] = None

    def close(self) -> None:
        if self._h:
            lib().mirt_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self) -> "Context":
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    def __del__(self) -> None:  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, scene: SceneData) -> None:
        c = scene.as_c()
        check(lib().mirt_ctx_set_scene(self._h, C.byref(c)))
        self._scene = scene

    def set_camera(self, camera: _abi.MirtGpuCamera) -> None:
        check(lib().mirt_ctx_set_camera(self._h, C.byref(camera)))

    def render(self, params: _abi.MirtParams) -> np.ndarray:
        """Render to host memory -> uint8 [rows, width, 4] (RGBA8, top row first)."""
        rows = params_out_rows(params)
        out = np.empty((max(rows, 0), params.width, 4), dtype=np.uint8)
        check(lib().mirt_ctx_render(self._h, C.byref(params), out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def render_device(self, params: _abi.MirtParams, d_ptr: int, nbytes: int, stream: Optional[int] = None) -> None:
        """Render asynchronously into device memory (e.g. a torch uint8 CUDA tensor's data_ptr()) on `stream`
        (see _stream_arg: None = the context's stream, 0 = the default stream, else a hipStream_t handle)."""
        check(lib().mirt_ctx_render_device(self._h, C.byref(params), C.c_void_p(d_ptr), nbytes, _stream_arg(stream)))

    def deinterleave_device(self, params: _abi.MirtParams, d_parts: int, part_stride: int, d_out: int,
                            out_nbytes: int, stream: Optional[int] = None) -> None:
        check(lib().mirt_ctx_deinterleave_device(self._h, C.byref(params), C.c_void_p(d_parts), part_stride,
                                                 C.c_void_p(d_out), out_nbytes, _stream_arg(stream)))

    # ---- progressive accumulation (RenderProgress::next_frame + the shader's image buffer) ----
    def accum_reset(self, params: _abi.MirtParams) -> None:
        check(lib().mirt_ctx_accum_reset(self._h, C.byref(params)))

    def accum_add(self, params: _abi.MirtParams, stream: Optional[int] = None) -> None:
        check(lib().mirt_ctx_accum_add(self._h, C.byref(params), _stream_arg(stream)))

    def accum_samples(self) -> int:
        return int(lib().mirt_ctx_accum_samples(self._h))

    def accum_resolve(self, params: _abi.MirtParams) -> np.ndarray:
        out = np.empty((params_out_rows(params), params.width, 4), dtype=np.uint8)
        check(lib().mirt_ctx_accum_resolve(self._h, C.byref(params), out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def accum_read(self, params: _abi.MirtParams) -> np.ndarray:
        out = np.empty((params_out_rows(params), params.width, 3), dtype=np.uint64)
        check(lib().mirt_ctx_accum_read(self._h, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def selftest_math(self) -> tuple:
        """(sqrt mismatches, reciprocal mismatches) of the fast sequences vs IEEE over all 2^32 floats."""
        out = (C.c_uint64 * 2)()
        check(lib().mirt_ctx_selftest_math(self._h, out))
        return int(out[0]), int(out[1])

    def last_kernel(self) -> str:
        """Name of the render kernel the last render call launched (as rocprofv3 traces spell it)."""
        s = lib().mirt_ctx_last_kernel(self._h)
        return s.decode() if s else ""

    def synchronize(self) -> None:
        check(lib().mirt_ctx_synchronize(self._h))

    def frame_stream(self, index: int) -> int:
        """hipStream_t handle (an int for the `stream` arguments here) of the context's frame stream 0 or 1: two streams on different
        hardware queues for hosts that keep two frames in flight (mirt_ctx_frame_stream)."""
        h = C.c_void_p()
        check(lib().mirt_ctx_frame_stream(self._h, index, C.byref(h)))
        return int(h.value)

    def set_timing(self, enabled: bool) -> None:
        """Kernel timing on (default: every launch carries an event pair, `stats()` reports kernel times) or off (launches carry no
        event unless the context needs one: the reference's interactive frames queue back to back 30 % faster)."""
        check(lib().mirt_ctx_set_timing(self._h, 1 if enabled else 0))

    def stats(self) -> dict:
        st = _abi.MirtStats()
        check(lib().mirt_ctx_get_stats(self._h, C.byref(st)))
        return st.as_dict()
