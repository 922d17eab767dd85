"""Host-side mirror of the reference's interface for the per-pixel path: the names, argument
meaning and error behaviour of `src/raytracer/{mod,layer,texture,angle}.rs` and the default pose of
`src/fly_camera.rs`, with `Layer::set_data` implemented as ONE call into the HIP library.

Nothing here computes pixels: `Layer.set_data` / `Raytracer.render` go through the C ABI
(include/mirt.h) to the gfx950 kernels, and fail if libmirt.so or a GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _abi
from ._lib import MirtError, check, lib
from .context import Context, SceneData, make_params

f32 = np.float32


def _v3(v) -> np.ndarray:
    a = np.asarray(v, dtype=np.float32).reshape(3)
    return a


# ------------------------------------------------------------------------------------------------
# Angle — src/raytracer/angle.rs:1-50
# ------------------------------------------------------------------------------------------------

@dataclass(frozen=True, order=True)
class Angle:
    _radians: float

    @staticmethod
    def degrees(degrees: float) -> "Angle":          # angle.rs:8-12
        return Angle(float(lib().mirt_degrees_to_radians(C.c_float(degrees))))

    @staticmethod
    def radians(radians: float) -> "Angle":          # angle.rs:15-17
        return Angle(float(f32(radians)))

    def as_degrees(self) -> float:                   # angle.rs:20-22
        return float(lib().mirt_radians_to_degrees(C.c_float(self._radians)))

    def as_radians(self) -> float:                   # angle.rs:25-27
        return self._radians

    def clamp(self, lo: "Angle", hi: "Angle") -> "Angle":   # angle.rs:29-39
        r = self._radians
        if r < lo._radians:
            r = lo._radians
        if r > hi._radians:
            r = hi._radians
        return Angle(r)

    def __add__(self, rhs: "Angle") -> "Angle":      # angle.rs:42-50
        return Angle(float(f32(self._radians) + f32(rhs._radians)))


# ------------------------------------------------------------------------------------------------
# Camera / GpuCamera / params — src/raytracer/mod.rs:440-541, 597-613, 681-755
# ------------------------------------------------------------------------------------------------

@dataclass
class Camera:
    eye_pos: np.ndarray
    eye_dir: np.ndarray
    up: np.ndarray
    vfov: Angle
    aperture: float
    focus_distance: float

    def to_c(self) -> _abi.MirtCamera:
        c = _abi.MirtCamera()
        c.eye_pos[:] = _v3(self.eye_pos).tolist()
        c.eye_dir[:] = _v3(self.eye_dir).tolist()
        c.up[:] = _v3(self.up).tolist()
        c.vfov_radians = self.vfov.as_radians()
        c.aperture = self.aperture
        c.focus_distance = self.focus_distance
        return c

    @staticmethod
    def from_c(c: _abi.MirtCamera) -> "Camera":
        return Camera(np.array(c.eye_pos[:], dtype=f32), np.array(c.eye_dir[:], dtype=f32),
                      np.array(c.up[:], dtype=f32), Angle(float(c.vfov_radians)), float(c.aperture),
                      float(c.focus_distance))


class GpuCamera:
    """`GpuCamera::new(&camera, viewport_size)` mod.rs:700-741 (host arithmetic inside libmirt)."""

    def __init__(self, c: _abi.MirtGpuCamera):
        self.c = c

    @staticmethod
    def new(camera: Camera, viewport_size: Tuple[int, int]) -> "GpuCamera":
        out = _abi.MirtGpuCamera()
        cam = camera.to_c()
        check(lib().mirt_camera_new(C.byref(cam), int(viewport_size[0]), int(viewport_size[1]), C.byref(out)))
        return GpuCamera(out)

    def as_array(self) -> np.ndarray:
        return np.frombuffer(bytes(self.c), dtype=np.float32).copy()


@dataclass
class SamplingParams:                                  # mod.rs:597-613
    max_samples_per_pixel: int = 128
    num_samples_per_pixel: int = 2
    num_bounces: int = 8

    def to_c(self) -> _abi.MirtSamplingParams:
        return _abi.MirtSamplingParams(self.max_samples_per_pixel, self.num_samples_per_pixel, self.num_bounces)


@dataclass
class SkyParams:                                       # mod.rs:543-565 (defaults)
    azimuth_degrees: float = 0.0
    zenith_degrees: float = 85.0
    turbidity: float = 4.0
    albedo: Tuple[float, float, float] = (1.0, 1.0, 1.0)


class RenderParamsValidationError(ValueError):
    """mod.rs:396-411; `.kind` is the Rust variant name."""
    KINDS = {
        _abi.MIRT_ERR_MAX_SAMPLES_MULTIPLE: "MaxSampleCountNotMultiple",
        _abi.MIRT_ERR_VIEWPORT_SIZE: "ViewportSize",
        _abi.MIRT_ERR_VFOV_RANGE: "VfovOutOfRange",
        _abi.MIRT_ERR_APERTURE_RANGE: "ApertureOutOfRange",
        _abi.MIRT_ERR_FOCUS_DISTANCE: "FocusDistanceOutOfRange",
        _abi.MIRT_ERR_SKY: "HwSkyModelValidationError",
        _abi.MIRT_ERR_SPP_ZERO: "NumSamplesZero",
    }

    def __init__(self, status: int, message: str):
        self.status = status
        self.kind = self.KINDS.get(status, "Unknown")
        super().__init__(message)


@dataclass
class RenderParams:                                    # mod.rs:442-447
    camera: Camera
    sky: SkyParams = field(default_factory=SkyParams)
    sampling: SamplingParams = field(default_factory=SamplingParams)
    viewport_size: Tuple[int, int] = (800, 600)

    def validate(self) -> None:                        # mod.rs:450-484
        cam, smp = self.camera.to_c(), self.sampling.to_c()
        rc = lib().mirt_validate_render_params(C.byref(cam), C.byref(smp), int(self.viewport_size[0]),
                                               int(self.viewport_size[1]))
        if rc != _abi.MIRT_OK:
            raise RenderParamsValidationError(rc, lib().mirt_last_error().decode())


# ------------------------------------------------------------------------------------------------
# FlyCameraController — default pose + camera_orientation, src/fly_camera.rs:24-64, 227-241
# ------------------------------------------------------------------------------------------------

@dataclass
class FlyCameraController:
    position: np.ndarray
    yaw: Angle
    pitch: Angle
    vfov_degrees: float
    aperture: float
    focus_distance: float

    @staticmethod
    def default() -> "FlyCameraController":           # fly_camera.rs:24-50
        look_from = np.array([-10.0, 2.0, -4.0], dtype=f32)
        look_at = np.array([0.0, 1.0, 0.0], dtype=f32)
        d = (look_at - look_from).astype(f32)
        # glm::magnitude = sqrt((d0*d0 + d1*d1) + d2*d2) in f32
        focus = float(np.sqrt(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]), dtype=f32))
        return FlyCameraController(look_from, Angle.degrees(25.0), Angle.degrees(-10.0), 30.0, 0.8, focus)

    def renderer_camera(self) -> Camera:              # fly_camera.rs:52-64
        out = _abi.MirtCamera()
        pos = (C.c_float * 3)(*_v3(self.position).tolist())
        check(lib().mirt_camera_from_fly_pose(pos, self.yaw.as_radians(), self.pitch.as_radians(),
                                              self.vfov_degrees, self.aperture, self.focus_distance, C.byref(out)))
        return Camera.from_c(out)


# ------------------------------------------------------------------------------------------------
# Scene model — mod.rs:413-438, 757-886; texture.rs:9-78
# ------------------------------------------------------------------------------------------------

class Sphere:
    """`Sphere::new(center, radius, material_idx)` mod.rs:423-431 -> 32-byte wire struct."""

    def __init__(self, center, radius: float, material_idx: int):
        self.center = _v3(center)
        self.radius = float(f32(radius))
        self.material_idx = int(material_idx)

    @staticmethod
    def new(center, radius: float, material_idx: int) -> "Sphere":
        return Sphere(center, radius, material_idx)

    def to_c(self) -> _abi.MirtSphere:
        s = _abi.MirtSphere()
        s.center[:] = [float(self.center[0]), float(self.center[1]), float(self.center[2]), 0.0]
        s.radius = self.radius
        s.material_idx = self.material_idx
        return s


class Texture:
    """texture.rs:9-78: dimensions + `Vec<[f32;3]>` (row-major, top row first)."""

    def __init__(self, dimensions: Tuple[int, int], data: np.ndarray):
        self._dimensions = (int(dimensions[0]), int(dimensions[1]))
        self._data = np.ascontiguousarray(data, dtype=np.float32).reshape(-1, 3)
        assert self._data.shape[0] == self._dimensions[0] * self._dimensions[1]

    @staticmethod
    def new_from_rgb8(rgb8: np.ndarray) -> "Texture":
        """[h][w][3] uint8 -> texels `inv_255 * (p as f32)` exactly as texture.rs:30-41."""
        rgb8 = np.asarray(rgb8, dtype=np.uint8)
        h, w, _ = rgb8.shape
        inv_255 = f32(1.0) / f32(255.0)
        data = (inv_255 * rgb8.astype(np.float32)).astype(np.float32)
        return Texture((w, h), data)

    @staticmethod
    def new_from_image(path: str) -> "Texture":       # texture.rs:21-46
        """Decode an image file.  `.npz` (key `rgb8`) and `.npy` hold already-decoded RGB8 texels; anything else is read
        as a JPEG (the reference passes `ImageFormat::Jpeg`, texture.rs:27-28) and decoded by the library's own decoder
        (csrc/mirt_jpeg.cpp: bit-identical to libjpeg-turbo; DECODER-UNPINNED against the `image` crate's jpeg-decoder,
        +-1 LSB in some texels is possible and nothing in the reference pins either)."""
        p = Path(path)
        if not p.exists():
            raise FileNotFoundError(path)             # TextureError::FileIoError
        if p.suffix == ".npz":
            return Texture.new_from_rgb8(np.load(p)["rgb8"])
        if p.suffix == ".npy":
            return Texture.new_from_rgb8(np.load(p))
        return Texture.new_from_jpeg_bytes(p.read_bytes())

    @staticmethod
    def new_from_jpeg_bytes(data: bytes) -> "Texture":
        """JPEG bytes -> RGB8 (mirt_jpeg_decode_rgb8) -> `inv_255 * (p as f32)` texels (mirt_rgb8_to_texels)."""
        rgb = decode_jpeg(data)
        h, w, _ = rgb.shape
        texels = np.empty((h * w, 3), dtype=np.float32)
        check(lib().mirt_rgb8_to_texels(rgb.ctypes.data_as(C.c_void_p), h * w, texels.ctypes.data_as(C.c_void_p)))
        return Texture((w, h), texels.reshape(h, w, 3))

    @staticmethod
    def new_from_color(color) -> "Texture":           # texture.rs:48-54
        return Texture((1, 1), _v3(color).reshape(1, 3))

    def as_slice(self) -> np.ndarray:
        return self._data

    def dimensions(self) -> Tuple[int, int]:
        return self._dimensions


class Material:
    """`enum Material` mod.rs:433-438."""

    @dataclass
    class Lambertian:
        albedo: Texture

    @dataclass
    class Metal:
        albedo: Texture
        fuzz: float

    @dataclass
    class Dielectric:
        refraction_index: float

    @dataclass
    class Checkerboard:
        even: Texture
        odd: Texture


class TextureDescriptor:
    @staticmethod
    def empty() -> _abi.MirtTextureDescriptor:        # mod.rs:878-886
        return _abi.MirtTextureDescriptor(0, 0, 0xFFFFFFFF)


class GpuMaterial:
    """mod.rs:767-830: flatten one Material into the 32-byte record + append its texels."""

    @staticmethod
    def _append(texture: Texture, global_texture_data: List[np.ndarray]) -> _abi.MirtTextureDescriptor:
        w, h = texture.dimensions()
        offset = sum(a.shape[0] for a in global_texture_data)
        global_texture_data.append(texture.as_slice())
        return _abi.MirtTextureDescriptor(w, h, offset)

    @staticmethod
    def lambertian(albedo: Texture, gtd: List[np.ndarray]) -> _abi.MirtMaterial:
        return _abi.MirtMaterial(0, GpuMaterial._append(albedo, gtd), TextureDescriptor.empty(), 0.0)

    @staticmethod
    def metal(albedo: Texture, fuzz: float, gtd: List[np.ndarray]) -> _abi.MirtMaterial:
        return _abi.MirtMaterial(1, GpuMaterial._append(albedo, gtd), TextureDescriptor.empty(), float(f32(fuzz)))

    @staticmethod
    def dielectric(refraction_index: float) -> _abi.MirtMaterial:
        return _abi.MirtMaterial(2, TextureDescriptor.empty(), TextureDescriptor.empty(), float(f32(refraction_index)))

    @staticmethod
    def checkerboard(even: Texture, odd: Texture, gtd: List[np.ndarray]) -> _abi.MirtMaterial:
        d1 = GpuMaterial._append(even, gtd)
        d2 = GpuMaterial._append(odd, gtd)
        return _abi.MirtMaterial(3, d1, d2, 0.0)


@dataclass
class Scene:                                           # mod.rs:413-416
    spheres: List[Sphere]
    materials: List[object]


def flatten_materials(materials: Sequence[object]) -> Tuple[List[_abi.MirtMaterial], np.ndarray]:
    """`Layer::set_global_data` layer.rs:125-148 (and the identical loop of Raytracer::new,
    mod.rs:160-183).  NOTE the reference's argument-order quirk: the call sites pass (odd, even)
    into a (even, odd) signature, so desc1 is the ODD colour (layer.rs:139-141 vs mod.rs:802-813)."""
    gtd: List[np.ndarray] = []
    out: List[_abi.MirtMaterial] = []
    for m in materials:
        if isinstance(m, Material.Lambertian):
            out.append(GpuMaterial.lambertian(m.albedo, gtd))
        elif isinstance(m, Material.Metal):
            out.append(GpuMaterial.metal(m.albedo, m.fuzz, gtd))
        elif isinstance(m, Material.Dielectric):
            out.append(GpuMaterial.dielectric(m.refraction_index))
        elif isinstance(m, Material.Checkerboard):
            out.append(GpuMaterial.checkerboard(m.odd, m.even, gtd))
        else:
            raise TypeError(f"not a Material: {m!r}")
    texels = np.concatenate(gtd, axis=0) if gtd else np.zeros((0, 3), dtype=np.float32)
    return out, texels


_ASSET_FIXTURES = {
    "assets/moon.jpeg": "moon_1024x512_rgb8.npz",
    "assets/earthmap.jpeg": "earthmap_1024x512_rgb8.npz",
}


def _jpeg_check(rc: int) -> None:
    if rc != 0:
        raise MirtError(rc, (lib().mirt_jpeg_last_error() or b"").decode())


def jpeg_info(data: bytes) -> Tuple[int, int]:
    """(width, height) of a JPEG file's frame header."""
    w, h = C.c_uint32(), C.c_uint32()
    _jpeg_check(lib().mirt_jpeg_info(data, len(data), C.byref(w), C.byref(h)))
    return int(w.value), int(h.value)


def decode_jpeg(data: bytes) -> np.ndarray:
    """JPEG file bytes -> uint8 [h, w, 3] with the library's decoder (no device involved)."""
    w, h = jpeg_info(data)
    out = np.empty((h, w, 3), dtype=np.uint8)
    _jpeg_check(lib().mirt_jpeg_decode_rgb8(data, len(data), out.ctypes.data_as(C.c_void_p), out.nbytes))
    return out


def asset_path(name: str) -> str:
    """Resolve the reference's hard-coded asset paths (layer.rs:97,108) to the decoded texel tables that
    ship as package data (weekend-raytracer-wgpu_amd/assets/, written by tools/make_texture_fixtures.py),
    unless the file itself exists relative to the CWD."""
    if Path(name).exists():
        return name
    return str(Path(__file__).resolve().parent / "assets" / _ASSET_FIXTURES.get(name, name))


# ------------------------------------------------------------------------------------------------
# Layer — src/raytracer/layer.rs:37-282 (UI upload paths omitted: out of scope)
# ------------------------------------------------------------------------------------------------

class Layer:
    def __init__(self, size, render_params: RenderParams, *, device: int = 0, scene: Optional[Scene] = None):
        # layer.rs:49-88
        self.vp_size = [float(f32(size[0])), float(f32(size[1]))]
        self.camera = GpuCamera.new(render_params.camera, (int(self.vp_size[0]), int(self.vp_size[1])))
        sc = scene if scene is not None else Layer.scene()
        self.world: List[Sphere] = list(sc.spheres)
        self.materials = list(sc.materials)
        self.global_texture_data = np.zeros((0, 3), dtype=np.float32)
        self.material_data: List[_abi.MirtMaterial] = []
        self.texture_id = 0
        self._device = device
        self._ctx: Optional[Context] = None
        self._rgba: Optional[np.ndarray] = None
        self.last_stats: Optional[dict] = None

    @staticmethod
    def new(size, render_params: RenderParams, **kw) -> "Layer":
        return Layer(size, render_params, **kw)

    @staticmethod
    def scene() -> Scene:                             # layer.rs:90-123
        materials = [
            Material.Checkerboard(even=Texture.new_from_color((0.5, 0.7, 0.8)),
                                  odd=Texture.new_from_color((0.9, 0.9, 0.9))),
            Material.Lambertian(albedo=Texture.new_from_image(asset_path("assets/moon.jpeg"))),
            Material.Metal(albedo=Texture.new_from_color((1.0, 0.85, 0.57)), fuzz=0.4),
            Material.Dielectric(refraction_index=1.5),
            Material.Lambertian(albedo=Texture.new_from_image(asset_path("assets/earthmap.jpeg"))),
        ]
        spheres = [
            Sphere.new((5.0, 1.2, -1.5), 1.2, 4),
            Sphere.new((0.0, -500.0, -1.0), 500.0, 0),
            Sphere.new((0.0, 1.0, 0.0), 1.0, 3),
            Sphere.new((-5.0, 1.0, 0.0), 1.0, 2),
            Sphere.new((2.0, -1.0, 0.0), 2.0, 3),
            Sphere.new((5.0, 0.8, 1.5), 0.8, 1),
        ]
        return Scene(spheres, materials)

    def set_global_data(self) -> bool:                # layer.rs:125-148
        self.material_data, self.global_texture_data = flatten_materials(self.materials)
        return True

    def scene_data(self) -> SceneData:
        """The bytes handed across the FFI: `world` (Vec<Box<Sphere>>) gathered contiguously."""
        return SceneData(self.camera.c, [s.to_c() for s in self.world], list(self.material_data),
                         self.global_texture_data)

    def set_data(self, render_params: RenderParams) -> None:   # layer.rs:264-282 -> ONE FFI call
        w, h = int(self.vp_size[0]), int(self.vp_size[1])
        if self._ctx is None:
            self._ctx = Context(self._device)
        self._ctx.set_scene(self.scene_data())
        params = make_params(w, h, render_params.sampling.num_samples_per_pixel, mode=_abi.MIRT_MODE_PARITY)
        self._rgba = self._ctx.render(params)
        self.last_stats = self._ctx.stats()

    def register_texture(self) -> np.ndarray:         # layer.rs:150-176: the RGBA8 bytes imgui would receive
        if self._rgba is None:
            raise RuntimeError("set_data has not been called")
        return self._rgba

    def imgbuf(self) -> Optional[np.ndarray]:         # layer.rs:182-186: ImageBuffer<Rgb<u8>>, [h][w][3]
        if self._rgba is None:
            return None
        h, w, _ = self._rgba.shape
        rgb = np.empty((h, w, 3), dtype=np.uint8)
        check(lib().mirt_rgba8_to_rgb8(self._rgba.ctypes.data_as(C.c_void_p), h * w, rgb.ctypes.data_as(C.c_void_p)))
        return rgb

    def update_camera(self, render_params: RenderParams) -> None:   # layer.rs:188-193 (does NOT re-render)
        self.camera = GpuCamera.new(render_params.camera, render_params.viewport_size)

    def resize(self, render_params: RenderParams) -> None:          # layer.rs:240-262
        width, height = render_params.viewport_size
        if self.vp_size[0] != float(width) or self.vp_size[1] != float(height):
            self.vp_size = [float(width), float(height)]
            self.camera = GpuCamera.new(render_params.camera, render_params.viewport_size)
            self.set_data(render_params)

    def close(self) -> None:
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None


# ------------------------------------------------------------------------------------------------
# Raytracer — the path-traced mode behind the names of src/raytracer/mod.rs:20-394
# (wgpu plumbing omitted; `render` = all frames of the progressive loop in one launch)
# ------------------------------------------------------------------------------------------------

class Raytracer:
    def __init__(self, scene: Scene, render_params: RenderParams, *, device: int = 0,
                 sky_state: Optional[_abi.MirtSkyState] = None, reference_stream: bool = False):
        """`reference_stream=True` makes `render_frame` (and `render`) draw the samples of one frame from ONE RNG
        stream per pixel, seeded with the frame number -- the reference's initRng / samplePixel (wgsl:498-502,
        105-122) for `num_samples_per_pixel` samples per frame (MirtParams.frame_spp).  The default keeps one stream
        per sample (frame_number = sample + 1), which does not depend on how samples are grouped into frames.

        The reference's `frame_number` starts at 1 (mod.rs:284), advances with EVERY `render_frame` call -- frames that add nothing
        because the accumulation is complete included (mod.rs:350) -- and is not reset by `render_progress.reset()` (mod.rs:385).
        This class counts the same way (`frame_number`), and an accumulation that starts after k earlier frames seeds its j-th frame
        with frame k + j + 1 (MirtParams.frame_begin = k), so a camera move mid-session continues the reference's streams."""
        render_params.validate()                                   # mod.rs:44-47
        self.reference_stream = reference_stream
        self.render_params = render_params
        self.material_data, self.global_texture_data = flatten_materials(scene.materials)   # mod.rs:160-183
        self.spheres = list(scene.spheres)
        self.camera = GpuCamera.new(render_params.camera, render_params.viewport_size)
        self.sky_state = sky_state
        self._ctx = Context(device)
        self._ctx.set_scene(self.scene_data())
        self.last_stats: Optional[dict] = None
        self._accumulated = None        # RenderProgress (mod.rs:615-679): None = reset pending
        self.frame_number = 1           # mod.rs:284; advanced by every render_frame call (mod.rs:350), never reset
        self._frame_begin = 0           # frames rendered before the current accumulation started

    def scene_data(self) -> SceneData:
        return SceneData(self.camera.c, [s.to_c() for s in self.spheres], list(self.material_data),
                         self.global_texture_data, self.sky_state)

    def _params(self, spp: int, seed: int, flags: int, frame_begin: Optional[int] = None) -> _abi.MirtParams:
        rp = self.render_params
        if self.sky_state is not None:
            flags |= _abi.MIRT_FLAG_SKY_HOSEK
        if frame_begin is None:
            frame_begin = self._frame_begin
        return make_params(rp.viewport_size[0], rp.viewport_size[1], spp, mode=_abi.MIRT_MODE_PT,
                           num_bounces=rp.sampling.num_bounces, flags=flags, seed=seed,
                           frame_spp=rp.sampling.num_samples_per_pixel if self.reference_stream else 0,
                           frame_begin=frame_begin if self.reference_stream else 0)

    def render_frame(self, *, seed: int = 0, flags: int = 0) -> np.ndarray:
        """`Raytracer::render_frame` (mod.rs:303-351) without the wgpu draw: add
        `num_samples_per_pixel` samples per call until `max_samples_per_pixel` is reached
        (`RenderProgress::next_frame`, mod.rs:626-670), return the current estimate as RGBA8."""
        smp = self.render_params.sampling
        if self._accumulated is None:                          # first frame after a reset: clear
            self._frame_begin = self.frame_number - 1          # this accumulation's frames are frame_number, frame_number + 1, ...
            self._ctx.accum_reset(self._params(smp.num_samples_per_pixel, seed, flags))
            self._accumulated = 0
        if self._accumulated + smp.num_samples_per_pixel <= smp.max_samples_per_pixel:
            self._ctx.accum_add(self._params(smp.num_samples_per_pixel, seed, flags))
            self._accumulated += smp.num_samples_per_pixel
        self.frame_number += 1                                 # mod.rs:350: also when the accumulation was complete already
        return self._ctx.accum_resolve(self._params(smp.num_samples_per_pixel, seed, flags))

    def progress(self) -> float:                               # mod.rs:390-393
        return (self._accumulated or 0) / float(self.render_params.sampling.max_samples_per_pixel)

    def set_render_params(self, render_params: RenderParams) -> None:   # mod.rs:353-388
        render_params.validate()
        self.render_params = render_params
        self.camera = GpuCamera.new(render_params.camera, render_params.viewport_size)
        self._ctx.set_camera(self.camera.c)
        self._accumulated = None                               # render_progress.reset() mod.rs:385

    def render(self, *, seed: int = 0, flags: int = 0, frame_begin: int = 0) -> np.ndarray:
        """All `max_samples_per_pixel` samples in one launch -> RGBA8 [h][w][4].

        With `reference_stream=True` the launch is the accumulation a FRESH reference `Raytracer` would converge to: its frames are
        numbered frame_begin + 1, frame_begin + 2, ... with `frame_begin = 0` by default -- not the frame count of this object's
        progressive loop, so the image does not depend on earlier `render_frame` / `set_render_params` calls.  Pass
        `frame_begin=rt.frame_number - 1` for the accumulation the progressive loop would start NOW.  Ignored without
        `reference_stream` (one stream per sample)."""
        params = self._params(self.render_params.sampling.max_samples_per_pixel, seed, flags, frame_begin=frame_begin)
        img = self._ctx.render(params)
        self.last_stats = self._ctx.stats()
        return img

    def close(self) -> None:
        self._ctx.close()


__all__ = ["Angle", "Camera", "GpuCamera", "SamplingParams", "SkyParams", "RenderParams",
           "RenderParamsValidationError", "FlyCameraController", "Sphere", "Texture", "Material",
           "TextureDescriptor", "GpuMaterial", "Scene", "Layer", "Raytracer", "flatten_materials",
           "asset_path", "MirtError", "decode_jpeg", "jpeg_info"]
