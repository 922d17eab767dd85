"""Shared scene builders for the tests (host-side data only)."""
from __future__ import annotations

from functools import lru_cache
from pathlib import Path

import numpy as np

import weekend_raytracer_wgpu_amd as m

GOLDEN = Path(__file__).resolve().parent / "golden"


@lru_cache(maxsize=None)
def _flattened(name: str):
    sc, cam = m.scenes.CONFIGS[name]()
    mats, tex = m.flatten_materials(sc.materials)
    return sc, cam, mats, tex


def scene_data(name: str, w: int, h: int) -> "m.SceneData":
    """SceneData of a BASELINE config scene with the camera built for a w x h viewport."""
    sc, cam, mats, tex = _flattened(name)
    return m.SceneData(m.GpuCamera.new(cam, (w, h)).c, [s.to_c() for s in sc.spheres], list(mats), tex)


@lru_cache(maxsize=None)
def _layer_tables():
    rp = m.RenderParams(camera=m.FlyCameraController.default().renderer_camera(), viewport_size=(800, 600))
    layer = m.Layer.new([800, 600], rp)
    layer.set_global_data()
    return layer.world, layer.material_data, layer.global_texture_data


def layer_scene_data(w: int, h: int, camera=None) -> "m.SceneData":
    """`Layer::scene` (layer.rs:90-123) under the default fly camera, for a w x h viewport."""
    world, mats, tex = _layer_tables()
    cam = camera if camera is not None else m.FlyCameraController.default().renderer_camera()
    return m.SceneData(m.GpuCamera.new(cam, (w, h)).c, [s.to_c() for s in world], list(mats), tex)


def simple_camera(w: int, h: int, eye=(0.0, 0.0, 3.0), direction=(0.0, 0.0, -1.0), vfov=60.0, aperture=0.0, focus=3.0):
    cam = m.Camera(np.asarray(eye, np.float32), np.asarray(direction, np.float32), np.asarray((0, 1, 0), np.float32),
                   m.Angle.degrees(vfov), aperture, focus)
    return m.GpuCamera.new(cam, (w, h)).c


def metal_table():
    """Three materials so that material_data[2] exists (layer.rs:345-349)."""
    mats = [m.Material.Metal(albedo=m.Texture.new_from_color((1.0, 0.85, 0.57)), fuzz=0.4) for _ in range(3)]
    return m.flatten_materials(mats)


def gradient_image(w: int, h: int, rows=None) -> np.ndarray:
    """K1: the fall-through colour `vec3_to_rgb8(v*255, u*255, 255)` (layer.rs:380) in f32."""
    ys = np.arange(h, dtype=np.float32) if rows is None else np.asarray(rows, dtype=np.float32)
    xs = np.arange(w, dtype=np.float32)
    v = (ys / np.float32(h)).astype(np.float32) * np.float32(255.0)
    u = (xs / np.float32(w)).astype(np.float32) * np.float32(255.0)
    img = np.empty((len(ys), w, 4), dtype=np.uint8)
    img[..., 0] = np.clip(np.trunc(v), 0, 255).astype(np.uint8)[:, None]
    img[..., 1] = np.clip(np.trunc(u), 0, 255).astype(np.uint8)[None, :]
    img[..., 2] = 255
    img[..., 3] = 255
    return img


def assert_images_equal(a: np.ndarray, b: np.ndarray, what: str = "") -> None:
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if not np.array_equal(a, b):
        diff = (a != b).any(axis=-1)
        ys, xs = np.nonzero(diff)
        first = [(int(x), int(y), a[y, x].tolist(), b[y, x].tolist()) for y, x in list(zip(ys, xs))[:5]]
        raise AssertionError(f"{what}: {int(diff.sum())}/{diff.size} pixels differ (u8-exact required); first: {first}")


def full_frame_bands(h: int, seconds_per_row_thread: float, cores: int, budget_seconds: float, band: int = 16):
    """Row ranges [first, last) the complete-frame test compares: the whole frame as ONE range when the oracle's pass fits the
    budget (rows are spread over half the host's cores, as measured on the GPU box), else as many `band`-row bands as fit --
    at least one --, evenly spread from the first rows to the last."""
    par = max(1.0, cores * 0.5)
    if seconds_per_row_thread * h / par <= budget_seconds:
        return [(0, h)]
    band = min(band, h)
    per_band = seconds_per_row_thread * band / min(par, band)          # a band keeps at most `band` threads busy
    n = max(1, min(int(budget_seconds / max(per_band, 1e-9)), h // band))
    if n == 1:
        first = max(0, h // 2 - band // 2)
        return [(first, first + band)]
    starts = [round(i * (h - band) / (n - 1)) for i in range(n)]
    return [(s0, s0 + band) for s0 in sorted(set(starts))]


def bimodal_soup(seed=0, n_small=1300, n_medium=1100, r=0.05, extent_radii=16.0):
    """A soup whose spheres are either r or 3.9 r (just under the 4-median-radii limit of the grid): at the default cell of 2.5
    median radii a medium sphere spans 4-5 cells per axis and the lists overflow their 16-bit index; at 4 they do not."""
    import numpy as np
    rng = np.random.default_rng(seed)
    rs = np.concatenate([np.full(n_small, r), np.full(n_medium, 3.9 * r)]).astype(np.float32)
    cs = rng.uniform(-extent_radii * r, extent_radii * r, (n_small + n_medium, 3)).astype(np.float32)
    return cs, rs
