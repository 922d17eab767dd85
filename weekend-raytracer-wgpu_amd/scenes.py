"""The five BASELINE.json configurations as (Scene, Camera) builders — host-side data only.

Scenes 3 and 4 are cut from the reference's own scene tables (src/main.rs:515-547); scenes 1/2 and
5 are builder-defined because the reference has no single-sphere or RTIOW-final scene.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

from .raytracer import (Angle, Camera, FlyCameraController, Material, Scene, Sphere, Texture, asset_path)

f32 = np.float32


def _look_camera(eye, direction, up, vfov_deg: float, aperture: float, focus: float) -> Camera:
    return Camera(np.asarray(eye, dtype=f32), np.asarray(direction, dtype=f32), np.asarray(up, dtype=f32),
                  Angle.degrees(vfov_deg), float(aperture), float(f32(focus)))


def main_rs_materials():
    """The 5-material table shared by src/main.rs:516-536 and layer.rs:91-111."""
    return [
        Material.Checkerboard(even=Texture.new_from_color((0.5, 0.7, 0.8)), odd=Texture.new_from_color((0.9, 0.9, 0.9))),
        Material.Lambertian(albedo=Texture.new_from_image(asset_path("assets/moon.jpeg"))),
        Material.Metal(albedo=Texture.new_from_color((1.0, 0.85, 0.57)), fuzz=0.4),
        Material.Dielectric(refraction_index=1.5),
        Material.Lambertian(albedo=Texture.new_from_image(asset_path("assets/earthmap.jpeg"))),
    ]


def single_sphere() -> Tuple[Scene, Camera]:
    """Configs 1 and 2: unit sphere at the origin; material_data[2] must exist because the CPU
    loop reads it on every primary hit (layer.rs:345-349)."""
    metal = lambda: Material.Metal(albedo=Texture.new_from_color((1.0, 0.85, 0.57)), fuzz=0.4)  # noqa: E731
    materials = [metal(), Material.Lambertian(albedo=Texture.new_from_color((0.5, 0.5, 0.5))), metal()]
    scene = Scene([Sphere.new((0.0, 0.0, 0.0), 1.0, 0)], materials)
    cam = _look_camera((0.0, 0.0, 3.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 60.0, 0.0, 3.0)
    return scene, cam


def three_spheres() -> Tuple[Scene, Camera]:
    """Config 3: the first three spheres of src/main.rs:539-541 (checker ground, glass, metal)
    under the default fly camera (src/fly_camera.rs:24-50)."""
    spheres = [
        Sphere.new((0.0, -500.0, -1.0), 500.0, 0),
        Sphere.new((0.0, 1.0, 0.0), 1.0, 3),
        Sphere.new((-5.0, 1.0, 0.0), 1.0, 2),
    ]
    return Scene(spheres, main_rs_materials()), FlyCameraController.default().renderer_camera()


def main_rs_scene() -> Tuple[Scene, Camera]:
    """The wgpu path's full scene, src/main.rs:515-547 (5 spheres)."""
    spheres = [
        Sphere.new((0.0, -500.0, -1.0), 500.0, 0),
        Sphere.new((0.0, 1.0, 0.0), 1.0, 3),
        Sphere.new((-5.0, 1.0, 0.0), 1.0, 2),
        Sphere.new((5.0, 0.8, 1.5), 0.8, 1),
        Sphere.new((5.0, 1.2, -1.5), 1.2, 4),
    ]
    return Scene(spheres, main_rs_materials()), FlyCameraController.default().renderer_camera()


def earth() -> Tuple[Scene, Camera]:
    """Config 4: the earth-textured sphere of src/main.rs:532-535, 543 over the checker ground."""
    spheres = [
        Sphere.new((0.0, -500.0, -1.0), 500.0, 0),
        Sphere.new((5.0, 1.2, -1.5), 1.2, 4),
    ]
    cam = _look_camera((5.0, 1.2, 3.5), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 30.0, 0.0, 5.0)
    return Scene(spheres, main_rs_materials()), cam


# ---- config 5: RTIOW final scene, generated with the shader's own PCG (raytracer.wgsl:493-511) ----

def _jenkins(x: int) -> int:
    m = 0xFFFFFFFF
    x = (x + (x << 10)) & m
    x ^= x >> 6
    x = (x + (x << 3)) & m
    x ^= x >> 11
    x = (x + (x << 15)) & m
    return x


class _Pcg:
    def __init__(self, seed: int):
        self.state = _jenkins(seed & 0xFFFFFFFF)

    def next(self) -> float:
        m = 0xFFFFFFFF
        old = (self.state + 747796405 + 2891336453) & m
        word = (((old >> ((old >> 28) + 4)) ^ old) * 277803737) & m
        self.state = ((word >> 22) ^ word) & m
        return float(f32(self.state) * f32(2.0 ** -32))


def rtiow_final(seed: int = 0x5EED) -> Tuple[Scene, Camera]:
    """'Ray Tracing in One Weekend' cover scene: ground r1000, a 22x22 grid of r0.2 spheres,
    three r1 spheres; one 1x1 texture per material (~485 spheres)."""
    rng = _Pcg(seed)
    materials = [Material.Lambertian(albedo=Texture.new_from_color((0.5, 0.5, 0.5)))]
    spheres = [Sphere.new((0.0, -1000.0, 0.0), 1000.0, 0)]
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose = rng.next()
            cx = a + 0.9 * rng.next()
            cz = b + 0.9 * rng.next()
            if math.sqrt((cx - 4.0) ** 2 + (0.2 - 0.2) ** 2 + cz ** 2) <= 0.9:
                continue
            if choose < 0.8:
                col = (rng.next() * rng.next(), rng.next() * rng.next(), rng.next() * rng.next())
                materials.append(Material.Lambertian(albedo=Texture.new_from_color(col)))
            elif choose < 0.95:
                col = (0.5 + 0.5 * rng.next(), 0.5 + 0.5 * rng.next(), 0.5 + 0.5 * rng.next())
                materials.append(Material.Metal(albedo=Texture.new_from_color(col), fuzz=0.5 * rng.next()))
            else:
                materials.append(Material.Dielectric(refraction_index=1.5))
            spheres.append(Sphere.new((cx, 0.2, cz), 0.2, len(materials) - 1))
    materials.append(Material.Dielectric(refraction_index=1.5))
    spheres.append(Sphere.new((0.0, 1.0, 0.0), 1.0, len(materials) - 1))
    materials.append(Material.Lambertian(albedo=Texture.new_from_color((0.4, 0.2, 0.1))))
    spheres.append(Sphere.new((-4.0, 1.0, 0.0), 1.0, len(materials) - 1))
    materials.append(Material.Metal(albedo=Texture.new_from_color((0.7, 0.6, 0.5)), fuzz=0.0))
    spheres.append(Sphere.new((4.0, 1.0, 0.0), 1.0, len(materials) - 1))

    eye = np.array([13.0, 2.0, 3.0], dtype=f32)
    direction = (np.zeros(3, dtype=f32) - eye).astype(f32)
    # camera basis as camera_orientation builds it: right = cross(forward, world_up), up = cross(right, forward)
    fwd = direction / np.linalg.norm(direction)
    right = np.cross(fwd, np.array([0.0, 1.0, 0.0], dtype=f32))
    up = np.cross(right, fwd).astype(f32)
    cam = _look_camera(eye, fwd.astype(f32), up, 20.0, 0.1, 10.0)
    return Scene(spheres, materials), cam


CONFIGS = {
    "single_sphere": single_sphere,
    "three_spheres": three_spheres,
    "main_rs_scene": main_rs_scene,
    "earth": earth,
    "rtiow_final": rtiow_final,
}
