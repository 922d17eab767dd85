"""Analytic known-answer tests of the parity-mode oracle, derived from the reference SOURCE
(SURVEY §8c K1-K7).  The reference holds no fixtures for this path ("parity unpinned"); these
properties are what pins the oracle's restatement of src/raytracer/layer.rs:264-444."""
import numpy as np
import pytest

import weekend_raytracer_wgpu_amd as m
from helpers import (assert_images_equal, gradient_image, layer_scene_data, metal_table, scene_data, simple_camera)


def _single(w, h, center=(0.0, 0.0, 0.0), radius=1.0, n_spheres=1):
    mats, tex = metal_table()
    spheres = [m.Sphere.new(center, radius, 0).to_c()] * n_spheres
    return m.SceneData(simple_camera(w, h), spheres, mats, tex)


def test_k1_empty_world_is_the_gradient(oracle):
    """layer.rs:380: no sphere -> every pixel = ((y/h*255) as u8, (x/w*255) as u8, 255)."""
    w, h = 97, 53
    sd = m.SceneData(simple_camera(w, h), [], [], np.zeros((0, 3), np.float32))
    for spp in (1, 2, 21, 100):
        img = oracle.render(sd, m.make_params(w, h, spp))
        assert_images_equal(img, gradient_image(w, h), f"K1 spp={spp}")


def test_k2_single_convex_sphere_low_spp_is_gradient(oracle):
    """layer.rs:357-380: the reflected ray leaves a lone convex sphere -> no second hit -> with
    spp <= 20 the loop falls through to the gradient everywhere (isolated silhouette pixels aside)."""
    w, h = 128, 128
    sd = _single(w, h)
    for spp in (1, 2, 20):
        img = oracle.render(sd, m.make_params(w, h, spp))
        diff = (img != gradient_image(w, h)).any(-1)
        assert diff.mean() < 0.002, f"spp={spp}: {diff.sum()} non-gradient pixels"


def test_k3_depth_exhaustion_blackens_the_disc(oracle):
    """layer.rs:318,333-337: depth=20 is per PIXEL; the 21st primary hit returns black."""
    w, h = 128, 128
    sd = _single(w, h)
    img20 = oracle.render(sd, m.make_params(w, h, 20))
    img21 = oracle.render(sd, m.make_params(w, h, 21))
    black = (img21[..., :3] == 0).all(-1)
    # the unit sphere seen from z=3 with vfov 60 covers a disc of radius ~ 0.3062*h around the centre
    yy, xx = np.mgrid[0:h, 0:w]
    r = np.hypot(xx - w / 2, yy - h / 2)
    assert black[r < 0.28 * h].all()
    assert not black[r > 0.34 * h].any()
    assert_images_equal(img21[~black], img20[~black], "K3 outside the disc")
    assert (img20[black][:, 2] == 255).all()          # at spp=20 those pixels are still gradient


def test_k4_more_samples_beyond_21_change_nothing(oracle):
    w, h = 80, 60
    sd = layer_scene_data(w, h)
    ref = oracle.render(sd, m.make_params(w, h, 21))
    for spp in (22, 42, 64, 65, 200):
        assert_images_equal(oracle.render(sd, m.make_params(w, h, spp)), ref, f"K4 spp={spp}")


def test_k5_seed_does_not_matter_in_parity_mode(oracle):
    """math.rs:107-110: random_f32() <= 2.94e-39, the jitter is numerically zero."""
    w, h = 80, 60
    sd = layer_scene_data(w, h)
    a = oracle.render(sd, m.make_params(w, h, 2, seed=0))
    b = oracle.render(sd, m.make_params(w, h, 2, seed=0xDEADBEEF12345678))
    assert_images_equal(a, b, "K5")


def test_k6_last_hit_wins_not_nearest(oracle):
    """layer.rs:426-439: the world scan keeps the LAST sphere hit in list order."""
    w, h = 160, 120
    sd = layer_scene_data(w, h)
    ref = oracle.render(sd, m.make_params(w, h, 2))
    rev = m.SceneData(sd.camera, list(sd.spheres)[::-1], list(sd.materials), sd.texels)
    other = oracle.render(rev, m.make_params(w, h, 2))
    assert (ref != other).any(-1).mean() > 0.01, "reordering the spheres must change the image"


def test_k6b_two_nested_spheres_pick_the_later_one(oracle):
    """A small sphere in front of a big one along the view axis: whichever is LAST in the list
    provides the hit normal; a nearest-hit tracer would give the same image for both orders."""
    w, h = 64, 64
    mats, tex = metal_table()
    near = m.Sphere.new((0.0, 0.0, 1.0), 0.3, 0).to_c()
    far = m.Sphere.new((0.0, 0.0, -2.0), 1.5, 0).to_c()
    a = oracle.render(m.SceneData(simple_camera(w, h), [near, far], mats, tex), m.make_params(w, h, 2))
    b = oracle.render(m.SceneData(simple_camera(w, h), [far, near], mats, tex), m.make_params(w, h, 2))
    assert (a != b).any()


def test_k7_row0_reads_the_next_texel(oracle):
    """mod.rs:1008-1014 on the 1x1 metal texture: image row 0 has v = 1 -> i = 1 -> idx = 1 ->
    the texel AFTER the metal colour (default scene: earthmap texel (0,0))."""
    w, h = 160, 120
    sd = layer_scene_data(w, h)
    ref = oracle.render(sd, m.make_params(w, h, 2))
    tex2 = sd.texels.copy()
    metal_off = sd.materials[2].desc1.offset
    assert metal_off == 524290 and sd.materials[4].desc1.offset == 524291      # layout of layer.rs:125-148
    tex2[metal_off + 1] = (0.0, 0.0, 0.0)       # blacken the texel after the metal colour
    img = oracle.render(m.SceneData(sd.camera, list(sd.spheres), list(sd.materials), tex2), m.make_params(w, h, 2))
    changed = (img != ref).any(-1)
    assert changed[0].any(), "row 0 must depend on texel offset+1"
    assert not changed[1:].any(), "rows >= 1 must not"
    tex3 = sd.texels.copy()
    tex3[metal_off] = (0.0, 0.0, 0.0)           # blacken the metal colour itself
    img3 = oracle.render(m.SceneData(sd.camera, list(sd.spheres), list(sd.materials), tex3), m.make_params(w, h, 2))
    changed3 = (img3 != ref).any(-1)
    assert not changed3[0].any() and changed3[1:].any()


def test_image_is_vertically_flipped_gradient(oracle):
    """math.rs:4-9 + mod.rs:745-754: v grows downward; the red channel of miss pixels is y/h*255."""
    w, h = 64, 48
    sd = m.SceneData(simple_camera(w, h), [], [], np.zeros((0, 3), np.float32))
    img = oracle.render(sd, m.make_params(w, h, 1))
    assert img[0, 0, 0] == 0 and img[h - 1, 0, 0] == int(np.float32(h - 1) / np.float32(h) * np.float32(255))


def test_pixel_classes_of_default_scene(oracle):
    """Indicative class fractions of SURVEY §8a (float32 emulation of the reference at 800x600):
    miss 20.8 %, hit-then-miss 40.8 % -> gradient 61.6 % at spp 2; black grows by 40.8 % at spp 21."""
    w, h = 800, 600
    sd = layer_scene_data(w, h)
    a = oracle.render(sd, m.make_params(w, h, 2))
    b = oracle.render(sd, m.make_params(w, h, 21))
    grad = (a[..., 2] == 255).mean()
    assert abs(grad - 0.616) < 0.003
    newly_black = ((b[..., :3] == 0).all(-1) & (a[..., 2] == 255)).mean()
    assert abs(newly_black - 0.408) < 0.003
    assert a[..., :3].reshape(-1, 3)[a[..., 2].reshape(-1) != 255].max() <= 51      # dark image: n*127.5*albedo*0.4


def test_faithful_variant_is_bit_identical(oracle):
    w, h = 64, 48
    sd = layer_scene_data(w, h)
    a = oracle.render(sd, m.make_params(w, h, 3), variant=oracle.CLEAN)
    b = oracle.render(sd, m.make_params(w, h, 3), variant=oracle.FAITHFUL)
    assert_images_equal(a, b, "faithful vs clean")


def test_thread_count_does_not_matter(oracle):
    w, h = 64, 48
    sd = layer_scene_data(w, h)
    assert_images_equal(oracle.render(sd, m.make_params(w, h, 2), n_threads=1),
                        oracle.render(sd, m.make_params(w, h, 2), n_threads=4), "threads")
