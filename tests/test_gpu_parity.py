"""Parity proper (MI355X): the HIP path, called through the C ABI, against the CPU oracle and the
committed golden images — parity mode (reference src/raytracer/layer.rs:264-444), u8-exact."""
import sys
from pathlib import Path

import numpy as np
import pytest

import weekend_raytracer_wgpu_amd as m
from helpers import (GOLDEN, assert_images_equal, gradient_image, layer_scene_data, metal_table, scene_data,
                     simple_camera)

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
GOLD = np.load(GOLDEN / "images_v1.npz")


def _render(ctx, sd, p):
    ctx.set_scene(sd)
    return ctx.render(p)


@pytest.mark.parametrize("name", [n for n in GOLD.files if n.startswith("parity_")])
def test_parity_goldens(gpu_ctx, name):
    import make_goldens
    sd, p = make_goldens.case_inputs(name)
    assert_images_equal(_render(gpu_ctx, sd, p), GOLD[name], name)


@pytest.mark.parametrize("w,h,spp", [(160, 120, 1), (160, 120, 2), (160, 120, 20), (160, 120, 21), (200, 150, 63),
                                     (200, 150, 64), (200, 150, 65), (333, 77, 2), (1, 1, 3), (17, 1, 2), (1, 19, 130),
                                     (800, 600, 2)])
def test_layer_scene_vs_oracle(gpu_ctx, oracle, w, h, spp):
    """Default 6-sphere Layer::scene under the default fly camera; ragged sizes; spp around the
    depth-20 cliff and around the 64-lane batch size."""
    sd = layer_scene_data(w, h)
    p = m.make_params(w, h, spp)
    assert_images_equal(_render(gpu_ctx, sd, p), oracle.render(sd, p), f"{w}x{h} spp{spp}")


@pytest.mark.parametrize("w,h,spp", [(160, 120, 2), (333, 77, 21), (200, 150, 40), (17, 1, 2), (800, 600, 2), (192, 108, 100), (96, 54, 1000)])
def test_lane_per_pixel_and_lane_per_sample_schedules_agree(gpu_ctx, oracle, w, h, spp):
    """Parity mode runs lane = pixel at every sample count (round 3: below 64; every lane walks the reference's sample loop for its own
    pixel, layer.rs:320-378); MIRT_FLAG_KERNEL_STRIP forces the lane = sample schedule.  Same image, same work counters,
    and the kernel names say which one ran."""
    sd = layer_scene_data(w, h)
    gpu_ctx.set_scene(sd)
    want = oracle.render(sd, m.make_params(w, h, spp))
    by_pixel = gpu_ctx.render(m.make_params(w, h, spp))
    assert gpu_ctx.last_kernel() == "render_parity_kernel<false,true>"
    by_sample = gpu_ctx.render(m.make_params(w, h, spp, flags=m.MIRT_FLAG_KERNEL_STRIP))
    assert gpu_ctx.last_kernel() == "render_parity_kernel<false,false>"
    assert_images_equal(by_pixel, want, "lane = pixel")
    assert_images_equal(by_sample, want, "lane = sample")
    counts = []
    for extra in (0, m.MIRT_FLAG_KERNEL_STRIP):
        assert_images_equal(gpu_ctx.render(m.make_params(w, h, spp, flags=m.MIRT_FLAG_COUNT_WORK | extra)), want, "counting build")
        st = gpu_ctx.stats()
        counts.append({k: st[k] for k in COUNTED} | {"scatter_metal": st["scatter"][1]})
    assert counts[0] == counts[1]


def test_k1_empty_world(gpu_ctx):
    w, h = 97, 53
    sd = m.SceneData(simple_camera(w, h), [], [], np.zeros((0, 3), np.float32))
    for spp in (1, 21, 100):
        assert_images_equal(_render(gpu_ctx, sd, m.make_params(w, h, spp)), gradient_image(w, h), f"K1 spp{spp}")


def test_k3_k4_depth_cliff_on_gpu(gpu_ctx, oracle):
    w, h = 128, 128
    mats, tex = metal_table()
    sd = m.SceneData(simple_camera(w, h), [m.Sphere.new((0, 0, 0), 1.0, 0).to_c()], mats, tex)
    img20 = _render(gpu_ctx, sd, m.make_params(w, h, 20))
    img21 = _render(gpu_ctx, sd, m.make_params(w, h, 21))
    img500 = _render(gpu_ctx, sd, m.make_params(w, h, 500))
    assert_images_equal(img20, oracle.render(sd, m.make_params(w, h, 20)), "spp20")
    assert_images_equal(img21, oracle.render(sd, m.make_params(w, h, 21)), "spp21")
    assert_images_equal(img500, img21, "K4: spp beyond 21 changes nothing")
    assert (img21[..., :3] == 0).all(-1)[h // 2, w // 2] and img20[h // 2, w // 2, 2] == 255


def test_k5_seed_invariance(gpu_ctx):
    w, h = 160, 120
    sd = layer_scene_data(w, h)
    assert_images_equal(_render(gpu_ctx, sd, m.make_params(w, h, 2, seed=0)),
                        _render(gpu_ctx, sd, m.make_params(w, h, 2, seed=0xABCDEF0123456789)), "K5")


def test_k6_sphere_order_matters(gpu_ctx, oracle):
    w, h = 160, 120
    sd = layer_scene_data(w, h)
    rev = m.SceneData(sd.camera, list(sd.spheres)[::-1], list(sd.materials), sd.texels)
    p = m.make_params(w, h, 2)
    a, b = _render(gpu_ctx, sd, p), _render(gpu_ctx, rev, p)
    assert (a != b).any()
    assert_images_equal(b, oracle.render(rev, p), "reversed list")


def test_k7_row0_texel_quirk(gpu_ctx, oracle):
    w, h = 160, 120
    sd = layer_scene_data(w, h)
    tex2 = sd.texels.copy()
    tex2[sd.materials[2].desc1.offset + 1] = 0.0
    sd2 = m.SceneData(sd.camera, list(sd.spheres), list(sd.materials), tex2)
    p = m.make_params(w, h, 2)
    a, b = _render(gpu_ctx, sd, p), _render(gpu_ctx, sd2, p)
    changed = (a != b).any(-1)
    assert changed[0].any() and not changed[1:].any()
    assert_images_equal(b, oracle.render(sd2, p), "K7")


@pytest.mark.parametrize("seed", range(6))
def test_random_scenes_and_cameras(gpu_ctx, oracle, seed):
    """Random sphere soups (overlapping, nested, behind the camera, tiny/huge radii) and random
    camera poses: every branch of closest_hit_raw / scatter_metal."""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(1, 40))
    mats, tex = metal_table()
    spheres = [m.Sphere.new(rng.normal(size=3) * 3, float(rng.uniform(0.05, 2.5) if rng.random() < 0.9 else 300.0), 0).to_c()
               for _ in range(n)]
    w, h = int(rng.integers(40, 200)), int(rng.integers(30, 150))
    fc = m.FlyCameraController(rng.normal(size=3).astype(np.float32) * 4, m.Angle.degrees(float(rng.uniform(-180, 180))),
                               m.Angle.degrees(float(rng.uniform(-60, 60))), float(rng.uniform(20, 90)), 0.0,
                               float(rng.uniform(1, 10)))
    sd = m.SceneData(m.GpuCamera.new(fc.renderer_camera(), (w, h)).c, spheres, mats, tex)
    for spp in (2, 33):
        p = m.make_params(w, h, spp)
        assert_images_equal(_render(gpu_ctx, sd, p), oracle.render(sd, p), f"seed{seed} {n} spheres {w}x{h} spp{spp}")


def test_bands_and_tiles_reassemble(gpu_ctx):
    w, h = 120, 101
    sd = layer_scene_data(w, h)
    gpu_ctx.set_scene(sd)
    base = m.make_params(w, h, 21)
    full = gpu_ctx.render(base)
    top = gpu_ctx.render(m.make_params(w, h, 21, row_begin=0, row_end=40))
    bot = gpu_ctx.render(m.make_params(w, h, 21, row_begin=40, row_end=0))
    assert_images_equal(np.concatenate([top, bot]), full, "bands")
    for world, tr in [(2, 4), (8, 4), (3, 16), (5, 1)]:
        parts = np.zeros((world, m.multi_gpu.max_part_rows(base, world, tr), w, 4), np.uint8)
        for r in range(world):
            img = gpu_ctx.render(m.multi_gpu.part_params(base, r, world, tr))
            parts[r, :img.shape[0]] = img
        assert_images_equal(m.multi_gpu.assemble_host(parts, base, world, tr), full, f"{world}x{tr}")


def test_layer_object_end_to_end(oracle):
    """The reference-shaped interface: Layer::new -> set_global_data -> set_data -> imgbuf / register_texture."""
    rp = m.RenderParams(camera=m.FlyCameraController.default().renderer_camera(), viewport_size=(200, 150))
    layer = m.Layer.new([200, 150], rp)
    assert layer.imgbuf() is None
    assert layer.set_global_data()
    layer.set_data(rp)
    rgba = layer.register_texture()
    want = oracle.render(layer.scene_data(), m.make_params(200, 150, 2))
    assert_images_equal(rgba, want, "Layer.set_data")
    assert np.array_equal(layer.imgbuf(), want[..., :3]) and (rgba[..., 3] == 255).all()
    # resize re-renders (layer.rs:240-262); update_camera alone does not (layer.rs:188-193)
    rp2 = m.RenderParams(camera=rp.camera, viewport_size=(96, 64))
    layer.resize(rp2)
    assert layer.register_texture().shape == (64, 96, 4)
    assert_images_equal(layer.register_texture(), oracle.render(layer.scene_data(), m.make_params(96, 64, 2)), "resize")
    layer.close()


def test_full_size_parity_properties(gpu_ctx, oracle):
    """BASELINE size 1920x1080: K1 analytically; Layer::scene at spp 1000 == spp 21 (K4); a strided
    sample of rows against the oracle; tiles reassemble to the full frame."""
    w, h = 1920, 1080
    empty = m.SceneData(simple_camera(w, h), [], [], np.zeros((0, 3), np.float32))
    assert_images_equal(_render(gpu_ctx, empty, m.make_params(w, h, 100)), gradient_image(w, h), "K1 1080p")
    sd = layer_scene_data(w, h)
    gpu_ctx.set_scene(sd)
    img1000 = gpu_ctx.render(m.make_params(w, h, 1000))
    img21 = gpu_ctx.render(m.make_params(w, h, 21))
    assert_images_equal(img1000, img21, "K4 1080p")
    for rb in (0, 333, 540, 1079):
        band = m.make_params(w, h, 21, row_begin=rb, row_end=rb + 1)
        assert_images_equal(img21[rb:rb + 1], oracle.render(sd, band), f"row {rb} vs oracle")
    base = m.make_params(w, h, 21)
    parts = np.zeros((8, m.multi_gpu.max_part_rows(base, 8, 4), w, 4), np.uint8)
    for r in range(8):
        img = gpu_ctx.render(m.multi_gpu.part_params(base, r, 8, 4))
        parts[r, :img.shape[0]] = img
    assert_images_equal(m.multi_gpu.assemble_host(parts, base, 8, 4), img21, "8-way tiles 1080p")


COUNTED = ("rays", "sphere_tests", "roots", "hits", "lane_iterations")


@pytest.mark.parametrize("spp", [2, 21, 100])
def test_parity_work_counters_match_oracle(gpu_ctx, oracle, spp):
    """MIRT_FLAG_COUNT_WORK in parity mode counts the work of the reference's SEQUENTIAL sample loop (it returns at
    the first terminating sample) although the kernel computes 64 samples of a pixel at once: rays, sphere tests,
    roots, hit records, scatter_metal calls and samples executed must equal the oracle's tallies exactly."""
    w, h = 200, 150
    sd = layer_scene_data(w, h)
    gpu_ctx.set_scene(sd)
    p = m.make_params(w, h, spp, flags=m.MIRT_FLAG_COUNT_WORK)
    got_img = gpu_ctx.render(p)
    got = gpu_ctx.stats()
    want_img = oracle.render(sd, p, n_threads=4)
    want = oracle.stats()
    assert_images_equal(got_img, want_img, "counting build image")
    for k in COUNTED:
        assert got[k] == want[k], (k, got[k], want[k])
    assert got["scatter"][1] == want["scatter"][1] > 0
    assert got["samples"] == w * h * spp
