"""The five BASELINE.json configurations at their FULL frame sizes (sample counts cut so that the
oracle's share finishes in seconds): sampled rows against the oracle, the default kernel choice
against the forced alternatives, and the 8-GPU row partition against the whole frame."""
import numpy as np
import pytest

import weekend_raytracer_wgpu_amd as m
from helpers import assert_images_equal, layer_scene_data, scene_data

pytestmark = pytest.mark.gpu

CONFIGS = [  # name, scene, width, height, spp
    ("config2 single sphere", "single_sphere", 1920, 1080, 100),
    ("config3 three spheres", "three_spheres", 1920, 1080, 48),
    ("config4 earth texture", "earth", 1920, 1080, 48),
    ("config5 RTIOW final", "rtiow_final", 3840, 2160, 4),
]


@pytest.mark.parametrize("name,scene,w,h,spp", CONFIGS)
def test_pt_config_rows_against_oracle(gpu_ctx, oracle, name, scene, w, h, spp):
    sd = scene_data(scene, w, h)
    gpu_ctx.set_scene(sd)
    base = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8)
    full = gpu_ctx.render(base)
    assert full.shape == (h, w, 4) and (full[..., 3] == 255).all()
    for rb in (0, h // 3, (2 * h) // 3 + 1, h - 1):
        band = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, row_begin=rb, row_end=rb + 1)
        assert_images_equal(full[rb:rb + 1], oracle.render(sd, band), f"{name}: row {rb}")
    # every schedule of the library gives the same frame
    for flags in (m.MIRT_FLAG_KERNEL_STRIP, m.MIRT_FLAG_KERNEL_POOL, m.MIRT_FLAG_NO_GRID | m.MIRT_FLAG_KERNEL_STRIP):
        alt = gpu_ctx.render(m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, flags=flags))
        assert_images_equal(alt, full, f"{name}: flags {flags:#x}")
    # 8 ranks' tiles reassemble the frame (what the single RCCL gather carries)
    parts = np.zeros((8, m.multi_gpu.max_part_rows(base, 8, 4), w, 4), np.uint8)
    for r in range(8):
        img = gpu_ctx.render(m.multi_gpu.part_params(base, r, 8, 4))
        parts[r, :img.shape[0]] = img
    assert_images_equal(m.multi_gpu.assemble_host(parts, base, 8, 4), full, f"{name}: 8-way tiles")


def test_config1_plumbing_and_config3_parity_mode(gpu_ctx, oracle):
    """Config 1 (256x256, 1 spp, single unit sphere, parity mode) and the parity-mode companion of config 3
    (the full 6-sphere Layer::scene at 1920x1080x1000 spp): complete frames against the oracle."""
    sd = scene_data("single_sphere", 256, 256)
    p = m.make_params(256, 256, 1)
    gpu_ctx.set_scene(sd)
    assert_images_equal(gpu_ctx.render(p), oracle.render(sd, p), "config 1")
    sd = layer_scene_data(1920, 1080)
    gpu_ctx.set_scene(sd)
    got = gpu_ctx.render(m.make_params(1920, 1080, 1000))
    # the oracle's literal loop would take minutes at 1000 spp; K4 says spp >= 21 changes nothing, so 21 is the same image
    assert_images_equal(got, oracle.render(sd, m.make_params(1920, 1080, 21)), "Layer::scene 1080p parity")
