"""MIRT_FLAG_TEXEL_TILES — BASELINE configs[3] "LDS texel tiles": the pooled kernel's tile build serves image-texel
fetches from a per-wave LDS window.  Texel VALUES are the same wherever they are read, so every image must equal the
default build's (and the oracle's), bit for bit; the counting build reports how many fetches the tiles served."""
import numpy as np
import pytest

import weekend_raytracer_wgpu_amd as m
from helpers import assert_images_equal, scene_data

pytestmark = pytest.mark.gpu

TILES = m.MIRT_FLAG_TEXEL_TILES


def _pt(w, h, spp, **kw):
    return m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, **kw)


def test_tile_build_runs_and_matches_default_and_oracle(gpu_ctx, oracle):
    w, h, spp = 960, 540, 64
    sd = scene_data("earth", w, h)
    gpu_ctx.set_scene(sd)
    ref = gpu_ctx.render(_pt(w, h, spp))
    assert gpu_ctx.last_kernel().startswith("render_pt_pool_kernel<256,112,")
    got = gpu_ctx.render(_pt(w, h, spp, flags=TILES))
    assert gpu_ctx.last_kernel().startswith("render_pt_pool_tile_kernel<256,112,6,false,false,2,false>"), gpu_ctx.last_kernel()
    assert_images_equal(got, ref, "tile build vs default build")
    rows = _pt(w, h, spp, row_begin=260, row_end=264)
    assert_images_equal(got[260:264], oracle.render(sd, rows), "tile build vs oracle")
    # exact sums and every work counter of the counting tile build equal the oracle's
    cnt = _pt(w, h, spp, flags=TILES | m.MIRT_FLAG_COUNT_WORK, row_begin=240, row_end=300)
    img = gpu_ctx.render(cnt)
    st = gpu_ctx.stats()
    assert gpu_ctx.last_kernel().startswith("render_pt_pool_tile_kernel<256,112,1,true,false,5,false>")
    assert_images_equal(img, got[240:300], "counting tile build")
    oracle.render(sd, cnt)
    ost = oracle.stats()
    for k in ("rays", "sphere_tests", "roots", "hits", "scatter", "sky_misses"):
        assert st[k] == ost[k], k
    f, t = st["texel_fetches"], st["texel_tile_hits"]
    # every lambertian scatter on the earth sphere fetches one image texel; the ground's checker colours are 1x1
    assert f[0] > 0 and f[1] > 0 and t[0] <= f[0] and t[1] <= f[1]
    assert f[0] + f[1] <= st["scatter"][0]
    # camera rays of a strip hit neighbouring texels: the window serves many of them, and few of the later bounces'
    assert t[0] > 0.25 * f[0], (t, f)
    assert t[0] * f[1] > 2 * t[1] * f[0], (t, f)


def test_tile_flag_is_a_hint_elsewhere(gpu_ctx):
    """No image texture, few samples per pixel, many-sphere scenes: the flag changes neither kernel nor image."""
    for scene, w, h, spp in (("three_spheres", 320, 180, 64), ("earth", 320, 180, 8), ("rtiow_final", 320, 180, 64)):
        gpu_ctx.set_scene(scene_data(scene, w, h))
        ref = gpu_ctx.render(_pt(w, h, spp))
        k = gpu_ctx.last_kernel()
        got = gpu_ctx.render(_pt(w, h, spp, flags=TILES))
        assert gpu_ctx.last_kernel() == k, (scene, k, gpu_ctx.last_kernel())
        assert_images_equal(got, ref, scene)


def test_tile_windows_at_texture_borders_and_small_textures(gpu_ctx, oracle):
    """Windows are clamped into the texture; textures smaller than a window are never tiled; u = 1 / v = 0 fetches (j = width,
    i = height: the reference's index runs into the next row / past the texture) go to the table as before."""
    rng = np.random.default_rng(7)
    w, h, spp = 256, 144, 64
    for tw, th in ((64, 32), (33, 5), (32, 4), (16, 16), (31, 64), (1024, 3)):
        sd = scene_data("earth", w, h)
        texels = rng.random((tw * th + 7, 3), dtype=np.float32)
        sd = _retexture(sd, texels, tw, th)
        gpu_ctx.set_scene(sd)
        got = gpu_ctx.render(_pt(w, h, spp, flags=TILES))
        ref = gpu_ctx.render(_pt(w, h, spp))
        assert_images_equal(got, ref, f"{tw}x{th} texture: tile vs default")
        rows = _pt(w, h, spp, row_begin=70, row_end=73)
        assert_images_equal(got[70:73], oracle.render(sd, rows), f"{tw}x{th} texture: tile vs oracle")


def _retexture(sd, texels, tw, th):
    """The earth scene with its image replaced by a tw x th texture (the 1x1 colours keep their place around it)."""
    mats = [m._abi.MirtMaterial.from_buffer_copy(bytes(x)) for x in sd.materials]
    descs = [d for x in mats for d in (x.desc1, x.desc2)]
    image = [d for d in descs if d.width * d.height > 1]
    assert image
    off, n = image[0].offset, image[0].width * image[0].height
    table = np.concatenate([sd.texels[:off], texels, sd.texels[off + n:]]).astype(np.float32)
    for d in descs:
        if d.offset == off and d.width * d.height > 1:
            d.width, d.height = tw, th
        elif d.offset > off and d.width * d.height >= 1:
            d.offset += texels.shape[0] - n
    return m.SceneData(sd.camera, sd.spheres, mats, table, sd.sky)


def test_tile_build_in_progressive_accumulation_and_parts(gpu_ctx, oracle):
    """The tile build behind the accumulation API (exact sums) and behind the multi-GPU row partition."""
    w, h = 320, 180
    sd = scene_data("earth", w, h)
    gpu_ctx.set_scene(sd)
    sums = {}
    for fl in (0, TILES):
        base = _pt(w, h, 64, flags=fl)
        gpu_ctx.accum_reset(base)
        for k in range(2):
            gpu_ctx.accum_add(_pt(w, h, 64, flags=fl, sample_begin=64 * k))
        assert gpu_ctx.last_kernel().startswith("render_pt_pool_tile_kernel<" if fl else "render_pt_pool_kernel<")
        sums[fl] = gpu_ctx.accum_read(base)
    assert np.array_equal(sums[0], sums[TILES])
    assert np.array_equal(sums[0], oracle.render_pt_sums(sd, _pt(w, h, 128)))
    base = _pt(w, h, 64, flags=TILES)
    full = gpu_ctx.render(base)
    parts = np.zeros((4, m.multi_gpu.max_part_rows(base, 4, 4), w, 4), np.uint8)
    for r in range(4):
        img = gpu_ctx.render(m.multi_gpu.part_params(base, r, 4, 4))
        parts[r, :img.shape[0]] = img
    assert_images_equal(m.multi_gpu.assemble_host(parts, base, 4, 4), full, "4-way tiles with the texel-tile build")
