"""The five BASELINE.json configurations at their FULL frame sizes (sample counts cut so that the
oracle's share finishes in seconds): sampled rows against the oracle, the default kernel choice
against the forced alternatives, and the 8-GPU row partition against the whole frame."""
import numpy as np
import pytest

import weekend_raytracer_wgpu_amd as m
from helpers import assert_images_equal, full_frame_bands, layer_scene_data, scene_data

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


FULL_SIZE = [  # name, scene, width, height, spp exactly as BASELINE.json names them, chunk of the accumulation, kernel expected
    ("config2", "single_sphere", 1920, 1080, 100, 25, "render_pt_stream_kernel<false>"),
    ("config3", "three_spheres", 1920, 1080, 1000, 250, "render_pt_pool_kernel<256,"),
    ("config4", "earth", 1920, 1080, 1000, 125, "render_pt_pool_kernel<256,"),
    ("config5", "rtiow_final", 3840, 2160, 4000, 500, "render_pt_pool_kernel<1024,"),
]


@pytest.mark.parametrize("name,scene,w,h,spp,chunk,kernel", FULL_SIZE)
def test_configs_at_full_size_through_size_independent_properties(gpu_ctx, name, scene, w, h, spp, chunk, kernel):
    """The BASELINE configurations at FULL size and FULL sample count.  Config 5 -- 3840 x 2160, 4000 samples per pixel, the 8-GPU
    job -- is out of the oracle's reach (one row takes ten minutes on a host thread; rows at 4 spp are held against it above, a
    16-row band at 500 spp by bench.py's verified_rows, bands of configs 2-4 at their full sample counts likewise).  What the
    domain offers at any size: the exact 64-bit radiance sums are ADDITIVE over sample ranges, and the frame does not depend on how
    it is partitioned.  So: (a) spp / chunk accumulations of `chunk` samples == one of spp, sum for sum; (b) that resolves to the
    one-shot image; (c) the 8 ranks' tiles of the full job reassemble exactly that image."""
    sd = scene_data(scene, w, h)
    gpu_ctx.set_scene(sd)
    mk = lambda n, **kw: m.make_params(w, h, n, mode=m.MIRT_MODE_PT, num_bounces=8, **kw)   # noqa: E731
    gpu_ctx.accum_reset(mk(chunk))
    for _ in range(spp // chunk):
        gpu_ctx.accum_add(mk(chunk))
    assert gpu_ctx.accum_samples() == spp
    in_chunks = gpu_ctx.accum_read(mk(chunk))
    resolved = gpu_ctx.accum_resolve(mk(chunk))
    gpu_ctx.accum_reset(mk(spp))
    gpu_ctx.accum_add(mk(spp))
    assert np.array_equal(gpu_ctx.accum_read(mk(spp)), in_chunks), name                  # (a)
    del in_chunks
    full = gpu_ctx.render(mk(spp))
    assert gpu_ctx.last_kernel().startswith(kernel), gpu_ctx.last_kernel()               # the schedule bench.py times
    assert_images_equal(resolved, full, f"{name}: {spp // chunk} x {chunk} spp accumulated vs {spp} spp at once")   # (b)
    base = mk(spp)
    parts = np.zeros((8, m.multi_gpu.max_part_rows(base, 8, 4), w, 4), np.uint8)
    for r in range(8):
        img = gpu_ctx.render(m.multi_gpu.part_params(base, r, 8, 4))
        parts[r, :img.shape[0]] = img
    assert_images_equal(m.multi_gpu.assemble_host(parts, base, 8, 4), full, f"{name}: 8-way tiles at {spp} spp")   # (c)
    gpu_ctx.accum_reset(m.make_params(8, 8, 1, mode=m.MIRT_MODE_PT))                     # give the sums (199 MB at 4K) back


@pytest.mark.parametrize("name,scene,w,h,spp", [("config2", "single_sphere", 1920, 1080, 100), ("config3", "three_spheres", 1920, 1080, 1000),
                                                ("config4", "earth", 1920, 1080, 1000),
                                                # config 5's frame, every pixel, at the smallest sample count that runs ITS kernel (the grid
                                                # build of the pooled kernel, from 16 spp): 1.3 x 10^8 samples through the oracle's flat scan
                                                ("config5 at 16 spp", "rtiow_final", 3840, 2160, 16),
                                                # the reference's own default scene (main.rs:527-545: five spheres, two image textures, all
                                                # four routines), 1080p at its default max_samples_per_pixel (mod.rs:605-613)
                                                ("main.rs scene at 128 spp", "main_rs_scene", 1920, 1080, 128)])
def test_complete_frames_at_the_full_sample_count_against_the_oracle(gpu_ctx, oracle, name, scene, w, h, spp):
    """Configs 2-4 exactly as BASELINE names them (and config 5's 8.3-megapixel frame at 16 samples), the COMPLETE frame: every pixel's exact 64-bit radiance sums from the GPU
    against the oracle's -- 2 x 10^9 samples of config 3, about 20 s of oracle time on the GPU box's host cores (bench.py's
    verified_rows holds 80 of the 1080 rows; this is all of them).  On a host too slow for the whole pass inside two minutes
    (projected from four probe rows) the test does NOT skip: it compares as many 16-row bands as fit that budget, spread over
    the frame, at the full sample count -- never fewer than one band -- and says in its report how many rows it held."""
    import os
    import time
    sd = scene_data(scene, w, h)
    gpu_ctx.set_scene(sd)
    p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8)
    gpu_ctx.accum_reset(p)
    gpu_ctx.accum_add(p)
    got = gpu_ctx.accum_read(p)
    gpu_ctx.accum_reset(m.make_params(8, 8, 1, mode=m.MIRT_MODE_PT))
    t0 = time.perf_counter()
    probe_rows = (0, h // 2 - 1, h // 2, h - 1)
    for rb in probe_rows:
        band = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, row_begin=rb, row_end=rb + 1)
        assert np.array_equal(got[rb:rb + 1], oracle.render_pt_sums(sd, band, n_threads=1)), f"{name}: row {rb}"
    per_row_thread = (time.perf_counter() - t0) / len(probe_rows)
    budget = float(os.environ.get("MIRT_FULL_FRAME_BUDGET_S", "120"))
    bands = full_frame_bands(h, per_row_thread, 2 * oracle.host_cores(), budget)          # (full_frame_bands halves its core count: SMT siblings)
    for first, last in bands:
        band = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, row_begin=first, row_end=last)
        want = oracle.render_pt_sums(sd, band)
        assert np.array_equal(got[first:last], want), f"{name}: rows {first}..{last}: {int((got[first:last] != want).any(-1).sum())} pixels differ"
    held = sum(b - a for a, b in bands)
    assert held >= min(16, h)
    if held < h:                                     # visible with -rA / in the junit report; the test still PASSES on what it compared
        print(f"{name}: slow host ({per_row_thread:.2f} s per row and thread): {held} of {h} rows compared in {len(bands)} bands")


def test_the_reference_render_loop_at_its_operating_point_complete_frame(gpu_ctx, oracle):
    """What the reference's window shows after its default accumulation: main.rs's scene, 1920 x 1080, 2 samples per pixel and frame
    (mod.rs:605-613) up to max_samples_per_pixel = 128, every frame seeded as Raytracer::render_frame seeds it (ONE stream per pixel
    and frame: MirtParams.frame_spp = 2; frame_number starts at 1, mod.rs:284).  64 accumulate calls of 2 samples on the GPU -- the
    reference's own loop shape -- against ONE oracle pass over all 128: the exact sums of every pixel, then the resolved image."""
    w, h, n, total = 1920, 1080, 2, 128
    sd = scene_data("main_rs_scene", w, h)
    gpu_ctx.set_scene(sd)
    mk = lambda spp: m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, frame_spp=n)   # noqa: E731
    gpu_ctx.accum_reset(mk(n))
    for _ in range(total // n):
        gpu_ctx.accum_add(mk(n))
    assert gpu_ctx.accum_samples() == total and gpu_ctx.last_kernel().startswith("render_pt_strip_kernel<")
    got = gpu_ctx.accum_read(mk(n))
    want = oracle.render_pt_sums(sd, mk(total))
    assert np.array_equal(got, want), f"{int((got != want).any(-1).sum())} of {w * h} pixels differ"
    assert_images_equal(gpu_ctx.accum_resolve(mk(n)), oracle.render(sd, mk(total)), "resolved frame")
    gpu_ctx.accum_reset(m.make_params(8, 8, 1, mode=m.MIRT_MODE_PT))
