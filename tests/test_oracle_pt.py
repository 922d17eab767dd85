"""Path-traced oracle: RNG stream KATs computed independently in Python from the WGSL text,
exactness/partition properties that the HIP path is later held to, and behaviour checks."""
import numpy as np
import pytest

import weekend_raytracer_wgpu_amd as m
from helpers import assert_images_equal, scene_data, simple_camera

M32 = 0xFFFFFFFF


def jenkins(x):                       # raytracer.wgsl:513-521, transcribed independently
    x = (x + (x << 10)) & M32
    x ^= x >> 6
    x = (x + (x << 3)) & M32
    x ^= x >> 11
    x = (x + (x << 15)) & M32
    return x


def pcg_stream(pixel_index, sample, n):   # wgsl:493-511 with frame = sample + 1 (mod.rs:284)
    state = jenkins(pixel_index ^ jenkins(sample + 1))
    out = []
    for _ in range(n):
        old = (state + 747796405 + 2891336453) & M32
        word = (((old >> ((old >> 28) + 4)) ^ old) * 277803737) & M32
        state = ((word >> 22) ^ word) & M32
        out.append(np.float32(state) / np.float32(4294967295.0))     # f32(state) / f32(0xffffffffu)
    return np.array(out, dtype=np.float32)


def test_rng_stream_matches_the_wgsl_generator(oracle):
    for pix, s in [(0, 0), (1, 0), (1919 + 1079 * 1920, 999), (12345, 77)]:
        assert np.array_equal(oracle.rng_stream(pix, s, 0, 16), pcg_stream(pix, s, 16))
    assert not np.array_equal(oracle.rng_stream(5, 5, 1, 8), oracle.rng_stream(5, 5, 0, 8))


def test_sample_ranges_add_exactly(oracle):
    """S2: fixed-point sums are exact, so samples [0,a) + [a,b) == [0,b) bit for bit."""
    w, h = 48, 32
    sd = scene_data("three_spheres", w, h)
    full = oracle.render_pt_sums(sd, m.make_params(w, h, 24, mode=m.MIRT_MODE_PT))
    a = oracle.render_pt_sums(sd, m.make_params(w, h, 10, mode=m.MIRT_MODE_PT))
    b = oracle.render_pt_sums(sd, m.make_params(w, h, 14, mode=m.MIRT_MODE_PT, sample_begin=10))
    assert np.array_equal(full, a + b)


def test_row_partitions_reassemble_the_frame(oracle):
    w, h = 40, 37
    sd = scene_data("three_spheres", w, h)
    base = m.make_params(w, h, 8, mode=m.MIRT_MODE_PT)
    full = oracle.render(sd, base)
    # contiguous bands
    top = oracle.render(sd, m.make_params(w, h, 8, mode=m.MIRT_MODE_PT, row_begin=0, row_end=20))
    bot = oracle.render(sd, m.make_params(w, h, 8, mode=m.MIRT_MODE_PT, row_begin=20, row_end=37))
    assert_images_equal(np.concatenate([top, bot]), full, "bands")
    # tile interleave with a ragged last tile, 1..5 parts
    for world in (2, 3, 5):
        for tile_rows in (1, 4, 16):
            parts = np.zeros((world, m.multi_gpu.max_part_rows(base, world, tile_rows), w, 4), np.uint8)
            total = 0
            for r in range(world):
                p = m.multi_gpu.part_params(base, r, world, tile_rows)
                img = oracle.render(sd, p)
                parts[r, :img.shape[0]] = img
                total += img.shape[0]
            assert total == h
            assert_images_equal(m.multi_gpu.assemble_host(parts, base, world, tile_rows), full, f"{world} parts x {tile_rows}")


def test_zero_bounces_is_black_and_one_bounce_is_sky_only(oracle):
    w, h = 32, 24
    sd = scene_data("single_sphere", w, h)
    lin = m.MIRT_FLAG_NO_TONEMAP | m.MIRT_FLAG_NO_SRGB
    img0 = oracle.render(sd, m.make_params(w, h, 4, mode=m.MIRT_MODE_PT, num_bounces=0, flags=lin))
    assert (img0[..., :3] == 0).all() and (img0[..., 3] == 255).all()
    img1 = oracle.render(sd, m.make_params(w, h, 4, mode=m.MIRT_MODE_PT, num_bounces=1, flags=lin))
    assert (img1[h // 2, w // 2, :3] == 0).all()          # sphere centre: path cut after the hit
    assert (img1[0, 0, :3] > 100).all()                    # corner: sky


def test_empty_world_renders_the_gradient_sky(oracle):
    w, h = 16, 64
    sd = m.SceneData(simple_camera(w, h, vfov=90.0), [], [], np.zeros((0, 3), np.float32))
    img = oracle.render(sd, m.make_params(w, h, 16, mode=m.MIRT_MODE_PT, flags=m.MIRT_FLAG_NO_TONEMAP | m.MIRT_FLAG_NO_SRGB))
    col = img[:, w // 2, :3].astype(int)
    assert (col[:, 2] == 255).all()                        # blue channel of (1-t)+t*1.0
    assert col[0, 0] < col[-1, 0]                          # looking up (top rows) is bluer: red falls with height
    st = oracle.stats()
    assert st["rays"] == st["sky_misses"] == w * h * 16 and st["hits"] == 0


def test_work_counters_are_consistent(oracle):
    w, h = 48, 32
    sd = scene_data("main_rs_scene", w, h)
    oracle.render(sd, m.make_params(w, h, 8, mode=m.MIRT_MODE_PT))
    st = oracle.stats()
    assert st["sphere_tests"] == st["rays"] * 5
    assert st["hits"] == sum(st["scatter"])
    assert st["rays"] == st["hits"] + st["sky_misses"]
    assert st["samples"] == w * h * 8 and st["sky_misses"] <= st["samples"]


def test_seed_changes_the_noise_not_the_picture(oracle):
    w, h = 48, 32
    sd = scene_data("three_spheres", w, h)
    a = oracle.render(sd, m.make_params(w, h, 32, mode=m.MIRT_MODE_PT, seed=0))
    b = oracle.render(sd, m.make_params(w, h, 32, mode=m.MIRT_MODE_PT, seed=1))
    assert (a != b).any()
    assert abs(a[..., :3].mean() - b[..., :3].mean()) < 1.0


def test_hosek_blob_path_runs_and_is_deterministic(oracle):
    """The 144-byte GpuSkyState is opaque input (hw-skymodel is not available): a synthetic blob
    exercises wgsl:154-166,316-343."""
    w, h = 32, 24
    sd = scene_data("single_sphere", w, h)
    sky = m._abi.MirtSkyState()
    for c in range(3):
        vals = [-1.1, -0.3, 0.5, 1.2, -2.5, 0.4, 0.2, 1.5, 0.6]
        for i, v in enumerate(vals):
            sky.params[9 * c + i] = v * (1.0 + 0.1 * c)
        sky.radiances[c] = 1.0 + c
    sky.sun_direction[:] = [0.0, 0.6, 0.8, 0.0]
    sd.sky = sky
    p = m.make_params(w, h, 8, mode=m.MIRT_MODE_PT, flags=m.MIRT_FLAG_SKY_HOSEK)
    a, b = oracle.render(sd, p), oracle.render(sd, p)
    assert_images_equal(a, b, "hosek determinism")
    assert a[..., :3].std() > 0
    sd.sky = None
    with pytest.raises(oracle.OracleError) as e:
        oracle.render(sd, p)
    assert e.value.status_name == "MIRT_ERR_SKY"
