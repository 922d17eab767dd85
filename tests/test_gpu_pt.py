"""Parity proper (MI355X): path-traced mode (behaviours of reference src/raytracer/raytracer.wgsl)
through the C ABI against the CPU oracle: RGBA8 u8-exact, linear (un-tonemapped) output u8-exact,
and the work counters — every branch decision of every path — equal."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

import weekend_raytracer_wgpu_amd as m
from helpers import GOLDEN, assert_images_equal, scene_data, simple_camera

pytestmark = pytest.mark.gpu


def _lane_per_pixel(kernel: str) -> bool:
    """The lane-per-pixel schedule: the strip kernel's BY_PIXEL build, or -- flat scenes from 16 spp on -- its streaming build."""
    return kernel.startswith("render_pt_stream_kernel<") or ("strip" in kernel and kernel.endswith(",true>"))
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
GOLD = np.load(GOLDEN / "images_v1.npz")
LINEAR = m.MIRT_FLAG_NO_TONEMAP | m.MIRT_FLAG_NO_SRGB
COUNTERS = ["rays", "sphere_tests", "roots", "hits", "scatter", "sky_misses"]


def _render(ctx, sd, p):
    ctx.set_scene(sd)
    return ctx.render(p)


@pytest.mark.parametrize("name", [n for n in GOLD.files if n.startswith("pt_")])
def test_pt_goldens(gpu_ctx, name):
    import make_goldens
    sd, p = make_goldens.case_inputs(name)
    assert_images_equal(_render(gpu_ctx, sd, p), GOLD[name], name)


@pytest.mark.parametrize("scene,w,h,spp", [("single_sphere", 96, 64, 8), ("three_spheres", 128, 72, 16),
                                            ("three_spheres", 61, 47, 130), ("earth", 128, 72, 16),
                                            ("main_rs_scene", 128, 72, 70), ("rtiow_final", 64, 36, 4),
                                            ("three_spheres", 1, 1, 1), ("three_spheres", 3, 1, 64), ("earth", 1, 5, 65)])
@pytest.mark.parametrize("flags", [0, LINEAR])
@pytest.mark.parametrize("kernel", [m.MIRT_FLAG_KERNEL_STRIP, m.MIRT_FLAG_KERNEL_POOL])
def test_pt_vs_oracle_images_and_counters(gpu_ctx, oracle, scene, w, h, spp, flags, kernel):
    """Both schedules of the path-traced kernel (strip: wave = 64 samples of a pixel; pool: paths
    sorted by material in LDS) must give the oracle's image and the oracle's work counters."""
    sd = scene_data(scene, w, h)
    flags |= kernel
    p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, flags=flags | m.MIRT_FLAG_COUNT_WORK)
    got = _render(gpu_ctx, sd, p)
    gs = gpu_ctx.stats()
    want = oracle.render(sd, p)
    os_ = oracle.stats()
    assert_images_equal(got, want, f"{scene} {w}x{h} spp{spp} flags{flags}")
    assert {k: gs[k] for k in COUNTERS} == {k: os_[k] for k in COUNTERS}
    assert gs["samples"] == w * h * spp
    # the counting build and the plain build produce the same image
    p2 = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, flags=flags)
    assert_images_equal(gpu_ctx.render(p2), want, "plain build")


@pytest.mark.parametrize("eye", [(0.0, 0.0, 3.0), (-0.0, 0.0, 3.0), (0.5, -0.0, 3.0), (1.5, 0.25, 3.0)])
@pytest.mark.parametrize("kernel,spp", [(m.MIRT_FLAG_KERNEL_STRIP, 6), (m.MIRT_FLAG_KERNEL_STRIP, 70), (m.MIRT_FLAG_KERNEL_POOL, 70)])
def test_pinhole_cameras_skip_the_lens_exactly(gpu_ctx, oracle, eye, kernel, spp):
    """Aperture 0 (BASELINE configs 2 and 4): the kernels take `origin = eye` without evaluating the lens disc, which is
    exact unless an eye component is -0 (then the host leaves the shortcut off); the two lens draws are still consumed.
    The oracle always evaluates the full thin-lens formula (wgsl:456-478).  Eye components of +0 and -0 included; the
    sphere sits on the axes so that hit points with exactly-zero components occur."""
    w, h = 96, 64
    sc, _, mats, tex = __import__("helpers")._flattened("three_spheres")
    cam = simple_camera(w, h, eye=eye, direction=(0.0, 0.0, -1.0), vfov=50.0, aperture=0.0, focus=3.0)
    assert cam.lens_radius == 0.0
    spheres = [m.Sphere.new((0.0, 0.0, 0.0), 1.0, 2).to_c(), m.Sphere.new((0.0, -101.0, 0.0), 100.0, 0).to_c(),
               m.Sphere.new((1.5, 0.0, 0.5), 0.5, 1).to_c()]
    sd = m.SceneData(cam, spheres, list(mats), tex)
    p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=6, flags=kernel | m.MIRT_FLAG_COUNT_WORK)
    got = _render(gpu_ctx, sd, p)
    gs = gpu_ctx.stats()
    want = oracle.render(sd, p)
    os_ = oracle.stats()
    assert_images_equal(got, want, f"pinhole eye {eye} spp {spp}")
    assert {k: gs[k] for k in COUNTERS} == {k: os_[k] for k in COUNTERS}
    assert_images_equal(gpu_ctx.render(m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=6, flags=kernel)), want, "plain build")


@pytest.mark.parametrize("bounces", [0, 1, 2, 4, 10, 300])
@pytest.mark.parametrize("kernel", [m.MIRT_FLAG_KERNEL_STRIP, m.MIRT_FLAG_KERNEL_POOL])
def test_bounce_limits(gpu_ctx, oracle, bounces, kernel):
    """num_bounces = 300 exceeds the pool kernel's 8-bit bounce counter: the library must fall back
    to the strip kernel by itself."""
    w, h = 64, 48
    sd = scene_data("main_rs_scene", w, h)
    p = m.make_params(w, h, 16, mode=m.MIRT_MODE_PT, num_bounces=bounces, flags=LINEAR | kernel | m.MIRT_FLAG_COUNT_WORK)
    got = _render(gpu_ctx, sd, p)
    gs = gpu_ctx.stats()
    assert_images_equal(got, oracle.render(sd, p), f"bounces={bounces}")
    os_ = oracle.stats()
    assert {k: gs[k] for k in COUNTERS} == {k: os_[k] for k in COUNTERS}


def test_one_by_one_texture_edge_cases(gpu_ctx, oracle):
    """The exact shortcut for 1x1 textures (csrc albedo_at): a camera whose rays all lie in the plane
    z = 0 hits the sphere with n.z == 0 exactly, so every left-hemisphere hit has atan2(-0, n.x<0) = pi,
    u = 1.0, j = 1 and must read texel offset+1 (wgsl:377-387 quirk) — the guarded slow path."""
    w, h = 256, 1
    cam = m._abi.MirtGpuCamera()
    cam.eye[:] = [0.0, 0.0, 0.0]
    cam.horizontal[:] = [0.0, 4.0, 0.0]              # sweep in y only
    cam.vertical[:] = [0.0, 0.0, 0.0]
    cam.u[:] = [0.0, 1.0, 0.0]
    cam.v[:] = [0.0, 0.0, 1.0]
    cam.lens_radius = 0.0
    cam.lower_left_corner[:] = [-3.0, -2.0, 0.0]     # looking down -x
    mats, tex = m.flatten_materials([m.Material.Lambertian(m.Texture.new_from_color((0.9, 0.1, 0.1))),
                                     m.Material.Lambertian(m.Texture.new_from_color((0.1, 0.9, 0.1)))])
    # sphere straddling the view axis: hits have n.x of both signs? no: from the origin looking -x the
    # visible cap has n.x > 0; put the eye INSIDE a big sphere instead so that n.x < 0 on the far side
    spheres = [m.Sphere.new((0.0, 0.0, 0.0), 5.0, 0).to_c()]
    sd = m.SceneData(cam, spheres, mats, tex)
    for kernel in (m.MIRT_FLAG_KERNEL_STRIP, m.MIRT_FLAG_KERNEL_POOL):
        p = m.make_params(w, h, 64, mode=m.MIRT_MODE_PT, num_bounces=2, flags=LINEAR | kernel | m.MIRT_FLAG_COUNT_WORK)
        got = _render(gpu_ctx, sd, p)
        gs = gpu_ctx.stats()
        want = oracle.render(sd, p)
        assert_images_equal(got, want, "planar rays")
        os_ = oracle.stats()
        assert {k: gs[k] for k in COUNTERS} == {k: os_[k] for k in COUNTERS}
    # the quirk is really exercised: blackening texel offset+1 changes the picture
    tex2 = tex.copy()
    tex2[1] = 0.0
    sd2 = m.SceneData(cam, spheres, mats, tex2)
    p = m.make_params(w, h, 64, mode=m.MIRT_MODE_PT, num_bounces=2, flags=LINEAR)
    assert (oracle.render(sd2, p) != oracle.render(sd, p)).any()
    assert_images_equal(_render(gpu_ctx, sd2, p), oracle.render(sd2, p), "planar rays, texel+1 changed")
    # south-pole hits (n.y == -1): straight-down rays through the centre of a sphere below the eye
    cam2 = m._abi.MirtGpuCamera()
    cam2.eye[:] = [0.0, 3.0, 0.0]
    cam2.horizontal[:] = [0.0, 0.0, 0.0]
    cam2.vertical[:] = [0.0, 0.0, 0.0]
    cam2.u[:] = [1.0, 0.0, 0.0]
    cam2.v[:] = [0.0, 0.0, 1.0]
    cam2.lower_left_corner[:] = [0.0, 13.0, 0.0]      # direction (0, +10, 0): hits the top... of an enclosing sphere
    sd3 = m.SceneData(cam2, [m.Sphere.new((0.0, 3.0, 0.0), 4.0, 0).to_c()], mats, tex)
    cam3 = m._abi.MirtGpuCamera.from_buffer_copy(bytes(cam2))
    cam3.lower_left_corner[:] = [0.0, -7.0, 0.0]      # direction (0, -10, 0): hit normal (0,-1,0) exactly
    sd4 = m.SceneData(cam3, [m.Sphere.new((0.0, 3.0, 0.0), 4.0, 0).to_c()], mats, tex)
    for sdx in (sd3, sd4):
        p = m.make_params(4, 4, 64, mode=m.MIRT_MODE_PT, num_bounces=3, flags=LINEAR)
        assert_images_equal(_render(gpu_ctx, sdx, p), oracle.render(sdx, p), "polar hits")


@pytest.mark.parametrize("seed", [1, 0xFFFFFFFF, 0x123456789ABCDEF0])
def test_seeds(gpu_ctx, oracle, seed):
    w, h = 64, 48
    sd = scene_data("three_spheres", w, h)
    p = m.make_params(w, h, 16, mode=m.MIRT_MODE_PT, seed=seed)
    assert_images_equal(_render(gpu_ctx, sd, p), oracle.render(sd, p), f"seed={seed:#x}")


def test_sample_begin_offsets_the_stream(gpu_ctx, oracle):
    w, h = 48, 32
    sd = scene_data("three_spheres", w, h)
    p = m.make_params(w, h, 14, mode=m.MIRT_MODE_PT, sample_begin=10, flags=LINEAR)
    assert_images_equal(_render(gpu_ctx, sd, p), oracle.render(sd, p), "sample_begin")


def test_hosek_sky_blob(gpu_ctx, oracle):
    w, h = 64, 48
    sd = scene_data("three_spheres", w, h)
    sky = m._abi.MirtSkyState()
    for c in range(3):
        for i, v in enumerate([-1.1, -0.3, 0.5, 1.2, -2.5, 0.4, 0.2, 1.5, 0.6]):
            sky.params[9 * c + i] = v * (1.0 + 0.1 * c)
        sky.radiances[c] = 1.0 + c
    sky.sun_direction[:] = [0.0, 0.6, 0.8, 0.0]
    sd.sky = sky
    for flags in (m.MIRT_FLAG_SKY_HOSEK, m.MIRT_FLAG_SKY_HOSEK | LINEAR):
        p = m.make_params(w, h, 16, mode=m.MIRT_MODE_PT, flags=flags | m.MIRT_FLAG_COUNT_WORK)
        got = _render(gpu_ctx, sd, p)
        gs = gpu_ctx.stats()
        assert_images_equal(got, oracle.render(sd, p), f"hosek flags={flags}")
        os_ = oracle.stats()
        assert {k: gs[k] for k in COUNTERS} == {k: os_[k] for k in COUNTERS}


def _sky_blob():
    sky = m._abi.MirtSkyState()
    for c in range(3):
        for i, v in enumerate([-1.1, -0.3, 0.5, 1.2, -2.5, 0.4, 0.2, 1.5, 0.6]):
            sky.params[9 * c + i] = v * (1.0 + 0.1 * c)
        sky.radiances[c] = 1.0 + c
    sky.sun_direction[:] = [0.0, 0.6, 0.8, 0.0]
    return sky


@pytest.mark.parametrize("scene,w,h,spp,bounces", [("three_spheres", 131, 23, 16, 8), ("three_spheres", 96, 64, 17, 1), ("main_rs_scene", 70, 41, 24, 2),
                                                     ("single_sphere", 523, 127, 100, 8), ("single_sphere", 397, 170, 401, 50), ("earth", 80, 45, 30, 8)])
def test_streaming_lane_per_pixel_build_vs_oracle(oracle, scene, w, h, spp, bounces, monkeypatch):
    """render_pt_stream_kernel (flat scenes from 16 spp on): a lane whose path has ended starts its pixel's next sample without waiting for
    the other lanes' paths.  Same samples, same exact sums: the frame is the oracle's at every sample count / bounce limit / ragged size,
    with the RTIOW sky and with a Hosek blob, under a seed and a sample offset, through the accumulation buffer and with the reference's
    per-frame stream -- and byte for byte the frame of the plain lane-per-pixel build (MIRT_STREAM=0)."""
    sd = scene_data(scene, w, h)
    ctx = m.Context(0)
    monkeypatch.setenv("MIRT_STREAM", "0")
    plain = m.Context(0)                                       # tuning knobs are read once, in mirt_ctx_create
    monkeypatch.delenv("MIRT_STREAM")
    try:
        for sky in (None, _sky_blob()):
            sd.sky = sky
            ctx.set_scene(sd)
            plain.set_scene(sd)
            flags = m.MIRT_FLAG_SKY_HOSEK if sky is not None else 0
            for kw in (dict(), dict(seed=0xFEEDFACE, sample_begin=7), dict(frame_spp=1), dict(frame_spp=spp)):
                p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=bounces, flags=flags, **kw)
                got = ctx.render(p)
                assert ctx.last_kernel() == ("render_pt_stream_kernel<true>" if sky is not None else "render_pt_stream_kernel<false>")
                assert_images_equal(got, oracle.render(sd, p), f"{scene} {w}x{h} spp {spp} bounces {bounces} {kw} sky {sky is not None}")
                assert np.array_equal(plain.render(p), got) and plain.last_kernel().startswith("render_pt_strip_kernel<false,")
            p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=bounces, flags=flags)
            ctx.accum_reset(p)
            ctx.accum_add(p)
            ctx.accum_add(p)                                   # the second add continues the sample range
            want = oracle.render_pt_sums(sd, m.make_params(w, h, 2 * spp, mode=m.MIRT_MODE_PT, num_bounces=bounces, flags=flags))
            assert np.array_equal(ctx.accum_read(p), want)
    finally:
        ctx.close()
        plain.close()


@pytest.mark.parametrize("kernel,spp", [(m.MIRT_FLAG_KERNEL_POOL, 32), (m.MIRT_FLAG_KERNEL_STRIP, 8), (0, 100)])
def test_hosek_sky_on_a_many_sphere_scene_and_tile_partitions(gpu_ctx, oracle, kernel, spp):
    """The Hosek builds of the grid kernels (pool: candidate lists, ShadeRec; strip: lane per pixel with candidate lists) on RTIOW, and
    the same frame rendered as 3 interleaved tile partitions (absolute rows feed the candidate bound): all equal to the oracle."""
    w, h = 160, 45
    sd = scene_data("rtiow_final", w, h)
    sky = m._abi.MirtSkyState()
    for c in range(3):
        for i, v in enumerate([-1.1, -0.3, 0.5, 1.2, -2.5, 0.4, 0.2, 1.5, 0.6]):
            sky.params[9 * c + i] = v * (1.0 + 0.1 * c)
        sky.radiances[c] = 1.0 + c
    sky.sun_direction[:] = [0.0, 0.6, 0.8, 0.0]
    sd.sky = sky
    gpu_ctx.set_scene(sd)
    for flags in (m.MIRT_FLAG_SKY_HOSEK | kernel, kernel):
        base = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=5, flags=flags)
        want = oracle.render(sd, base)
        assert_images_equal(gpu_ctx.render(base), want, f"whole frame, flags {flags:#x}, {gpu_ctx.last_kernel()}")
        world, tr = 3, 4
        parts = np.zeros((world, m.multi_gpu.max_part_rows(base, world, tr), w, 4), np.uint8)
        for r in range(world):
            img = gpu_ctx.render(m.multi_gpu.part_params(base, r, world, tr))
            parts[r, :img.shape[0]] = img
        assert_images_equal(m.multi_gpu.assemble_host(parts, base, world, tr), want, f"3 tile partitions, flags {flags:#x}")


def test_missing_material_id_and_every_material(gpu_ctx, oracle):
    """All five scatter branches incl. the pink 'missing material' one (wgsl:309-314), textures > 1x1
    on metal and checker, fuzz 0 and 1, ior < 1."""
    rng = np.random.default_rng(7)
    tex_img = (rng.random((8, 16, 3)) * 255).astype(np.uint8)
    T = m.Texture
    mats = [m.Material.Lambertian(T.new_from_rgb8(tex_img)),
            m.Material.Metal(T.new_from_rgb8(tex_img[:4, :4]), 0.0),
            m.Material.Metal(T.new_from_color((0.9, 0.9, 0.9)), 1.0),
            m.Material.Dielectric(1.5), m.Material.Dielectric(0.7),
            m.Material.Checkerboard(even=T.new_from_rgb8(tex_img[:2, :8]), odd=T.new_from_color((0.2, 0.3, 0.4)))]
    gm, texels = m.flatten_materials(mats)
    bogus = m._abi.MirtMaterial(9, m.TextureDescriptor.empty(), m.TextureDescriptor.empty(), 0.0)
    gm.append(bogus)
    spheres = [m.Sphere.new((0, -100.5, 0), 100.0, 5).to_c()] + \
              [m.Sphere.new((-3 + 1.0 * i, 0.0, float(rng.uniform(-1, 1))), 0.5, i).to_c() for i in range(7)]
    w, h = 160, 90
    sd = m.SceneData(simple_camera(w, h, eye=(0, 0.5, 5), vfov=50, aperture=0.2, focus=5), spheres, gm, texels)
    p = m.make_params(w, h, 32, mode=m.MIRT_MODE_PT, flags=m.MIRT_FLAG_COUNT_WORK)
    got = _render(gpu_ctx, sd, p)
    gs = gpu_ctx.stats()
    assert_images_equal(got, oracle.render(sd, p), "all materials")
    os_ = oracle.stats()
    assert {k: gs[k] for k in COUNTERS} == {k: os_[k] for k in COUNTERS}
    assert all(n > 0 for n in gs["scatter"]), gs["scatter"]


_FUZZ0 = int(os.environ.get("MIRT_FUZZ_FIRST_SEED", "0"))        # soak runs beyond the seeds of earlier soaks: MIRT_FUZZ_FIRST_SEED=1000


@pytest.mark.parametrize("seed", range(_FUZZ0, _FUZZ0 + int(os.environ.get("MIRT_FUZZ_SEEDS", "8"))))     # deeper runs: MIRT_FUZZ_SEEDS=300
def test_random_scenes_materials_and_cameras(gpu_ctx, oracle, seed):
    """Fuzz of the path-traced mode: random sphere soups (overlapping, nested, behind the camera, huge), random
    material tables (every routine incl. the missing-material one, random fuzz / refraction index, 1x1 and
    image textures), random cameras (aperture 0 and wide, any pose).  Every schedule of the library must give
    the oracle's exact 64-bit sums -- with 1 to 5 routines present the pool kernel's 4- and 6-queue builds,
    the lane-per-sample and the lane-per-pixel strip schedules (plain and streaming) are all exercised."""
    rng = np.random.default_rng(500 + seed)
    T = m.Texture
    img = (rng.random((6, 10, 3)) * 255).astype(np.uint8)

    def tex():
        return T.new_from_rgb8(img[: int(rng.integers(1, 6)), : int(rng.integers(1, 10))]) if rng.random() < 0.3 \
            else T.new_from_color(tuple(float(x) for x in rng.random(3)))

    kinds = int(rng.integers(1, 6))                       # how many different routines this scene may use
    makers = [lambda: m.Material.Lambertian(tex()), lambda: m.Material.Metal(tex(), float(rng.random())),
              lambda: m.Material.Dielectric(float(rng.uniform(0.6, 2.4))),
              lambda: m.Material.Checkerboard(even=tex(), odd=tex())]
    order = list(rng.permutation(5))[:kinds]              # 4 = the missing-material routine
    mats = [makers[int(rng.choice([k for k in order if k < 4] or [0]))]() for _ in range(int(rng.integers(1, 7)))]
    gm, texels = m.flatten_materials(mats)
    if 4 in order:
        gm.append(m._abi.MirtMaterial(int(rng.integers(4, 100)), m.TextureDescriptor.empty(), m.TextureDescriptor.empty(), 0.0))
    n = int(rng.integers(1, 24))
    spheres = [m.Sphere.new(rng.normal(size=3) * 3, float(rng.uniform(0.05, 2.0) if rng.random() < 0.9 else 200.0),
                            int(rng.integers(0, len(gm)))).to_c() for _ in range(n)]
    w, h = int(rng.integers(24, 90)), int(rng.integers(16, 60))
    fc = m.FlyCameraController(rng.normal(size=3).astype(np.float32) * 4, m.Angle.degrees(float(rng.uniform(-180, 180))),
                               m.Angle.degrees(float(rng.uniform(-60, 60))), float(rng.uniform(20, 90)),
                               0.0 if rng.random() < 0.4 else float(rng.uniform(0.0, 1.0)), float(rng.uniform(1, 10)))
    sd = m.SceneData(m.GpuCamera.new(fc.renderer_camera(), (w, h)).c, spheres, gm, texels)
    gpu_ctx.set_scene(sd)
    for spp, flags in ((3, 0), (11, 0), (24, 0), (64, m.MIRT_FLAG_KERNEL_STRIP), (64, m.MIRT_FLAG_KERNEL_POOL), (50, 0)):     # 24: the streaming build
        bounces = int(rng.integers(1, 10))
        p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=bounces, seed=seed, flags=flags)
        gpu_ctx.accum_reset(p)
        gpu_ctx.accum_add(p)
        want = oracle.render_pt_sums(sd, m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=bounces, seed=seed))
        assert np.array_equal(gpu_ctx.accum_read(p), want), f"seed {seed}: {n} spheres, routines {order}, {w}x{h}, spp {spp}, flags {flags:#x}"


def test_tiles_and_sample_split_reassemble(gpu_ctx):
    w, h = 96, 77
    sd = scene_data("three_spheres", w, h)
    gpu_ctx.set_scene(sd)
    base = m.make_params(w, h, 48, mode=m.MIRT_MODE_PT)
    full = gpu_ctx.render(base)
    for world, tr in [(2, 4), (8, 4), (3, 16)]:
        parts = np.zeros((world, m.multi_gpu.max_part_rows(base, world, tr), w, 4), np.uint8)
        for r in range(world):
            img = gpu_ctx.render(m.multi_gpu.part_params(base, r, world, tr))
            parts[r, :img.shape[0]] = img
        assert_images_equal(m.multi_gpu.assemble_host(parts, base, world, tr), full, f"{world}x{tr}")
    assert_images_equal(gpu_ctx.render(base), full, "determinism")


def test_raytracer_object_end_to_end(oracle):
    scene, cam = m.scenes.three_spheres()
    rp = m.RenderParams(camera=cam, viewport_size=(96, 64), sampling=m.SamplingParams(64, 2, 8))
    rt = m.Raytracer(scene, rp)
    got = rt.render()
    want = oracle.render(rt.scene_data(), m.make_params(96, 64, 64, mode=m.MIRT_MODE_PT, num_bounces=8))
    assert_images_equal(got, want, "Raytracer.render")
    with pytest.raises(m.RenderParamsValidationError):
        rt.set_render_params(m.RenderParams(camera=cam, viewport_size=(96, 64), sampling=m.SamplingParams(64, 5, 8)))
    rt.close()


def test_full_size_pt_properties(gpu_ctx, oracle):
    """BASELINE size (1920x1080; spp cut to 64 so the oracle rows finish in seconds): sampled rows
    vs the oracle, 8-way tile partition == whole frame, counters == analytic identities."""
    w, h, spp = 1920, 1080, 64
    sd = scene_data("three_spheres", w, h)
    gpu_ctx.set_scene(sd)
    base = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, flags=m.MIRT_FLAG_COUNT_WORK)
    full = gpu_ctx.render(base)
    st = gpu_ctx.stats()
    assert st["samples"] == w * h * spp and st["sphere_tests"] == 3 * st["rays"]
    assert st["rays"] == st["hits"] + st["sky_misses"] and st["hits"] == sum(st["scatter"])
    for rb in (0, 400, 700, 1079):
        band = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, row_begin=rb, row_end=rb + 1)
        assert_images_equal(full[rb:rb + 1], oracle.render(sd, band), f"row {rb} vs oracle")
    plain = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT)
    parts = np.zeros((8, m.multi_gpu.max_part_rows(plain, 8, 4), w, 4), np.uint8)
    for r in range(8):
        img = gpu_ctx.render(m.multi_gpu.part_params(plain, r, 8, 4))
        parts[r, :img.shape[0]] = img
    assert_images_equal(m.multi_gpu.assemble_host(parts, plain, 8, 4), full, "8-way tiles 1080p")


@pytest.mark.parametrize("kernel", [m.MIRT_FLAG_KERNEL_STRIP, m.MIRT_FLAG_KERNEL_POOL])
def test_progressive_accumulation_is_exact(gpu_ctx, oracle, kernel):
    """mirt_ctx_accum_*: frames of 10 + 14 + 40 samples add up, bit for bit, to the oracle's exact
    fixed-point sums for 64 samples (the fp32-intermediate check of the parity claim: every sample's
    radiance, quantised to 2^-20, must be identical), and resolve to the one-shot image."""
    w, h = 96, 54
    sd = scene_data("main_rs_scene", w, h)
    gpu_ctx.set_scene(sd)
    mk = lambda spp: m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, flags=kernel)   # noqa: E731
    gpu_ctx.accum_reset(mk(10))
    for spp in (10, 14, 40):
        gpu_ctx.accum_add(mk(spp))
    assert gpu_ctx.accum_samples() == 64
    want_sums = oracle.render_pt_sums(sd, m.make_params(w, h, 64, mode=m.MIRT_MODE_PT, num_bounces=8))
    assert np.array_equal(gpu_ctx.accum_read(mk(10)), want_sums)
    assert_images_equal(gpu_ctx.accum_resolve(mk(10)), oracle.render(sd, m.make_params(w, h, 64, mode=m.MIRT_MODE_PT)), "resolve")
    assert_images_equal(gpu_ctx.accum_resolve(mk(10)), gpu_ctx.render(mk(64)), "one-shot")
    # reset really clears
    gpu_ctx.accum_reset(mk(10))
    assert gpu_ctx.accum_samples() == 0 and not gpu_ctx.accum_read(mk(10)).any()


@pytest.mark.parametrize("spp_per_frame", [1, 2, 5, 9])
def test_few_samples_per_launch_ragged_frame(gpu_ctx, oracle, spp_per_frame):
    """The reference's interactive loop adds 2 samples per pixel and frame (mod.rs:606-611): such launches run
    the lane-per-pixel schedule (64 pixels per wave; round-robin units below 8 spp, dispensed above).  A frame
    whose pixel count is not a multiple of 64 has a ragged last unit; plain and progressive launches, whole
    frame and one band, must all equal the oracle."""
    w, h = 70, 33
    sd = scene_data("main_rs_scene", w, h)
    gpu_ctx.set_scene(sd)
    mk = lambda spp, **kw: m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, **kw)   # noqa: E731
    assert_images_equal(gpu_ctx.render(mk(spp_per_frame)), oracle.render(sd, mk(spp_per_frame)), "one launch")
    band = mk(spp_per_frame, row_begin=5, row_end=12)
    assert_images_equal(gpu_ctx.render(band), oracle.render(sd, band), "band")
    gpu_ctx.accum_reset(mk(spp_per_frame))
    for f in range(3):
        gpu_ctx.accum_add(mk(spp_per_frame))
    assert gpu_ctx.accum_samples() == 3 * spp_per_frame
    assert np.array_equal(gpu_ctx.accum_read(mk(spp_per_frame)), oracle.render_pt_sums(sd, mk(3 * spp_per_frame)))
    assert_images_equal(gpu_ctx.accum_resolve(mk(spp_per_frame)), oracle.render(sd, mk(3 * spp_per_frame)), "three frames")


@pytest.mark.parametrize("groups", [1, 2, 3])
def test_lane_per_pixel_units_with_sample_groups(oracle, groups, monkeypatch):
    """The lane-per-pixel strip kernel deals the samples of a unit's pixels to 1 / 2 / 4 groups of lanes (64 / 32 / 16 pixels per
    unit) when that gives a short launch enough units; MIRT_PX_GROUPS forces a grouping.  Exact integer sums: every grouping gives
    the oracle's image (ragged frames included) and, through the accumulation buffer, the oracle's sums."""
    monkeypatch.setenv("MIRT_PX_GROUPS", str(groups))
    ctx = m.Context(0)
    try:
        for scene, w, h, spp, flags in (("single_sphere", 200, 37, 48, 0), ("three_spheres", 131, 23, 32, m.MIRT_FLAG_KERNEL_STRIP),
                                        ("single_sphere", 64, 2, 8, 0), ("rtiow_final", 96, 20, 8, 0)):
            sd = scene_data(scene, w, h)
            ctx.set_scene(sd)
            p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=6, flags=flags)
            got = ctx.render(p)
            assert _lane_per_pixel(ctx.last_kernel()), ctx.last_kernel()
            assert_images_equal(got, oracle.render(sd, p), f"{scene} {w}x{h} spp {spp}, {1 << (groups - 1)} sample groups")
        # progressive accumulation through the same kernel: two adds of 16 samples == the oracle's sums of 32
        sd = scene_data("three_spheres", 70, 9)
        ctx.set_scene(sd)
        p16 = m.make_params(70, 9, 16, mode=m.MIRT_MODE_PT, num_bounces=6, flags=m.MIRT_FLAG_KERNEL_STRIP)
        ctx.accum_reset(p16)
        ctx.accum_add(p16)
        ctx.accum_add(p16)
        want = oracle.render_pt_sums(sd, m.make_params(70, 9, 32, mode=m.MIRT_MODE_PT, num_bounces=6))
        assert np.array_equal(ctx.accum_read(p16), want)
    finally:
        ctx.close()


def test_raytracer_render_frame_progression(oracle):
    """The reference's progressive loop (mod.rs:626-670): N spp per frame until max, then frames stop adding."""
    scene, cam = m.scenes.three_spheres()
    rp = m.RenderParams(camera=cam, viewport_size=(64, 40), sampling=m.SamplingParams(12, 4, 8))
    rt = m.Raytracer(scene, rp)
    sd = rt.scene_data()
    for k in (4, 8, 12, 12, 12):
        img = rt.render_frame()
        assert_images_equal(img, oracle.render(sd, m.make_params(64, 40, k, mode=m.MIRT_MODE_PT, num_bounces=8)), f"after {k} spp")
    assert rt.progress() == 1.0
    rt.set_render_params(rp)               # any parameter change resets the accumulation (mod.rs:385)
    assert rt.progress() == 0.0
    assert_images_equal(rt.render_frame(), oracle.render(sd, m.make_params(64, 40, 4, mode=m.MIRT_MODE_PT, num_bounces=8)), "after reset")
    rt.close()


def _sphere_soup(rng, n, spread, r_lo, r_hi, n_big=2):
    T = m.Texture
    mats = [m.Material.Lambertian(T.new_from_color(rng.random(3))), m.Material.Metal(T.new_from_color(0.5 + 0.5 * rng.random(3)), 0.3),
            m.Material.Dielectric(1.5), m.Material.Checkerboard(even=T.new_from_color((0.2, 0.3, 0.1)), odd=T.new_from_color((0.9, 0.9, 0.9)))]
    gm, tex = m.flatten_materials(mats)
    spheres = [m.Sphere.new((0.0, -1000.0, 0.0), 1000.0, 3).to_c()]
    for _ in range(n_big):
        spheres.append(m.Sphere.new(rng.normal(size=3) * spread * 0.3 + (0, 1.5, 0), 1.5, int(rng.integers(0, 3))).to_c())
    for _ in range(n):
        c = rng.normal(size=3) * spread
        c[1] = abs(c[1]) * 0.3 + 0.1
        spheres.append(m.Sphere.new(c, float(rng.uniform(r_lo, r_hi)), int(rng.integers(0, 4))).to_c())
    return spheres, gm, tex


@pytest.mark.parametrize("seed,n,spread", [(0, 40, 3.0), (1, 200, 6.0), (2, 484, 8.0), (3, 1000, 5.0), (4, 64, 0.5)])
def test_grid_nearest_hit_equals_flat_scan(gpu_ctx, oracle, seed, n, spread):
    """Many-sphere scenes go through the uniform grid (csrc nearest_hit_grid).  The image must equal the
    oracle's flat scan bit for bit — overlapping, nested and duplicated spheres included (duplicates force
    the index tie-break) — and must equal the library's own flat-scan build (MIRT_FLAG_NO_GRID)."""
    rng = np.random.default_rng(1000 + seed)
    spheres, gm, tex = _sphere_soup(rng, n, spread, 0.05, 0.35)
    spheres += spheres[5:15]                               # exact duplicates, later in the list
    w, h, spp = 96, 54, 8
    fc = m.FlyCameraController(np.float32([spread * 1.5, 1.5 + seed, spread * 1.2]), m.Angle.degrees(215.0 + 7 * seed),
                               m.Angle.degrees(-12.0), 40.0, 0.05, float(spread * 1.8))
    sd = m.SceneData(m.GpuCamera.new(fc.renderer_camera(), (w, h)).c, spheres, gm, tex)
    gpu_ctx.set_scene(sd)
    p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=6, flags=LINEAR)
    want = oracle.render(sd, p)
    got_grid = gpu_ctx.render(p)
    t_grid = gpu_ctx.stats()["kernel_ms"]
    got_flat = gpu_ctx.render(m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=6, flags=LINEAR | m.MIRT_FLAG_NO_GRID))
    t_flat = gpu_ctx.stats()["kernel_ms"]
    assert_images_equal(got_flat, want, "flat scan")
    assert_images_equal(got_grid, want, f"grid ({t_grid:.2f} ms vs flat {t_flat:.2f} ms)")
    print(f"grid {t_grid:.2f} ms, flat {t_flat:.2f} ms")


@pytest.mark.parametrize("case", ["rtiow", "row_of_spheres_along_the_view", "wide_lens", "narrow_frame", "inside_a_sphere", "behind_and_beside"])
def test_camera_ray_candidate_lists(gpu_ctx, oracle, case):
    """Pooled kernel, grid build: the camera rays of a strip scan a per-strip list of the spheres their bundle can touch instead of
    walking the grid (csrc strip_candidates).  The list must be a superset of what any of those rays hits -- images and every work
    counter equal to the oracle's flat scan -- including the fall-backs: more than 16 candidates (spheres lined up along the view),
    a lens too wide for a thin bundle, strips that wrap into the next image row, a camera inside a sphere, spheres behind the eye
    and just beside the bundle."""
    rng = np.random.default_rng(77)
    spheres, gm, tex = _sphere_soup(rng, 120, 4.0, 0.08, 0.3)
    w, h, spp, aperture, eye, yaw, pitch = 160, 48, 24, 0.05, (9.0, 1.2, 7.0), 218.0, -7.0
    if case == "rtiow":
        sd = scene_data("rtiow_final", 192, 64)
        w, h = 192, 64
    else:
        if case == "row_of_spheres_along_the_view":
            direction = np.float32([np.cos(np.radians(pitch)) * np.sin(np.radians(yaw)), np.sin(np.radians(pitch)), np.cos(np.radians(pitch)) * np.cos(np.radians(yaw))])
            spheres += [m.Sphere.new(np.float32(eye) + direction * (2.0 + 0.5 * i) + rng.normal(size=3) * 0.02, 0.05, int(i % 3)).to_c() for i in range(40)]
        if case == "wide_lens":
            aperture = 1.0
        if case == "narrow_frame":
            w, h = 24, 90                              # 24 pixels per row: every other 16-pixel strip wraps into the next row
        if case == "inside_a_sphere":
            spheres.append(m.Sphere.new(eye, 0.8, 2).to_c())           # the eye sits at the centre of a glass sphere
        if case == "behind_and_beside":
            spheres += [m.Sphere.new((eye[0] + 0.3 * i, eye[1] + 0.2, eye[2] + 0.4 * i), 0.25, 1).to_c() for i in range(1, 8)]
        fc = m.FlyCameraController(np.float32(eye), m.Angle.degrees(yaw), m.Angle.degrees(pitch), 35.0, aperture, 9.0)
        sd = m.SceneData(m.GpuCamera.new(fc.renderer_camera(), (w, h)).c, spheres, gm, tex)
    gpu_ctx.set_scene(sd)
    p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=5, flags=m.MIRT_FLAG_KERNEL_POOL)
    want = oracle.render(sd, m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=5, flags=m.MIRT_FLAG_COUNT_WORK))
    want_counts = oracle.stats()
    got = gpu_ctx.render(p)
    assert gpu_ctx.last_kernel().startswith("render_pt_pool_kernel<1024,") and ",1,true," in gpu_ctx.last_kernel()
    assert_images_equal(got, want, case)
    pc = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=5, flags=m.MIRT_FLAG_KERNEL_POOL | m.MIRT_FLAG_COUNT_WORK | m.MIRT_FLAG_COUNT_GRID)
    assert_images_equal(gpu_ctx.render(pc), want, case + " (counting build)")
    st = gpu_ctx.stats()
    for k in ("rays", "hits", "sky_misses", "scatter"):                   # the paths are the flat scan's, ray by ray
        assert st[k] == want_counts[k], (k, st[k], want_counts[k])


def test_a_scene_whose_lists_overflow_at_the_default_cell_keeps_its_grid(gpu_ctx, oracle):
    """a bimodal soup (half the spheres just under 4 median radii) lists more than 65 535 entries at the default cell of
    2.5 median radii; the builder retries coarser instead of falling back to the flat scan (tests/test_abi.py holds the plan).  The
    blob leaves no room for path pools, so the strip kernel's grid build renders it -- and must give the oracle's flat-scan image."""
    from helpers import bimodal_soup
    cs, rs = bimodal_soup()
    mats, tex = m.flatten_materials([m.Material.Lambertian(albedo=m.Texture.new_from_color((0.6, 0.5, 0.4))),
                                     m.Material.Metal(albedo=m.Texture.new_from_color((0.8, 0.8, 0.9)), fuzz=0.1),
                                     m.Material.Dielectric(refraction_index=1.5)])
    spheres = [m.Sphere.new(tuple(float(x) for x in c), float(r), i % 3).to_c() for i, (c, r) in enumerate(zip(cs, rs))]
    w, h = 64, 36
    sd = m.SceneData(simple_camera(w, h, eye=(0.0, 0.3, 3.0), direction=(0.0, -0.05, -1.0)), spheres, mats, tex)
    gpu_ctx.set_scene(sd)
    p = m.make_params(w, h, 4, mode=m.MIRT_MODE_PT, num_bounces=4)
    got = gpu_ctx.render(p)
    assert gpu_ctx.last_kernel().startswith("render_pt_strip_kernel<false,false,true,"), gpu_ctx.last_kernel()      # GRID = true
    assert_images_equal(got, oracle.render(sd, p), "bimodal soup through the grid at cell = 4")


def test_grid_with_rays_parallel_to_axes(gpu_ctx, oracle):
    """Axis-aligned rays (zero direction components) through the grid: the DDA's 1/0 handling."""
    rng = np.random.default_rng(77)
    spheres, gm, tex = _sphere_soup(rng, 150, 4.0, 0.1, 0.3)
    cam = m._abi.MirtGpuCamera()
    cam.eye[:] = [0.0, 0.4, 12.0]
    cam.horizontal[:] = [8.0, 0.0, 0.0]
    cam.vertical[:] = [0.0, 0.0, 0.0]                     # every ray has d.y == 0 exactly ... and the centre column d.x == 0
    cam.u[:] = [1.0, 0.0, 0.0]
    cam.v[:] = [0.0, 1.0, 0.0]
    cam.lower_left_corner[:] = [-4.0, 0.4, 0.0]
    sd = m.SceneData(cam, spheres, gm, tex)
    p = m.make_params(65, 3, 8, mode=m.MIRT_MODE_PT, num_bounces=4, flags=LINEAR)
    assert_images_equal(_render(gpu_ctx, sd, p), oracle.render(sd, p), "axis-parallel rays")


@pytest.mark.parametrize("kernel", [m.MIRT_FLAG_KERNEL_STRIP, m.MIRT_FLAG_KERNEL_POOL])
def test_grid_counting_build(gpu_ctx, oracle, kernel):
    """MIRT_FLAG_COUNT_WORK | MIRT_FLAG_COUNT_GRID counts the grid build that renders many-sphere scenes in
    production: same image, same rays / hits / scatters / sky misses as the reference's flat scan (the path is
    identical), but FEWER sphere tests, plus the cells it walked."""
    w, h, spp = 96, 54, 64
    sd = scene_data("rtiow_final", w, h)
    gpu_ctx.set_scene(sd)
    flat = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, flags=kernel | m.MIRT_FLAG_COUNT_WORK)
    grid = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, flags=kernel | m.MIRT_FLAG_COUNT_WORK | m.MIRT_FLAG_COUNT_GRID)
    img_flat = gpu_ctx.render(flat)
    st_flat = gpu_ctx.stats()
    img_grid = gpu_ctx.render(grid)
    st_grid = gpu_ctx.stats()
    assert_images_equal(img_grid, img_flat, "grid counting build image")
    assert_images_equal(img_flat, oracle.render(sd, flat), "vs oracle")
    for k in ("rays", "hits", "sky_misses", "scatter"):
        assert st_grid[k] == st_flat[k], k
    assert st_flat["grid_cells"] == 0 and st_flat["sphere_tests"] == st_flat["rays"] * len(sd.spheres)
    assert 0 < st_grid["sphere_tests"] < st_flat["sphere_tests"] // 10
    assert st_grid["grid_cells"] > 0 and st_grid["grid_cells"] <= 64 * st_grid["grid_wave_cells"]


@pytest.mark.parametrize("scene,w,h,spp,n", [("three_spheres", 96, 64, 8, 2), ("three_spheres", 61, 47, 12, 4), ("earth", 64, 40, 6, 1),
                                              ("main_rs_scene", 80, 48, 128, 2), ("rtiow_final", 64, 36, 4, 2), ("three_spheres", 3, 1, 70, 1)])
def test_reference_frame_stream_vs_oracle(gpu_ctx, oracle, scene, w, h, spp, n):
    """MirtParams.frame_spp = n: frame f seeds the pixel's RNG once and its n samples draw from that stream in turn
    (initRng / samplePixel, wgsl:498-502, 105-122) -- image, exact sums and work counters against the oracle; forced kernel
    flags must not matter (the lane-per-pixel schedule is the only one that can run it)."""
    sd = scene_data(scene, w, h)
    gpu_ctx.set_scene(sd)
    for flags in (0, m.MIRT_FLAG_KERNEL_POOL, LINEAR | m.MIRT_FLAG_COUNT_WORK):
        p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, flags=flags, frame_spp=n)
        got = gpu_ctx.render(p)
        assert _lane_per_pixel(gpu_ctx.last_kernel()), gpu_ctx.last_kernel()
        gs = gpu_ctx.stats()
        want = oracle.render(sd, p)
        assert_images_equal(got, want, f"{scene} frame_spp {n} flags {flags}")
        if flags & m.MIRT_FLAG_COUNT_WORK:
            os_ = oracle.stats()
            assert {k: gs[k] for k in COUNTERS} == {k: os_[k] for k in COUNTERS}
    # frame by frame through the accumulation API == all frames in one launch == the oracle's sums
    p1 = m.make_params(w, h, n, mode=m.MIRT_MODE_PT, num_bounces=8, frame_spp=n)
    gpu_ctx.accum_reset(p1)
    for _ in range(spp // n):
        gpu_ctx.accum_add(p1)
    whole = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, frame_spp=n)
    assert np.array_equal(gpu_ctx.accum_read(p1), oracle.render_pt_sums(sd, whole))
    if n > 1:
        assert not np.array_equal(gpu_ctx.render(whole), gpu_ctx.render(m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8)))
    # an accumulation that starts at a later frame (MirtParams.frame_begin: the reference's frame_number is never reset)
    later = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, frame_spp=n, frame_begin=11)
    assert_images_equal(gpu_ctx.render(later), oracle.render(sd, later), f"{scene} frame_begin 11")
    gpu_ctx.accum_reset(later)
    l1 = m.make_params(w, h, n, mode=m.MIRT_MODE_PT, num_bounces=8, frame_spp=n, frame_begin=11)
    for _ in range(spp // n):
        gpu_ctx.accum_add(l1)
    assert np.array_equal(gpu_ctx.accum_read(l1), oracle.render_pt_sums(sd, later))


def test_frame_stream_param_errors(gpu_ctx, oracle):
    sd = scene_data("three_spheres", 16, 16)
    gpu_ctx.set_scene(sd)
    for kw in (dict(spp=10, frame_spp=4), dict(spp=8, frame_spp=4, sample_begin=2)):
        p = m.make_params(16, 16, kw.pop("spp"), mode=m.MIRT_MODE_PT, **kw)
        with pytest.raises(m.MirtError) as e:
            gpu_ctx.render(p)
        assert e.value.status_name == "MIRT_ERR_FRAME_SPP"
        assert oracle.render_status(sd.as_c(), p) == m._abi.MIRT_ERR_FRAME_SPP


@pytest.mark.parametrize("flags", [0, m.MIRT_FLAG_KERNEL_STRIP, m.MIRT_FLAG_KERNEL_POOL])
def test_the_top_of_the_sample_range(gpu_ctx, oracle, flags):
    """Sample indices run to 2^32 - 1 (MIRT_ERR_SPP_RANGE beyond): the last ones render like any others, in every kernel."""
    w, h = 40, 12
    sd = scene_data("three_spheres", w, h)
    gpu_ctx.set_scene(sd)
    for spp in (2, 70):
        p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, sample_begin=0xffffffff - spp, flags=flags)
        assert np.array_equal(gpu_ctx.render(p), oracle.render(sd, p))
    with pytest.raises(m.MirtError) as e:
        gpu_ctx.render(m.make_params(w, h, 70, mode=m.MIRT_MODE_PT, sample_begin=0xffffffff - 69, flags=flags))
    assert e.value.status_name == "MIRT_ERR_SPP_RANGE"


def test_accumulation_refuses_a_frame_stream_that_would_start_mid_frame(gpu_ctx):
    """mirt_ctx_accum_add continues at the samples accumulated so far, whatever sample_begin the caller passes: that count must be
    a multiple of frame_spp, or the lane-per-pixel kernel would draw from an unseeded stream until the next frame boundary."""
    sd = scene_data("three_spheres", 24, 16)
    gpu_ctx.set_scene(sd)
    p4 = m.make_params(24, 16, 4, mode=m.MIRT_MODE_PT, frame_spp=2)
    gpu_ctx.accum_reset(p4)
    gpu_ctx.accum_add(p4)
    assert gpu_ctx.accum_samples() == 4
    p3 = m.make_params(24, 16, 3, mode=m.MIRT_MODE_PT, frame_spp=3)          # valid on its own (3 | 3, 3 | 0) ...
    with pytest.raises(m.MirtError) as e:
        gpu_ctx.accum_add(p3)                                               # ... but 4 samples are in the buffer
    assert e.value.status_name == "MIRT_ERR_FRAME_SPP" and gpu_ctx.accum_samples() == 4
    gpu_ctx.accum_add(p4)                                                   # the buffer is still usable
    assert gpu_ctx.accum_samples() == 8
    gpu_ctx.accum_reset(p3)
    gpu_ctx.accum_add(p3)
    assert gpu_ctx.accum_samples() == 3


def test_raytracer_render_frame_with_the_reference_stream(oracle):
    scene, cam = m.scenes.three_spheres()
    rp = m.RenderParams(camera=cam, viewport_size=(64, 40), sampling=m.SamplingParams(8, 2, 8))
    rt = m.Raytracer(scene, rp, reference_stream=True)
    sd = rt.scene_data()
    for k in (2, 4, 6, 8, 8):
        img = rt.render_frame()
        assert_images_equal(img, oracle.render(sd, m.make_params(64, 40, k, mode=m.MIRT_MODE_PT, num_bounces=8, frame_spp=2)), f"after {k} spp")
    assert rt.frame_number == 6                      # five render_frame calls, the idle one (accumulation complete) included: mod.rs:350
    # a camera move resets the accumulation (mod.rs:385) but NOT the frame number: the new accumulation's frames are 6, 7, ...
    cam2 = m.Camera(cam.eye_pos + np.float32([0.5, 0.0, 0.0]), cam.eye_dir, cam.up, cam.vfov, cam.aperture, cam.focus_distance)
    rt.set_render_params(m.RenderParams(camera=cam2, viewport_size=(64, 40), sampling=m.SamplingParams(8, 2, 8)))
    sd2 = rt.scene_data()
    for k in (2, 4):
        img = rt.render_frame()
        want = oracle.render(sd2, m.make_params(64, 40, k, mode=m.MIRT_MODE_PT, num_bounces=8, frame_spp=2, frame_begin=5))
        assert_images_equal(img, want, f"after the camera move, {k} spp")
    assert rt.frame_number == 8
    # the one-shot render() does NOT inherit the progressive loop's frame count: frames 1..4 by default (a fresh reference Raytracer),
    # whatever render_frame / set_render_params calls came before; frame_begin = frame_number - 1 asks for the accumulation the loop
    # would start now
    fresh = oracle.render(sd2, m.make_params(64, 40, 8, mode=m.MIRT_MODE_PT, num_bounces=8, frame_spp=2, frame_begin=0))
    assert_images_equal(rt.render(), fresh, "render() after a session = frames 1..4")
    now = oracle.render(sd2, m.make_params(64, 40, 8, mode=m.MIRT_MODE_PT, num_bounces=8, frame_spp=2, frame_begin=7))
    assert_images_equal(rt.render(frame_begin=rt.frame_number - 1), now, "render(frame_begin=frame_number - 1)")
    rt.close()


@pytest.mark.parametrize("seed", range(_FUZZ0, _FUZZ0 + int(os.environ.get("MIRT_GRID_FUZZ_SEEDS", "10"))))      # deeper runs: MIRT_GRID_FUZZ_SEEDS=60
def test_grid_builds_on_random_soups(gpu_ctx, oracle, seed):
    """Random soups of 32-700 spheres (0-3 big ones besides the ground, random cell sizes through the radii, random
    cameras with and without aperture, 2-8 bounces): the pool kernel's grid build -- several pool geometries occur
    (160 / 152 / 128 / 96 slots, whichever fits LDS beside the blob), walks in instalments, candidate lists, one scatter queue -- the strip
    kernel's grid build and the default choice must all give the oracle's flat-scan image exactly."""
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(32, 700))
    spread = float(rng.uniform(0.5, 9.0))
    spheres, gm, tex = _sphere_soup(rng, n, spread, 0.03, float(rng.uniform(0.1, 0.6)), n_big=int(rng.integers(0, 4)))
    w, h, spp = 96, 54, int(rng.choice([48, 64, 100]))
    cam = simple_camera(w, h, eye=tuple(rng.normal(size=3) * 4 + np.array([0, 3, 8])), direction=(float(rng.normal() * 0.3), -0.3, -1.0),
                        vfov=float(rng.uniform(25, 70)), aperture=float(rng.choice([0.0, 0.3])), focus=8.0)
    sd = m.SceneData(cam, spheres, gm, tex)
    gpu_ctx.set_scene(sd)
    p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=int(rng.integers(2, 9)), flags=LINEAR)
    want = oracle.render(sd, p)
    for fl in (0, m.MIRT_FLAG_KERNEL_POOL, m.MIRT_FLAG_KERNEL_STRIP):
        p.flags = LINEAR | fl
        assert_images_equal(gpu_ctx.render(p), want, f"seed {seed}: {n} spheres, flags {fl}, {gpu_ctx.last_kernel()}")
        if fl == m.MIRT_FLAG_KERNEL_POOL:
            assert gpu_ctx.last_kernel().startswith("render_pt_pool_kernel<1024,") and ",1,true," in gpu_ctx.last_kernel()


def _ground_plane_soup(rng, n, spread, r_lo, r_hi, n_big=2, lattice=False):
    """Spheres resting on the ground sphere (RTIOW's layout): with radii this close the uniform grid is ONE cell high."""
    T = m.Texture
    mats = [m.Material.Lambertian(T.new_from_color(rng.random(3))), m.Material.Metal(T.new_from_color(0.5 + 0.5 * rng.random(3)), 0.3),
            m.Material.Dielectric(1.5), m.Material.Checkerboard(even=T.new_from_color((0.2, 0.3, 0.1)), odd=T.new_from_color((0.9, 0.9, 0.9)))]
    gm, tex = m.flatten_materials(mats)
    spheres = [m.Sphere.new((0.0, -1000.0, 0.0), 1000.0, 3).to_c()]
    for _ in range(n_big):
        spheres.append(m.Sphere.new((float(rng.normal() * spread * 0.3), 1.0, float(rng.normal() * spread * 0.3)), 1.0, int(rng.integers(0, 3))).to_c())
    k = int(np.ceil(np.sqrt(n)))
    for i in range(n):
        r = float(rng.uniform(r_lo, r_hi))
        if lattice:             # centres ON a lattice of 0.5 (binary fractions): rays along the axes meet cell faces and sphere poles exactly
            x, z = 0.5 * (i % k - k // 2), 0.5 * (i // k - k // 2)
            r = 0.125
        else:
            x, z = float(rng.normal() * spread), float(rng.normal() * spread)
        spheres.append(m.Sphere.new((x, r, z), r, int(rng.integers(0, 4))).to_c())
    return spheres, gm, tex


@pytest.mark.parametrize("seed", range(_FUZZ0, _FUZZ0 + int(os.environ.get("MIRT_FLAT_GRID_FUZZ_SEEDS", "8"))))
def test_flat_grids_walk_in_two_dimensions(oracle, seed, monkeypatch):
    """A grid ONE cell high (spheres on a ground plane: RTIOW, BASELINE configs[4]) is walked in two dimensions by the pooled kernel
    (csrc grid_walk<COUNT, FLATY>; the y slab ends the walk through the clip's tmax).  Random ground-plane soups under random cameras
    -- above, inside the slab, grazing, with and without a lens; every other seed on a lattice that puts cell faces, sphere poles and
    ray origins on binary fractions -- must give the oracle's flat-scan image exactly, and byte for byte the frame of the
    three-dimensional walk (MIRT_GRID_FLAT_Y=0)."""
    rng = np.random.default_rng(15000 + seed)
    lattice = seed % 2 == 1
    n = int(rng.integers(40, 600))
    spheres, gm, tex = _ground_plane_soup(rng, n, float(rng.uniform(1.0, 6.0)), 0.16, 0.2, n_big=int(rng.integers(0, 4)), lattice=lattice)
    w, h, spp = 96, 54, int(rng.choice([32, 48, 100]))
    kind = seed % 4
    if kind == 0:
        cam = simple_camera(w, h, eye=(float(rng.normal() * 3), float(rng.uniform(0.5, 4.0)), float(rng.uniform(4, 10))),
                            direction=(float(rng.normal() * 0.3), -0.3, -1.0), vfov=float(rng.uniform(25, 70)), aperture=float(rng.choice([0.0, 0.2])), focus=8.0)
    elif kind == 1:               # inside the slab, looking along -z on a cell face (x = 0 exactly; the lattice's centres are multiples of 0.5)
        cam = simple_camera(w, h, eye=(0.0, 0.125, 6.0), direction=(0.0, 0.0, -1.0), vfov=40.0, aperture=0.0, focus=6.0)
    elif kind == 2:               # grazing: just above the slab, almost horizontal
        cam = simple_camera(w, h, eye=(float(rng.normal()), 0.45, 9.0), direction=(0.05, -0.02, -1.0), vfov=30.0, aperture=0.0, focus=9.0)
    else:                         # from straight above, and a wide lens
        cam = simple_camera(w, h, eye=(0.25, 12.0, 0.25), direction=(0.0, -1.0, -1e-3), vfov=50.0, aperture=0.5, focus=12.0)
    sd = m.SceneData(cam, spheres, gm, tex)
    p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=int(rng.integers(2, 9)), flags=LINEAR | m.MIRT_FLAG_KERNEL_POOL)
    want = oracle.render(sd, p)
    flat = m.Context(0)
    monkeypatch.setenv("MIRT_GRID_FLAT_Y", "0")
    cubic = m.Context(0)                                       # tuning knobs are read once, in mirt_ctx_create
    monkeypatch.delenv("MIRT_GRID_FLAT_Y")
    try:
        flat.set_scene(sd)
        cubic.set_scene(sd)
        got = flat.render(p)
        assert flat.last_kernel().startswith("render_pt_pool_kernel<1024,") and flat.last_kernel().endswith(",1,true,true>"), flat.last_kernel()
        assert_images_equal(got, want, f"seed {seed}: {n} spheres on the ground, camera kind {kind}, lattice {lattice}, {flat.last_kernel()}")
        got3 = cubic.render(p)
        assert cubic.last_kernel().endswith(",1,true,false>"), cubic.last_kernel()
        assert np.array_equal(got3, got)
        # the strip kernel's grid build (lane = pixel / lane = sample) takes the same two-dimensional walk behind a wave-uniform flag
        for spp_s in (3, 16):
            ps = m.make_params(w, h, spp_s, mode=m.MIRT_MODE_PT, num_bounces=p.num_bounces, flags=LINEAR | m.MIRT_FLAG_KERNEL_STRIP)
            got_s = flat.render(ps)
            assert flat.last_kernel().startswith("render_pt_strip_kernel<false,false,true,"), flat.last_kernel()
            assert_images_equal(got_s, oracle.render(sd, ps), f"seed {seed}: strip kernel, {spp_s} spp, {flat.last_kernel()}")
            assert np.array_equal(cubic.render(ps), got_s)
        pc = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=p.num_bounces, flags=LINEAR | m.MIRT_FLAG_KERNEL_POOL | m.MIRT_FLAG_COUNT_WORK | m.MIRT_FLAG_COUNT_GRID)
        assert_images_equal(flat.render(pc), want, "counting build of the two-dimensional walk")
        if flat.last_kernel().endswith(",true,true>"):      # (counting builds of the grid kernel exist for its two largest pool geometries; others count with the flat scan)
            assert flat.last_kernel().endswith(",1,true,true>") and ",true,false,1," in flat.last_kernel(), flat.last_kernel()
            st2, _ = flat.stats(), cubic.render(pc)
            st3 = cubic.stats()
            for k in ("rays", "hits", "sky_misses", "scatter", "sphere_tests"):      # same cells, same tests (an exact tie of a z crossing with the slab's end apart)
                if k == "sphere_tests":
                    assert abs(st2[k] - st3[k]) <= 1e-4 * st3[k], (k, st2[k], st3[k])
                else:
                    assert st2[k] == st3[k], (k, st2[k], st3[k])
    finally:
        flat.close()
        cubic.close()


def test_default_kernel_choice_follows_the_measured_crossovers(gpu_ctx):
    """mirt_kernels.h kPoolMinSpp*: several shading routines -> pool from 32 spp; one routine -> lane-per-pixel strip kernel below
    800 spp (on frames large enough to feed every CU), pool from there; many-sphere scenes -> grid pool from 16 spp.  Every
    choice renders the same image as the forced alternatives (checked throughout this file); here: the names."""
    def kernel(scene, w, h, spp):
        gpu_ctx.set_scene(scene_data(scene, w, h))
        gpu_ctx.render(m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8))
        return gpu_ctx.last_kernel()
    assert kernel("three_spheres", 640, 360, 12) == "render_pt_strip_kernel<false,false,false,true>"
    assert kernel("three_spheres", 640, 360, 28) == "render_pt_stream_kernel<false>"
    assert kernel("three_spheres", 640, 360, 32).startswith("render_pt_pool_kernel<256,112,6,")
    assert kernel("single_sphere", 640, 360, 100) == "render_pt_stream_kernel<false>"
    assert kernel("single_sphere", 640, 360, 792) == "render_pt_stream_kernel<false>"
    assert kernel("single_sphere", 640, 360, 800).startswith("render_pt_pool_kernel<256,112,6,")
    assert kernel("single_sphere", 64, 36, 100) == "render_pt_strip_kernel<false,false,false,false>"      # tiny frame: lanes on samples
    assert kernel("rtiow_final", 640, 360, 12) == "render_pt_strip_kernel<false,false,true,true>"
    assert kernel("rtiow_final", 640, 360, 16).startswith("render_pt_pool_kernel<1024,")
