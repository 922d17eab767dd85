"""Host-side mirror of the reference interface: table layouts, camera construction, validation
and row-partition arithmetic of the product agree with the oracle's independent restatement."""
import ctypes as C
import itertools

import numpy as np
import pytest

import weekend_raytracer_wgpu_amd as m
from weekend_raytracer_wgpu_amd import _abi
from helpers import layer_scene_data


def test_set_global_data_layout():
    """layer.rs:125-148 + mod.rs:767-830: ids, descriptor offsets and the (odd, even) argument swap."""
    sd = layer_scene_data(80, 60)
    mats = sd.materials
    assert [x.id for x in mats] == [3, 0, 1, 2, 0]
    assert (mats[0].desc1.offset, mats[0].desc2.offset) == (0, 1)
    assert (mats[1].desc1.width, mats[1].desc1.height, mats[1].desc1.offset) == (1024, 512, 2)
    assert (mats[2].desc1.offset, mats[4].desc1.offset) == (524290, 524291)
    assert sd.texels.shape == (1048579, 3)
    # call sites pass (odd, even) into (even, odd): desc1 is the ODD colour (0.9,0.9,0.9)
    assert np.allclose(sd.texels[0], 0.9) and np.allclose(sd.texels[1], (0.5, 0.7, 0.8))
    assert mats[3].desc1.offset == 0xFFFFFFFF and mats[3].x == np.float32(1.5)
    assert mats[2].x == np.float32(0.4)
    # texel = inv_255 * (p as f32) (texture.rs:30-41); earthmap texel (0,0) is white
    assert np.array_equal(sd.texels[524291], np.float32(1.0 / 255.0) * np.float32([255, 255, 255]))


def test_layer_scene_sphere_order():
    sc = m.Layer.scene()       # layer.rs:113-120 — order matters (last hit wins)
    got = [(tuple(float(x) for x in s.center), s.radius, s.material_idx) for s in sc.spheres]
    f = np.float32
    assert got == [((5.0, float(f(1.2)), -1.5), float(f(1.2)), 4), ((0.0, -500.0, -1.0), 500.0, 0),
                   ((0.0, 1.0, 0.0), 1.0, 3), ((-5.0, 1.0, 0.0), 1.0, 2), ((2.0, -1.0, 0.0), 2.0, 3),
                   ((5.0, float(f(0.8)), 1.5), float(f(0.8)), 1)]


def test_camera_new_bitwise_equal_to_oracle(oracle):
    rng = np.random.default_rng(3)
    cams = [m.FlyCameraController.default().renderer_camera()]
    for _ in range(50):
        fc = m.FlyCameraController(rng.normal(size=3).astype(np.float32) * 5, m.Angle.degrees(float(rng.uniform(-180, 180))),
                                   m.Angle.degrees(float(rng.uniform(-80, 80))), float(rng.uniform(5, 90)),
                                   float(rng.uniform(0, 1)), float(rng.uniform(0.5, 20)))
        cams.append(fc.renderer_camera())
    for cam, (w, h) in zip(cams, itertools.cycle([(800, 600), (1920, 1080), (333, 77), (1, 1)])):
        c = cam.to_c()
        a, b = _abi.MirtGpuCamera(), _abi.MirtGpuCamera()
        assert m.lib().mirt_camera_new(C.byref(c), w, h, C.byref(a)) == 0
        assert oracle.LIB.mirt_oracle_camera_new(C.byref(c), w, h, C.byref(b)) == 0
        assert bytes(a) == bytes(b)


def test_fly_pose_bitwise_equal_to_oracle(oracle):
    pos = (C.c_float * 3)(-10.0, 2.0, -4.0)
    a, b = _abi.MirtCamera(), _abi.MirtCamera()
    yaw, pitch = m.Angle.degrees(25.0).as_radians(), m.Angle.degrees(-10.0).as_radians()
    assert m.lib().mirt_camera_from_fly_pose(pos, yaw, pitch, 30.0, 0.8, 10.816654, C.byref(a)) == 0
    assert oracle.LIB.mirt_oracle_camera_from_fly_pose(pos, yaw, pitch, 30.0, 0.8, 10.816654, C.byref(b)) == 0
    assert bytes(a) == bytes(b)
    # default pose (fly_camera.rs:24-50): focus distance = |(10,-1,4)| = sqrt(117)
    fc = m.FlyCameraController.default()
    assert fc.focus_distance == float(np.sqrt(np.float32(117.0)))
    cam = fc.renderer_camera()
    assert abs(float(np.dot(cam.eye_dir, cam.up))) < 1e-6 and abs(np.linalg.norm(cam.eye_dir) - 1) < 1e-6


def _rp(**kw):
    cam = m.FlyCameraController.default().renderer_camera()
    for k in ("vfov", "aperture", "focus_distance"):
        if k in kw:
            setattr(cam, k, kw.pop(k))
    return m.RenderParams(camera=cam, **kw)


@pytest.mark.parametrize("kwargs,kind", [
    (dict(sampling=m.SamplingParams(128, 3, 8)), "MaxSampleCountNotMultiple"),
    (dict(viewport_size=(0, 600)), "ViewportSize"),
    (dict(viewport_size=(800, 0)), "ViewportSize"),
    (dict(vfov=m.Angle.degrees(91.0)), "VfovOutOfRange"),
    (dict(vfov=m.Angle.degrees(-1.0)), "VfovOutOfRange"),
    (dict(aperture=1.5), "ApertureOutOfRange"),
    (dict(aperture=-0.1), "ApertureOutOfRange"),
    (dict(focus_distance=-1.0), "FocusDistanceOutOfRange"),
])
def test_render_params_validate_errors(kwargs, kind, oracle):
    """mod.rs:450-484: same variants, same order of checks, product == oracle."""
    rp = _rp(**kwargs)
    with pytest.raises(m.RenderParamsValidationError) as e:
        rp.validate()
    assert e.value.kind == kind
    cam, smp = rp.camera.to_c(), rp.sampling.to_c()
    assert oracle.LIB.mirt_oracle_validate_render_params(C.byref(cam), C.byref(smp), *rp.viewport_size) == e.value.status


def test_render_params_validate_ok_and_order():
    _rp().validate()
    _rp(vfov=m.Angle.degrees(90.0), aperture=1.0, focus_distance=0.0).validate()     # inclusive bounds, focus >= 0
    # the sample-count check comes first (mod.rs:451), then the viewport
    with pytest.raises(m.RenderParamsValidationError) as e:
        _rp(sampling=m.SamplingParams(128, 3, 8), viewport_size=(0, 0)).validate()
    assert e.value.kind == "MaxSampleCountNotMultiple"


def test_row_partition_functions_agree_with_oracle(oracle):
    rng = np.random.default_rng(4)
    for _ in range(300):
        h = int(rng.integers(1, 200))
        rb = int(rng.integers(0, h))
        re = int(rng.integers(rb, h + 2))
        tr = int(rng.integers(0, 20))
        n = int(rng.integers(0, 9))
        part = int(rng.integers(0, 9))
        p = m.make_params(17, h, 1, row_begin=rb, row_end=re, tile_rows=tr, n_parts=n, part=part)
        rows = m.params_out_rows(p)
        assert rows == oracle.out_rows(p)
        for i in list(range(min(rows, 40))) + [rows, rows + 5]:
            assert m.params_out_row_index(p, i) == oracle.out_row_index(p, i)


def test_row_partition_covers_every_row_exactly_once():
    for h, tr, world in [(1080, 4, 8), (1080, 8, 8), (2160, 16, 8), (37, 4, 3), (5, 8, 4), (600, 1, 7)]:
        base = m.make_params(64, h, 1)
        seen = []
        for r in range(world):
            p = m.multi_gpu.part_params(base, r, world, tr)
            seen += [m.params_out_row_index(p, i) for i in range(m.params_out_rows(p))]
        assert sorted(seen) == list(range(h))


def test_texture_from_fixture_and_color():
    t = m.Texture.new_from_image(m.asset_path("assets/earthmap.jpeg"))
    assert t.dimensions() == (1024, 512) and t.as_slice().shape == (524288, 3)
    c = m.Texture.new_from_color((1.0, 0.85, 0.57))
    assert c.dimensions() == (1, 1) and np.array_equal(c.as_slice()[0], np.float32([1.0, 0.85, 0.57]))
    with pytest.raises(FileNotFoundError):
        m.Texture.new_from_image("/nonexistent/texture.jpeg")


def test_rgba_to_rgb_view():
    rgba = np.arange(4 * 6, dtype=np.uint8).reshape(6, 4)
    rgb = np.zeros((6, 3), np.uint8)
    assert m.lib().mirt_rgba8_to_rgb8(rgba.ctypes.data_as(C.c_void_p), 6, rgb.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(rgb, rgba[:, :3])


def test_rtiow_generator_is_deterministic():
    a, _ = m.scenes.rtiow_final()
    b, _ = m.scenes.rtiow_final()
    assert 470 <= len(a.spheres) <= 500 and len(a.spheres) == len(b.spheres) == len(a.materials)
    assert all(np.array_equal(x.center, y.center) for x, y in zip(a.spheres, b.spheres))


def test_stream_argument_of_the_device_wrappers():
    """None = the context's own stream (NULL in the C ABI); torch's default stream has handle 0, which must NOT
    collapse into NULL -- it is passed as hipStreamLegacy (1); any other handle goes through unchanged."""
    from weekend_raytracer_wgpu_amd.context import HIP_STREAM_LEGACY, _stream_arg
    assert _stream_arg(None) is None
    assert _stream_arg(0).value == HIP_STREAM_LEGACY == 1
    assert _stream_arg(0x7f00dead0000).value == 0x7f00dead0000
