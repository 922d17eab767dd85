"""The C oracle (oracle/mirt_oracle_parity.c) against a second, independent restatement of the reference's CPU loop
written in numpy float32 straight from the reference's source text (oracle/mirt_oracle_parity_np.py): pixel-exact on
`Layer::scene` at the reference's own 800x600 and on random sphere soups / cameras.  This does not pin parity (the
reference holds no fixtures for this path) but it removes the single-author risk of one restatement."""
import sys
from pathlib import Path

import numpy as np
import pytest

import weekend_raytracer_wgpu_amd as m
from helpers import assert_images_equal, layer_scene_data, metal_table

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
import mirt_oracle_parity_np as npo  # noqa: E402


def _np_render(sd, w, h, spp, rows=None):
    cam = {k: np.array(getattr(sd.camera, k)[:3], dtype=np.float32) for k in ("eye", "horizontal", "vertical", "lower_left_corner")}
    spheres = [(s.center[0], s.center[1], s.center[2], s.radius) for s in sd.spheres]
    mat2 = None
    if len(sd.materials) > 2:
        d = sd.materials[2].desc1
        mat2 = (d.width, d.height, d.offset, sd.materials[2].x)
    return npo.render_parity(cam, spheres, mat2, sd.texels, w, h, spp, rows)


@pytest.mark.parametrize("spp", [2, 20, 21, 100])
def test_layer_scene_800x600(oracle, spp):
    w, h = 800, 600
    sd = layer_scene_data(w, h)
    want = oracle.render(sd, m.make_params(w, h, spp))
    assert_images_equal(_np_render(sd, w, h, spp), want, f"Layer::scene 800x600 spp {spp}")
    # the image has all three pixel classes (coloured, gradient, and black once spp >= 21)
    assert (want[..., :3].reshape(-1, 3) != 0).any()


def test_layer_scene_1080p_rows(oracle):
    w, h = 1920, 1080
    sd = layer_scene_data(w, h)
    rows = [0, 1, 333, 540, 811, 1079]
    got = _np_render(sd, w, h, 21, rows)
    for k, r in enumerate(rows):
        want = oracle.render(sd, m.make_params(w, h, 21, row_begin=r, row_end=r + 1))
        assert_images_equal(got[k:k + 1], want, f"row {r}")


@pytest.mark.parametrize("seed", range(12))
def test_random_soups_and_cameras(oracle, seed):
    """The generator of tests/test_gpu_parity.py::test_random_scenes_and_cameras: overlapping, nested, tiny and huge
    spheres, cameras anywhere -- every branch of closest_hit_raw / scatter_metal, last-hit-wins included."""
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
        assert_images_equal(_np_render(sd, w, h, spp), oracle.render(sd, m.make_params(w, h, spp)), f"seed {seed} spp {spp}")


def test_empty_world_is_the_gradient(oracle):
    w, h = 64, 48
    from helpers import simple_camera
    sd = m.SceneData(simple_camera(w, h), [], [], np.zeros((0, 3), np.float32))
    assert_images_equal(_np_render(sd, w, h, 5), oracle.render(sd, m.make_params(w, h, 5)), "K1")
