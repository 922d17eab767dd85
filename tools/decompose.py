#!/usr/bin/env python3
"""Developer tool: per-op cost decomposition of the pool kernel from synthetic scenes."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import weekend_raytracer_wgpu_amd as m
from helpers import scene_data, simple_camera

w, h, spp = 1920, 1080, 500
ctx = m.Context(0)
T = m.Texture
def mats(*ms): return m.flatten_materials(list(ms))
cam = m.GpuCamera.new(m.FlyCameraController.default().renderer_camera(), (w, h)).c
far = lambda i, mi: m.Sphere.new((1e4 + 10 * i, 1e4, 1e4), 1.0, mi).to_c()
cases = {}
gm, tx = mats(m.Material.Metal(T.new_from_color((1, 1, 1)), 0.1), m.Material.Dielectric(1.5))
cases["empty (GEN only, N=0)"] = m.SceneData(cam, [], gm, tx)
cases["3 unreachable spheres (GEN + 3 tests)"] = m.SceneData(cam, [far(0, 0), far(1, 1), far(2, 0)], gm, tx)
cases["12 unreachable spheres"] = m.SceneData(cam, [far(i, i % 2) for i in range(12)], gm, tx)
big = lambda mat: [m.Sphere.new((0, -1000.0, 0), 1000.0, 0).to_c(), far(1, 1)]
for name, mat in [("lambertian ground", m.Material.Lambertian(T.new_from_color((0.5, 0.5, 0.5)))),
                  ("checker ground", m.Material.Checkerboard(even=T.new_from_color((0.5, 0.7, 0.8)), odd=T.new_from_color((0.9, 0.9, 0.9)))),
                  ("metal ground", m.Material.Metal(T.new_from_color((0.9, 0.9, 0.9)), 0.3)),
                  ("glass ground", m.Material.Dielectric(1.5))]:
    g2, t2 = mats(mat, m.Material.Dielectric(1.5))
    cases[name + " + 1 far sphere"] = m.SceneData(cam, big(mat), g2, t2)
cases["config 3"] = scene_data("three_spheres", w, h)
for name, sd in cases.items():
    ctx.set_scene(sd)
    for kname, kf in (("pool", m.MIRT_FLAG_KERNEL_POOL), ("strip", m.MIRT_FLAG_KERNEL_STRIP)):
        p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, flags=kf)
        ctx.render(p); ctx.render(p)
        ms = ctx.stats()["kernel_ms"]
        p.flags |= m.MIRT_FLAG_COUNT_WORK
        ctx.render(p)
        st = ctx.stats()
        ops = st["rays"]
        wave_ops = st["wave_iterations"]
        # SIMD cycles per wave-op: time x clock x SIMDs / wave-ops
        cyc = ms * 1e-3 * 2.37e9 * 1024 / max(1, wave_ops)
        print(f"{name:42s} {kname:5s} {ms:7.2f} ms  rays/sample {ops / st['samples']:.2f}  scatter {st['scatter']}  "
              f"lane-util {st['lane_iterations'] / max(1, 64 * wave_ops):.3f}  SIMD-cycles/wave-op {cyc:7.0f}")
