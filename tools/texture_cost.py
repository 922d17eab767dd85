#!/usr/bin/env python3
"""Developer tool: what do texel fetches cost?  Config 4 (earth) with its 1024x512 texture vs the same
scene with the texture replaced by a 1x1 colour (no global-memory access in the lookup)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import weekend_raytracer_wgpu_amd as m

w, h, spp = 1920, 1080, 500
ctx = m.Context(0)
sc, cam = m.scenes.earth()
gcam = m.GpuCamera.new(cam, (w, h)).c
for label, mats in (("1024x512 earth texture", sc.materials),
                    ("1x1 colour instead", sc.materials[:4] + [m.Material.Lambertian(m.Texture.new_from_color((0.3, 0.4, 0.5)))])):
    gm, tex = m.flatten_materials(mats)
    ctx.set_scene(m.SceneData(gcam, [s.to_c() for s in sc.spheres], gm, tex))
    for kname, kf in (("pool", m.MIRT_FLAG_KERNEL_POOL), ("strip", m.MIRT_FLAG_KERNEL_STRIP)):
        p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, flags=kf)
        ts = []
        for _ in range(4):
            ctx.render(p)
            ts.append(ctx.stats()["kernel_ms"])
        print(f"{label:26s} {kname:5s} median {np.median(ts[1:]):7.2f} ms  {w * h * spp / np.median(ts[1:]) / 1e3:9.1f} Msamples/s")
