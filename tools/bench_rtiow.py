#!/usr/bin/env python3
"""Developer tool: config 5 (RTIOW final scene, ~484 spheres) — uniform grid vs flat scan."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import weekend_raytracer_wgpu_amd as m
from helpers import scene_data

w, h, spp = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (960, 540, 64)
ctx = m.Context(0)
ctx.set_scene(scene_data("rtiow_final", w, h))
ref = None
for name, flags in (("default", 0), ("strip + grid", m.MIRT_FLAG_KERNEL_STRIP), ("pool + grid", m.MIRT_FLAG_KERNEL_POOL),
                    ("flat strip", m.MIRT_FLAG_NO_GRID | m.MIRT_FLAG_KERNEL_STRIP), ("flat pool", m.MIRT_FLAG_NO_GRID | m.MIRT_FLAG_KERNEL_POOL)):
    p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, flags=flags)
    img = ctx.render(p)
    img = ctx.render(p)
    img = ctx.render(p)
    ms = ctx.stats()["kernel_ms"]
    if ref is None:
        ref = img
    print(f"{name:16s} {ms:9.2f} ms  {w * h * spp / ms / 1e3:9.1f} Msamples/s  identical={np.array_equal(img, ref)}")
