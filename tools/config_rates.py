#!/usr/bin/env python3
"""Developer tool: throughput of every BASELINE config's scene at 1920x1080 x 1000 spp (8 bounces) with the
library's default kernel selection, plus the RTIOW scene at 960x540 x 64 spp.  Prints one line per scene."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import weekend_raytracer_wgpu_amd as m
from helpers import scene_data

ctx = m.Context(0)
for name, w, h, spp in (("single_sphere", 1920, 1080, 1000), ("three_spheres", 1920, 1080, 1000), ("earth", 1920, 1080, 1000),
                        ("main_rs_scene", 1920, 1080, 1000), ("rtiow_final", 960, 540, 64)):
    ctx.set_scene(scene_data(name, w, h))
    p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8)
    ts = []
    for _ in range(4):
        ctx.render(p)
        ts.append(ctx.stats()["kernel_ms"])
    ms = float(np.median(ts[1:]))
    print(f"{name:14s} {w}x{h}x{spp:<5d} {ms:9.3f} ms  {w * h * spp / ms / 1e3:10.1f} Msamples/s")
