#!/usr/bin/env python3
"""Developer tool: kernel time of ONE rank's share of the bench frame for 1/2/4/8-way partitions
(what each GPU of an N-GPU run executes), to see how strong scaling behaves before the gather."""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import weekend_raytracer_wgpu_amd as m
from helpers import scene_data
w, h, spp = 1920, 1080, 1000
ctx = m.Context(0)
ctx.set_scene(scene_data("three_spheres", w, h))
base = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8)
t1 = None
worlds = tuple(int(x) for x in sys.argv[1].split(',')) if len(sys.argv) > 1 else (1, 2, 4, 8)
for world in worlds:
    ts = []
    for part in range(min(world, 3)):
        p = m.multi_gpu.part_params(base, part, world, 4)
        ctx.render(p); ctx.render(p)
        ts.append(ctx.stats()["kernel_ms"])
    t = max(ts)
    t1 = t1 or t
    print(f"world {world}: slowest of {len(ts)} parts {t:7.3f} ms  parts {['%.3f' % x for x in ts]}  ideal {t1 / world:7.3f}  efficiency {t1 / world / t:.3f}")
