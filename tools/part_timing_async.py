#!/usr/bin/env python3
"""Developer tool: like part_timing.py, but the launches of one part are queued back to back on the stream
(as consecutive bench steps are) and timed per launch with the library's HIP events, so that clock ramp-up
after an idle gap does not enter the figure."""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import weekend_raytracer_wgpu_amd as m
from helpers import scene_data
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="three_spheres")
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--spp", type=int, default=1000)
ap.add_argument("--all-parts", action="store_true", help="time every part of each partition (the slowest one is a rank's time), not part 0 only")
a = ap.parse_args()
w, h = map(int, a.size.split("x"))
spp = a.spp
ctx = m.Context(0)
ctx.set_scene(scene_data(a.scene, w, h))
base = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8)
buf = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
t1 = None
for world in (1, 2, 4, 8):
    times = []
    for part in (range(world) if a.all_parts else (0,)):
        p = m.multi_gpu.part_params(base, part, world, 4)
        n = 6 if spp * w * h < 3e9 else 3
        for _ in range(2 if n == 6 else 1):              # untimed: the clock ramps up after an idle gap (4 % on a 23 ms launch)
            ctx.render_device(p, buf.data_ptr(), buf.numel(), stream)
        torch.cuda.synchronize()
        ctx.stats()
        for _ in range(n):
            ctx.render_device(p, buf.data_ptr(), buf.numel(), stream)
        torch.cuda.synchronize()
        st = ctx.stats()
        times.append(st["kernel_ms_total"] / st["launches"])
    t = max(times)
    t1 = t1 or t
    print(f"world {world}: {'slowest of all parts' if a.all_parts else 'part 0'}, launches back to back: {t:8.3f} ms per launch  ideal {t1 / world:8.3f}  "
          f"efficiency {t1 / world / t:.3f}  (parts: {' '.join(f'{x:.3f}' for x in times)})", flush=True)
