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
w, h, spp = 1920, 1080, 1000
ctx = m.Context(0)
ctx.set_scene(scene_data("three_spheres", w, h))
base = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8)
buf = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
t1 = None
for world in (1, 2, 4, 8):
    p = m.multi_gpu.part_params(base, 0, world, 4)
    ctx.stats()
    n = 6
    for _ in range(n):
        ctx.render_device(p, buf.data_ptr(), buf.numel(), stream)
    torch.cuda.synchronize()
    st = ctx.stats()
    t = st["kernel_ms_total"] / st["launches"]
    t1 = t1 or t
    print(f"world {world}: part 0, {st['launches']} launches back to back: {t:7.3f} ms per launch  ideal {t1 / world:7.3f}  efficiency {t1 / world / t:.3f}")
