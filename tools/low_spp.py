#!/usr/bin/env python3
"""Developer tool: throughput of low-spp launches (the reference's interactive path adds 2 spp per frame,
mod.rs:606-611) on config 3 at 1920x1080: plain render and progressive accumulation."""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import weekend_raytracer_wgpu_amd as m
from helpers import scene_data
w, h = 1920, 1080
ctx = m.Context(0); ctx.set_scene(scene_data("three_spheres", w, h))
buf = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for spp in (1, 2, 4, 8, 16, 32, 64, 128):
    p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8)
    ctx.stats()
    for _ in range(5):
        ctx.render_device(p, buf.data_ptr(), buf.numel(), stream)
    torch.cuda.synchronize()
    st = ctx.stats(); t = st["kernel_ms_total"] / st["launches"]
    ctx.accum_reset(p)
    ctx.stats()
    for i in range(5):
        p.sample_begin = i * spp
        ctx.accum_add(p, stream)
    torch.cuda.synchronize()
    st = ctx.stats(); ta = st["kernel_ms_total"] / st["launches"]
    print(f"spp {spp:4d}: render {t:8.3f} ms {w * h * spp / t / 1e3:9.1f} Msamples/s | accum_add {ta:8.3f} ms {w * h * spp / ta / 1e3:9.1f} Msamples/s")
