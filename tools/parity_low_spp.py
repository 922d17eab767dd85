#!/usr/bin/env python3
"""`Layer::set_data` at the reference's own operating point -- 800x600 (main.rs:28) and 1080p, 2 samples per pixel
(mod.rs:605-613) -- on the GPU: the lane = pixel schedule of render_parity_kernel (default below 64 spp) against the
lane = sample schedule (MIRT_FLAG_KERNEL_STRIP), same image.

    python tools/parity_low_spp.py [--launches 50]
"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import weekend_raytracer_wgpu_amd as m  # noqa: E402
from helpers import layer_scene_data  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=50)
a = ap.parse_args()
ctx = m.Context(0)
rows = []
for (w, h) in ((800, 600), (1920, 1080)):
    sd = layer_scene_data(w, h)
    ctx.set_scene(sd)
    for spp in (2, 8, 32):
        imgs, line = [], {"frame": f"{w}x{h}", "spp": spp}
        for name, flags in (("lane_per_pixel", 0), ("lane_per_sample", m.MIRT_FLAG_KERNEL_STRIP)):
            p = m.make_params(w, h, spp, flags=flags)
            imgs.append(ctx.render(p))
            ctx.stats()
            import torch
            out = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
            for _ in range(a.launches):
                ctx.render_device(p, out.data_ptr(), out.numel(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            st = ctx.stats()
            ms = st["kernel_ms_total"] / st["launches"]
            line[name] = {"kernel": ctx.last_kernel(), "kernel_us": round(ms * 1e3, 2), "msamples_per_s": round(w * h * spp / ms / 1e3, 1)}
        line["images_equal"] = bool(np.array_equal(imgs[0], imgs[1]))
        rows.append(line)
        print(json.dumps(line), flush=True)
ctx.close()
