#!/usr/bin/env python3
"""A/B the path-traced kernel variants on the bench workload in ONE process, interleaved rounds
(cdna_hip_programming.md §5.4 rule 24).  Usage: python tools/ab_pool.py [spp] [rounds]"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import weekend_raytracer_wgpu_amd as m  # noqa: E402
from helpers import scene_data  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
scene = sys.argv[3] if len(sys.argv) > 3 else "three_spheres"
w, h = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1920, 1080)
ctx = m.Context(0)
ctx.set_scene(scene_data(scene, w, h))
variants = [("strip", m.MIRT_FLAG_KERNEL_STRIP, None)] + [(f"pool{c}", m.MIRT_FLAG_KERNEL_POOL, c) for c in (0, 4)]
times = {n: [] for n, _, _ in variants}
ref = None
for r in range(rounds + 1):
    for name, flag, cfg in variants:
        if cfg is not None:
            os.environ["MIRT_POOL_CONFIG"] = str(cfg)
        p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, flags=flag)
        img = ctx.render(p)
        t = ctx.stats()["kernel_ms"]
        if r == 0:
            if ref is None:
                ref = img
            assert np.array_equal(img, ref), name
            continue
        times[name].append(t)
util = {}
for name, flag, cfg in variants:
    if cfg is not None:
        os.environ["MIRT_POOL_CONFIG"] = str(cfg)
    ctx.render(m.make_params(w, h, min(spp, 100), mode=m.MIRT_MODE_PT, num_bounces=8, flags=flag | m.MIRT_FLAG_COUNT_WORK))
    st = ctx.stats()
    util[name] = st["lane_iterations"] / max(1, 64 * st["wave_iterations"])
for name, ts in times.items():
    print(f"lane-util {util[name]:.3f} ", end="")
    print(f"{name:8s} median {np.median(ts):8.2f} ms  min {np.min(ts):8.2f} ms  -> {w * h * spp / np.median(ts) / 1e3:9.1f} Msamples/s")
