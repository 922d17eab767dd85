#!/usr/bin/env python3
"""Step statistics of the pooled kernel's grid build (diagnosis build of the library, -DMIRT_DIAG_STEPS):

    make -C weekend-raytracer-wgpu_amd/csrc -j3 OUT=../../tools/_scratch/libs/libmirt_diag.so OBJDIR=build/obj_diag EXTRA=-DMIRT_DIAG_STEPS
    python tools/step_stats.py [--lib tools/_scratch/libs/libmirt_diag.so] [--scene rtiow_final] [--size 1920x1080] [--spp 64]

One counting launch (MIRT_FLAG_COUNT_WORK | MIRT_FLAG_COUNT_GRID): steps and paths per step by kind (scatter / generate / resumed
walk), fast-forwarded steps, and the outcomes of the traces (cut walks, hits, misses).  The library prints the raw counters; this
script adds the ratios."""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import weekend_raytracer_wgpu_amd as m  # noqa: E402
from weekend_raytracer_wgpu_amd import _abi  # noqa: E402
from helpers import scene_data  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=str(ROOT / "tools/_scratch/libs/libmirt_diag.so"))
ap.add_argument("--scene", default="rtiow_final")
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--flags", type=lambda v: int(v, 0), default=0, help="extra MIRT_FLAG_* bits (e.g. 0x20 = force the pooled kernel)")
ap.add_argument("--plain", action="store_true", help="a plain launch (no counting build): for the -DMIRT_DIAG_STAMPS build, which prints MIRT_STAMPS")
a = ap.parse_args()
w, h = map(int, a.size.split("x"))
lib = C.CDLL(str(Path(a.lib).resolve()))
_abi.bind(lib)
ctx = C.c_void_p()
assert lib.mirt_ctx_create(0, C.byref(ctx)) == 0, lib.mirt_last_error()
sd = scene_data(a.scene, w, h)
sc = sd.as_c()
assert lib.mirt_ctx_set_scene(ctx, C.byref(sc)) == 0, lib.mirt_last_error()
p = m.make_params(w, h, a.spp, mode=m.MIRT_MODE_PT, num_bounces=8, flags=a.flags | (0 if a.plain else (m.MIRT_FLAG_COUNT_WORK | m.MIRT_FLAG_COUNT_GRID)))
out = np.empty((h, w, 4), np.uint8)
assert lib.mirt_ctx_render(ctx, C.byref(p), out.ctypes.data_as(C.c_void_p), out.nbytes) == 0, lib.mirt_last_error()
st = _abi.MirtStats()
lib.mirt_ctx_get_stats(ctx, C.byref(st))          # the diagnosis build prints "MIRT_DIAG ..." to stderr here
if a.plain:
    print(json.dumps({"kernel": lib.mirt_ctx_last_kernel(ctx).decode(), "kernel_ms": st.kernel_ms}))
    lib.mirt_ctx_destroy(ctx)
    sys.exit(0)
print(json.dumps({"kernel": lib.mirt_ctx_last_kernel(ctx).decode(), "samples": st.samples, "rays": st.rays, "tests": st.sphere_tests,
                  "steps": st.wave_iterations, "fresh_traces": st.lane_iterations, "cells": st.grid_cells, "wave_cells": st.grid_wave_cells,
                  "rays_per_sample": st.rays / st.samples, "tests_per_ray": st.sphere_tests / st.rays, "cells_per_ray": st.grid_cells / st.rays,
                  "walk_lane_use": st.grid_cells / (64.0 * st.grid_wave_cells), "steps_per_64_samples": 64.0 * st.wave_iterations / st.samples}))
lib.mirt_ctx_destroy(ctx)
