#!/usr/bin/env python3
"""Evidence for BASELINE config 4's "LDS texel tiles" (SURVEY §8f-3): how many of the earth texture's texel fetches
would a 64 x 32 texel tile (8 KB as RGBX8) per wave serve, for the camera rays' hits and for later bounces?

Runs the counting build of an experiment library (make ... EXTRA=-DMIRT_PROBE_TEXELS, see below) on config 4: per
shading step it takes the tile of the wave's first image-texture fetch as "the tile the wave would have loaded" and
counts the fetches of that step that fall into it.  Writes profiles/<tag>_texel_tiles.json.

    make -C weekend-raytracer-wgpu_amd/csrc -j3 OUT=../../tools/_scratch/libs/libmirt_ptex.so OBJDIR=build/obj_ptex EXTRA=-DMIRT_PROBE_TEXELS
    python tools/texel_probe.py [tag]        (on the GPU box)
"""
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

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
lib = C.CDLL(str(ROOT / "tools/_scratch/libs/libmirt_ptex.so"))
_abi.bind(lib)
w, h, spp = 1920, 1080, 100
sd = scene_data("earth", w, h)
ctx = C.c_void_p()
assert lib.mirt_ctx_create(0, C.byref(ctx)) == 0
sc = sd.as_c()
assert lib.mirt_ctx_set_scene(ctx, C.byref(sc)) == 0
p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, flags=m.MIRT_FLAG_COUNT_WORK | m.MIRT_FLAG_KERNEL_POOL)
out = np.empty((h, w, 4), np.uint8)
assert lib.mirt_ctx_render(ctx, C.byref(p), out.ctypes.data_as(C.c_void_p), out.nbytes) == 0
st = _abi.MirtStats()
lib.mirt_ctx_get_stats(ctx, C.byref(st))
prim, prim_hit, sec, sec_hit = st.grid_cells, st.grid_wave_cells, st.lane_iterations, st.wave_iterations   # probe build's aliases
res = {
    "workload": f"BASELINE configs[3] (earth sphere over the checker ground), {w}x{h}, {spp} spp, pool kernel, counting build + MIRT_PROBE_TEXELS",
    "tile": "64 x 32 texels of the 1024 x 512 earth map (8 KB as RGBX8); the tile of a step's first image-texture fetch",
    "samples": w * h * spp,
    "image_texture_fetches": {"camera_ray_hits": prim, "later_bounces": sec, "per_sample": round((prim + sec) / (w * h * spp), 4)},
    "served_by_the_step_tile": {"camera_ray_hits": prim_hit, "later_bounces": sec_hit},
    "tile_hit_rate_pct": {"camera_ray_hits": round(100.0 * prim_hit / max(1, prim), 2), "later_bounces": round(100.0 * sec_hit / max(1, sec), 2),
                          "all": round(100.0 * (prim_hit + sec_hit) / max(1, prim + sec), 2)},
}
print(json.dumps(res, indent=1))
for d in ("profiles", "gpurun_out"):      # gpurun_out/ is what travels back from the GPU box
    (ROOT / d).mkdir(exist_ok=True)
    (ROOT / d / f"{tag}_texel_tiles.json").write_text(json.dumps(res, indent=1) + "\n")
lib.mirt_ctx_destroy(ctx)
