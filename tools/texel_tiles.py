#!/usr/bin/env python3
"""The LDS texel-tile build (MIRT_FLAG_TEXEL_TILES) on BASELINE configs[3] (earth sphere, 1920x1080): kernel time of the
default and the tile build (interleaved rounds, the library's HIP events), that the two images are identical, and the
tile hit rates of camera-ray hits and later bounces from the counting tile build.  Writes gpurun_out/texel_tiles[_TAG].json
(copied to profiles/ by hand).

    python tools/texel_tiles.py [--spp 1000] [--rounds 5] [--lib path/to/libmirt.so] [--tag TAG]
"""
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
ap.add_argument("--spp", type=int, default=1000)
ap.add_argument("--count-spp", type=int, default=100)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--lib", default=str(m.LIB_PATH))
ap.add_argument("--tag", default="")
a = ap.parse_args()
w, h = 1920, 1080
lib = C.CDLL(str(Path(a.lib).resolve()))
_abi.bind(lib)
ctx = C.c_void_p()
assert lib.mirt_ctx_create(0, C.byref(ctx)) == 0
sd = scene_data("earth", w, h)
sc = sd.as_c()
assert lib.mirt_ctx_set_scene(ctx, C.byref(sc)) == 0


def render(spp, flags):
    p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, flags=flags)
    out = np.empty((h, w, 4), np.uint8)
    assert lib.mirt_ctx_render(ctx, C.byref(p), out.ctypes.data_as(C.c_void_p), out.nbytes) == 0, lib.mirt_last_error()
    st = _abi.MirtStats()
    lib.mirt_ctx_get_stats(ctx, C.byref(st))
    return out, st, lib.mirt_ctx_last_kernel(ctx).decode()


variants = {"default": 0, "tiles": m.MIRT_FLAG_TEXEL_TILES}
times = {k: [] for k in variants}
images, kernels = {}, {}
for r in range(a.rounds + 1):
    for k, fl in variants.items():
        img, st, kn = render(a.spp, fl)
        if r == 0:
            images[k], kernels[k] = img, kn
        else:
            times[k].append(st.kernel_ms)
_, st, kcount = render(a.count_spp, m.MIRT_FLAG_TEXEL_TILES | m.MIRT_FLAG_COUNT_WORK)
f, t = list(st.texel_fetches), list(st.texel_tile_hits)
res = {
    "workload": f"BASELINE configs[3] (earth sphere over the checker ground), {w}x{h}, {a.spp} spp; hit rates from {a.count_spp} spp of the counting tile build",
    "library": Path(a.lib).name,
    "kernels": kernels,
    "counting_kernel": kcount,
    "kernel_ms_median": {k: float(np.median(v)) for k, v in times.items()},
    "msamples_per_s": {k: w * h * a.spp / float(np.median(v)) / 1e3 for k, v in times.items()},
    "tiles_vs_default_time": float(np.median(times["tiles"]) / np.median(times["default"])),
    "images_identical": bool(np.array_equal(images["default"], images["tiles"])),
    "image_texel_fetches": {"camera_ray_hits": f[0], "later_bounces": f[1], "per_sample": (f[0] + f[1]) / (w * h * a.count_spp)},
    "served_by_the_lds_tile": {"camera_ray_hits": t[0], "later_bounces": t[1]},
    "tile_hit_rate_pct": {"camera_ray_hits": 100.0 * t[0] / max(1, f[0]), "later_bounces": 100.0 * t[1] / max(1, f[1]),
                          "all": 100.0 * (t[0] + t[1]) / max(1, f[0] + f[1])},
}
out = ROOT / "gpurun_out" / f"texel_tiles{('_' + a.tag) if a.tag else ''}.json"
out.parent.mkdir(exist_ok=True)
out.write_text(json.dumps(res, indent=1))
print(json.dumps(res, indent=1))
