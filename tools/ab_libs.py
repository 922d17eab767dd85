#!/usr/bin/env python3
"""A/B several builds of libmirt.so on one workload in ONE process, interleaved rounds (cdna_hip_programming.md
§5.4 rule 24): kernel time from the library's own HIP events, images compared with the first build's.

    python tools/ab_libs.py [--scene three_spheres] [--size 1920x1080] [--spp 1000] [--rounds 5] [--flags 0]
                            [--allow-diff] lib_a.so lib_b.so lib_b.so@MIRT_PINHOLE=0 ...

`lib.so@NAME=VALUE[,NAME=VALUE]` creates that build's context with the tuning variables set (they are read once, in
mirt_ctx_create), so one build can appear several times with different knobs.

Builds are usually produced with `make -C weekend-raytracer-wgpu_amd/csrc OUT=../../tools/_scratch/libs/libmirt_x.so EXTRA=-D...`.
"""
import argparse
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import weekend_raytracer_wgpu_amd as m  # noqa: E402  (also loads torch's HIP runtime first)
from weekend_raytracer_wgpu_amd import _abi  # noqa: E402
from helpers import layer_scene_data, scene_data  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="three_spheres")
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--spp", type=int, default=1000)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--flags", type=lambda s: int(s, 0), default=0)
ap.add_argument("--mode", default="pt")
ap.add_argument("--bounces", type=int, default=8)
ap.add_argument("--allow-diff", action="store_true")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
w, h = map(int, a.size.split("x"))
sd = layer_scene_data(w, h) if a.scene == "layer_scene" else scene_data(a.scene, w, h)
p = m.make_params(w, h, a.spp, mode=m.MIRT_MODE_PT if a.mode == "pt" else m.MIRT_MODE_PARITY, num_bounces=a.bounces, flags=a.flags)

libs = []
for spec in a.libs:
    path, _, envs = spec.partition("@")
    lib = C.CDLL(str(Path(path).resolve()))
    _abi.bind(lib, {k: v for k, v in _abi.SYMBOLS.items() if hasattr(lib, k)})       # a build older than the header has fewer exports
    ctx = C.c_void_p()
    knobs = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
    saved = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        assert lib.mirt_ctx_create(0, C.byref(ctx)) == 0, lib.mirt_last_error()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    sc = sd.as_c()
    assert lib.mirt_ctx_set_scene(ctx, C.byref(sc)) == 0, lib.mirt_last_error()
    libs.append((Path(path).name + ("@" + envs if envs else ""), lib, ctx))


def render(lib, ctx):
    out = np.empty((h, w, 4), np.uint8)
    rc = lib.mirt_ctx_render(ctx, C.byref(p), out.ctypes.data_as(C.c_void_p), out.nbytes)
    assert rc == 0, lib.mirt_last_error()
    st = _abi.MirtStats()
    lib.mirt_ctx_get_stats(ctx, C.byref(st))
    return out, st.kernel_ms


times = {n: [] for n, _, _ in libs}
ref = None
for r in range(a.rounds + 1):
    for name, lib, ctx in libs:
        img, t = render(lib, ctx)
        if r == 0:
            if ref is None:
                ref = img
            elif not np.array_equal(img, ref):
                d = np.abs(img.astype(int) - ref.astype(int))
                msg = f"{name}: image differs from {libs[0][0]}: {int((d.max(-1) > 0).sum())} px, max |d| {int(d.max())}"
                if not a.allow_diff:
                    raise SystemExit(msg)
                print(msg)
            continue
        times[name].append(t)
base = np.median(times[libs[0][0]])
for name, lib, ctx in libs:
    ts = times[name]
    kn = lib.mirt_ctx_last_kernel(ctx).decode()
    print(f"{name:40s} median {np.median(ts):8.3f} ms  min {np.min(ts):8.3f} ms  {100 * np.median(ts) / base:6.1f} %  "
          f"-> {w * h * a.spp / np.median(ts) / 1e3:9.1f} Msamples/s   {kn}")
    lib.mirt_ctx_destroy(ctx)
