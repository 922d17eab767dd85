#!/usr/bin/env python3
"""Like ab_libs.py, but WALL time of N launches queued back to back on one stream (mirt_ctx_render_device into device memory, one
synchronize at the end) -- what consecutive bench steps or interactive frames cost, launch gaps and dispenser set-up included.

    python tools/wall_ab.py [--scene S] [--mode pt|parity] [--size WxH] [--spp N] [--launches 50] [--rounds 7] lib_a.so lib_b.so ...

`--scene layer_scene --mode parity --size 800x600 --spp 2 --launches 200` is the reference's own loop (`Layer::set_data` at its operating
point); more than 64 launches per round cross the context's ring of launch slots.  A build older than the header (fewer exports) is bound
with the symbols it has.
"""
import argparse
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch  # noqa: E402
import weekend_raytracer_wgpu_amd as m  # noqa: E402
from weekend_raytracer_wgpu_amd import _abi  # noqa: E402
from helpers import layer_scene_data, scene_data  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="three_spheres")
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--mode", default="pt")
ap.add_argument("--spp", type=int, default=4)
ap.add_argument("--launches", type=int, default=50)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
w, h = map(int, a.size.split("x"))
sd = layer_scene_data(w, h) if a.scene == "layer_scene" else scene_data(a.scene, w, h)
p = m.make_params(w, h, a.spp, mode=m.MIRT_MODE_PT if a.mode == "pt" else m.MIRT_MODE_PARITY, num_bounces=8)
out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
stream = torch.cuda.Stream()
libs = []
import os  # noqa: E402
for spec in a.libs:                                       # lib.so@NAME=VALUE[,NAME=VALUE]: knobs read by mirt_ctx_create, as in ab_libs.py
    path, _, envs = spec.partition("@")
    lib = C.CDLL(str(Path(path).resolve()))
    _abi.bind(lib, {k: v for k, v in _abi.SYMBOLS.items() if hasattr(lib, k)})
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
wall = {n: [] for n, _, _ in libs}
kern = {n: [] for n, _, _ in libs}
for r in range(a.rounds + 1):
    for name, lib, ctx in libs:
        st = _abi.MirtStats()
        lib.mirt_ctx_get_stats(ctx, C.byref(st))
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.launches):
            rc = lib.mirt_ctx_render_device(ctx, C.byref(p), C.c_void_p(out.data_ptr()), out.numel(), C.c_void_p(stream.cuda_stream))
            assert rc == 0, lib.mirt_last_error()
        stream.synchronize()
        t = time.perf_counter() - t0
        lib.mirt_ctx_get_stats(ctx, C.byref(st))
        if r:
            wall[name].append(1e6 * t / a.launches)
            kern[name].append(1e3 * st.kernel_ms_total / max(1, st.launches))
base = np.median(wall[libs[0][0]])
for name, lib, ctx in libs:
    print(f"{name:44s} wall per launch: median {np.median(wall[name]):9.2f} us  min {np.min(wall[name]):9.2f} us  {100 * np.median(wall[name]) / base:6.1f} %   "
          f"kernel (HIP events) {np.median(kern[name]):9.2f} us   {lib.mirt_ctx_last_kernel(ctx).decode()}")
    lib.mirt_ctx_destroy(ctx)
