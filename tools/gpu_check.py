#!/usr/bin/env python3
"""Quick on-GPU sanity run (developer tool, not a test): HIP path vs CPU oracle on small frames,
then a timing of the bench workload at reduced spp.  Usage on the GPU box:
    python tools/gpu_check.py [--full]
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]

import weekend_raytracer_wgpu_amd as m  # noqa: E402
from weekend_raytracer_wgpu_amd import scenes  # noqa: E402
import oracle_binding as ob  # noqa: E402


def scene_data(name: str, w: int, h: int) -> "m.SceneData":
    sc, cam = scenes.CONFIGS[name]()
    mats, tex = m.flatten_materials(sc.materials)
    return m.SceneData(m.GpuCamera.new(cam, (w, h)).c, [s.to_c() for s in sc.spheres], mats, tex)


def layer_scene_data(w: int, h: int) -> "m.SceneData":
    rp = m.RenderParams(camera=m.FlyCameraController.default().renderer_camera(), viewport_size=(w, h))
    L = m.Layer.new([w, h], rp)
    L.set_global_data()
    return L.scene_data()


def compare(tag: str, a: np.ndarray, b: np.ndarray) -> bool:
    diff = (a != b).any(axis=-1)
    n = int(diff.sum())
    mx = int(np.abs(a.astype(int) - b.astype(int)).max()) if n else 0
    print(f"  {tag}: {'OK  ' if n == 0 else 'FAIL'} mismatching pixels {n}/{diff.size} max|d|={mx}", flush=True)
    if n:
        ys, xs = np.nonzero(diff)
        for y, x in list(zip(ys, xs))[:5]:
            print(f"    ({x},{y}) gpu={a[y, x].tolist()} oracle={b[y, x].tolist()}")
    return n == 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    args = ap.parse_args()
    ok = True
    ctx = m.Context(0)

    print("== parity mode (layer.rs) ==", flush=True)
    for (w, h, spp) in [(160, 120, 1), (160, 120, 21), (200, 150, 70), (333, 77, 2)]:
        sd = layer_scene_data(w, h)
        ctx.set_scene(sd)
        p = m.make_params(w, h, spp)
        g = ctx.render(p)
        o = ob.render(sd, p)
        ok &= compare(f"Layer::scene {w}x{h} spp{spp} ({ctx.stats()['kernel_ms']:.3f} ms)", g, o)

    print("== path-traced mode (wgsl behaviours) ==", flush=True)
    for name, w, h, spp in [("single_sphere", 96, 64, 8), ("three_spheres", 128, 72, 16), ("earth", 128, 72, 16),
                            ("main_rs_scene", 128, 72, 70), ("rtiow_final", 64, 36, 4)]:
        sd = scene_data(name, w, h)
        ctx.set_scene(sd)
        for flags in (0, m.MIRT_FLAG_NO_TONEMAP | m.MIRT_FLAG_NO_SRGB):
            p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, flags=flags | m.MIRT_FLAG_COUNT_WORK)
            g = ctx.render(p)
            gs = ctx.stats()
            o = ob.render(sd, p)
            os_ = ob.stats()
            ok &= compare(f"{name} {w}x{h} spp{spp} flags={flags} ({gs['kernel_ms']:.3f} ms)", g, o)
            keys = ["rays", "sphere_tests", "roots", "hits", "scatter", "sky_misses"]
            same = all(gs[k] == os_[k] for k in keys)
            ok &= same
            print(f"    work counters {'equal' if same else 'DIFFER'}: gpu={[gs[k] for k in keys]} oracle={[os_[k] for k in keys]}"
                  f" lane-util={gs['lane_iterations'] / max(1, 64 * gs['wave_iterations']):.3f}", flush=True)

    print("== timing: config 3 (three_spheres PT 1920x1080) ==", flush=True)
    w, h = 1920, 1080
    sd = scene_data("three_spheres", w, h)
    ctx.set_scene(sd)
    ref = None
    for spp in ([100, 1000] if args.full else [100]):
        for name, kflag in (("strip", m.MIRT_FLAG_KERNEL_STRIP), ("pool", m.MIRT_FLAG_KERNEL_POOL)):
            p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, flags=kflag)
            ctx.render(p)
            img = ctx.render(p)
            st = ctx.stats()
            p.flags |= m.MIRT_FLAG_COUNT_WORK
            ctx.render(p)
            sc = ctx.stats()
            print(f"  spp={spp} {name}: kernel {st['kernel_ms']:.2f} ms, {w * h * spp / st['kernel_ms'] / 1e3:.1f} Msamples/s, "
                  f"lane-util {sc['lane_iterations'] / max(1, 64 * sc['wave_iterations']):.3f}, mean rgb {img[..., :3].reshape(-1, 3).mean(0)}", flush=True)
            if name == "strip":
                ref = img
            else:
                ok &= compare(f"pool vs strip 1080p spp{spp}", img, ref)
    print("== timing: parity Layer::scene 1920x1080 ==", flush=True)
    sd = layer_scene_data(w, h)
    ctx.set_scene(sd)
    for spp in (2, 100, 1000):
        p = m.make_params(w, h, spp)
        ctx.render(p)
        st = ctx.stats()
        print(f"  spp={spp}: kernel {st['kernel_ms']:.3f} ms, {w * h * spp / st['kernel_ms'] / 1e3:.1f} Msamples/s", flush=True)
    ctx.close()
    print("ALL OK" if ok else "FAILURES", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
