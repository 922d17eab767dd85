#!/usr/bin/env python3
"""bench.py — Msamples/s of the per-pixel ray-trace path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5|parity]

One process per GPU.  With N > 1 and no WORLD_SIZE in the environment this script LAUNCHES its own ranks
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, fresh child
processes, started before this process has touched torch or the GPU) and returns their exit code; started
under torch.distributed.run it is one of those ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env).

Default workload = BASELINE.json configs[2], the configuration `metric` is quoted on: path-traced mode,
three-sphere diffuse+metal+dielectric scene (reference src/main.rs:539-541), 1920x1080, 1000 spp, 8 bounces,
synthetic scene tables already resident in HBM.  One "step" = one full frame: every rank renders its row
tiles, ONE gather brings them to rank 0, rank 0 assembles the RGBA8 frame.  value = pixels x spp x K / wall
seconds / 1e6 (whole job, all ranks; total work is fixed as N grows -> "strong" scaling).
`--config` selects the other BASELINE configurations (same JSON shape; `config.workload` names them).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      fp32 vector-ALU roofline of the render kernel (north_star: "scalar-ray fp32 FMA, no MFMA"):
                achieved = ALGORITHMIC flops per launch (work counters of one counting launch x the per-unit
                figures of SURVEY §8d / DESIGN.md) / the kernel's average duration measured with HIP events on
                the launch stream inside the timed region.
  cpu_baseline  the CPU oracle (a port: the reference is Rust and cannot be built here) timed on the host cores
                on a bounded sample of the same workload (rank 0, N = 1 only; best of 3 runs on as many threads as
                the cgroup's CPU quota allows, `median` / `min` beside it), plus `layer_rs`: the reference's
                own CPU loop (`Layer::set_data`, parity mode) restated, faithful (with its per-sample
                world.clone() allocations) and clean, on 1 thread and on all cores.
  verified_rows rows of the frame the timed region produced -- for N > 1 the frame GATHERED over RCCL on rank 0 --
                compared byte for byte with oracle rows rendered at the full sample count (the checker, after the
                clock has stopped).  Bands of rows around a few probe rows, so that the checker uses the host's cores.
                Where one oracle row would take minutes (BASELINE configs[4]: 3840x2160 x 4000 spp) the bands are held
                against rank 0's OWN single-GPU render of those rows instead (`against: "rank0_single_gpu_render"`:
                partition + gather + de-interleave at full size, not kernel parity); N > 1 lines add
                `verified_whole_frame`, every pixel of the gathered frame against rank 0's render of the whole frame.
  gather_ms     N > 1: one gather + de-interleave on its own (no render), timed after the timed region.

`--dry-run` rehearses the N-rank path without a GPU (gloo, CPU tensors, a pattern renderer instead of the HIP
kernel): it proves that the ranks start, partition, gather and assemble; it measures nothing.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path[:0] = [str(ROOT)]

WIDTH, HEIGHT, SPP, BOUNCES = 1920, 1080, 1000, 8            # the headline configuration (BASELINE configs[2])
PEAK_FP32_VECTOR_TFLOPS = 157.3     # MI355X_MICROARCH.md "Peak FP32 (vector)": 256 CU x 256 flop/clk x 2.4 GHz
PEAK_HBM_GBS = 8000.0

# BASELINE.json `configs` as bench workloads (config 1 is the CPU-only plumbing case: a parity test, not a bench line)
CONFIGS = {
    "2": dict(scene="single_sphere", width=1920, height=1080, spp=100, mode="pt",
              workload="BASELINE configs[1]: single unit sphere (metal), 1920x1080, 100 spp, path-traced mode, 8 bounces"),
    "3": dict(scene="three_spheres", width=WIDTH, height=HEIGHT, spp=SPP, mode="pt",
              workload="BASELINE configs[2]: path-traced 3-sphere scene, checker-diffuse ground + glass + metal "
                       "(src/main.rs:539-541), 1920x1080, 1000 spp, 8 bounces, gradient sky, seed 0"),
    "4": dict(scene="earth", width=1920, height=1080, spp=1000, mode="pt",
              workload="BASELINE configs[3]: earth-textured sphere (assets/earthmap.jpeg, 1024x512 texels) over the checker "
                       "ground, 1920x1080, 1000 spp, 8 bounces"),
    "5": dict(scene="rtiow_final", width=3840, height=2160, spp=500, mode="pt",
              workload="BASELINE configs[4] CUT TO 500 spp: RTIOW final scene (484 spheres, generated, seed 0x5EED), 3840x2160, "
                       "8 bounces; 500 spp = the samples one GPU renders of the 8-GPU 4000-spp job, so that a step lasts < 1 s"),
    "parity": dict(scene="layer_scene", width=1920, height=1080, spp=1000, mode="parity",
                   workload="parity mode (layer.rs semantics, bit-faithful) on the 6-sphere Layer::scene (layer.rs:90-123), "
                            "1920x1080, 1000 spp nominal (the reference's loop returns at the first terminating sample)"),
}
VERIFY_ROWS = {1080: (0, 269, 540, 811, 1079), 2160: (0, 1080, 1500, 2159)}
VERIFY_CHECKER = {"auto": "auto", "oracle": "oracle", "self": "rank0_single_gpu_render"}     # --verify-against -> verify_rows(against=)

# algorithmic flops per unit of work (DESIGN.md "Algorithmic work")
FLOPS = {
    "sample": 17 + 12 + 6,          # ray generation + thin-lens offset + throughput*colour & fixed-point convert
    "parity_sample": 17,            # make_ray + the two jitter adds
    "test": 23,                     # ray-sphere test up to the discriminant
    "root": 4,                      # sqrt, negate, add/sub, scale
    "hit": 21,                      # p, n, u/v
    "scatter": [45, 35, 50, 54, 35],  # lambertian, metal, dielectric, checkerboard(+9 over lambertian), missing
    "parity_scatter": 20,           # scatter_metal: unit_vertor + reflect + dot
    "sky_gradient": 18,
    "sky_hosek": 360,
    "grid_ray": 45,                 # clip against the grid box, entry cell, per-axis crossing parameters and increments
    "grid_cell": 10,                # exit parameter, stop test, axis choice, step
}


def algorithmic_flops(st: dict, hosek: bool = False, mode: str = "pt", grid: bool = False) -> float:
    """Flops the REFERENCE formulas need for the work counted in `st` (not the instructions executed)."""
    if mode == "parity":
        return float(FLOPS["parity_sample"] * st["lane_iterations"] + FLOPS["test"] * st["sphere_tests"] + FLOPS["root"] * st["roots"]
                     + FLOPS["hit"] * st["hits"] + FLOPS["parity_scatter"] * st["scatter"][1])
    f = FLOPS["sample"] * st["samples"] + FLOPS["test"] * st["sphere_tests"] + FLOPS["root"] * st["roots"]
    f += FLOPS["hit"] * st["hits"] + sum(w * n for w, n in zip(FLOPS["scatter"], st["scatter"]))
    f += (FLOPS["sky_hosek"] if hosek else FLOPS["sky_gradient"]) * st["sky_misses"]
    if grid:
        f += FLOPS["grid_ray"] * st["rays"] + FLOPS["grid_cell"] * st.get("grid_cells", 0)
    return float(f)


def profiled_traffic(kernel: str, workload: str = None):
    """HBM bytes per launch of the render kernel from the newest committed rocprofv3 PMC summary
    (profiles/*_pmc_summary.json, written by tools/summarize_profile.py: separate --pmc passes,
    FETCH_SIZE x2 gfx950 correction) of this kernel ON this workload (several configs share a kernel).
    None if no such summary exists."""
    best = None
    for f in sorted((ROOT / "profiles").glob("*_pmc_summary.json")):
        try:
            d = json.loads(f.read_text())
        except (OSError, ValueError):
            continue
        if workload is not None and d.get("workload") != workload:
            continue
        if d.get("kernel_id") == kernel and "hbm_bytes_per_launch" in d.get("derived", {}):
            best = (d["derived"]["hbm_bytes_per_launch"], f.name, d["derived"].get("valu_issue_busy_pct"),
                    d["derived"].get("valu_lane_utilization_pct"))
    return best


def kernel_uses_grid(name: str) -> bool:
    """GRID template argument of the kernel names `Context.last_kernel()` reports (mirt_ctx_last_kernel)."""
    if "<" not in name:
        return False
    targs = name[name.index("<") + 1:name.rindex(">")].split(",")
    if name.startswith("render_pt_pool_kernel"):          # <threads, slots, min waves, COUNT, HOSEK, NQ, GRID, FLATY>
        return len(targs) > 6 and targs[6] == "true"
    if name.startswith("render_pt_strip_kernel"):
        return targs[2] == "true"
    return False


def kernel_schedule(name: str) -> str:
    """What decides a kernel's lane utilisation: its family and its lane mapping ("pool", "strip/pixel", "strip/sample", ...)."""
    if "<" not in name:
        return name
    fam = name[:name.index("<")].replace("fast_build::", "")
    targs = name[name.index("<") + 1:name.rindex(">")].split(",")
    if fam == "render_pt_strip_kernel":
        return "strip/pixel" if targs[3] == "true" else "strip/sample"
    if fam == "render_pt_stream_kernel":                # lane = pixel, a lane starts its next sample without waiting for the wave
        return "strip/pixel-stream"
    if fam == "render_parity_kernel":
        return "parity/pixel" if len(targs) > 1 and targs[1] == "true" else "parity/sample"
    return "pool" if fam.startswith("render_pt_pool") else fam


def algorithmic_bytes(sd, mode: str, out_bytes: int) -> int:
    """HBM bytes the workload needs per launch: the framebuffer once, the camera, the sphere and material tables
    once, and the texels of the textures the launch can READ -- image textures (larger than 1x1) of materials some sphere
    uses in path-traced mode (1x1 colours travel inside the prepared material table), the one or two texels of
    material_data[2].desc1 in parity mode (layer.rs:345-351 and the row-0 quirk).  Not the whole texel table: config 3
    never reads a texel (8 294 752 B)."""
    tables = 96 + 32 * len(sd.spheres) + 32 * len(sd.materials)
    if mode == "parity":
        return out_bytes + tables + 2 * 12
    seen, texel_bytes = set(), 0
    for s in sd.spheres:
        if s.material_idx >= len(sd.materials):
            continue
        mat = sd.materials[s.material_idx]
        for d in (mat.desc1, mat.desc2):
            key = (d.width, d.height, d.offset)
            if d.width * d.height > 1 and d.offset != 0xffffffff and key not in seen:
                seen.add(key)
                texel_bytes += 12 * d.width * d.height
    return out_bytes + tables + texel_bytes


def rank_device_error(world: int, n_devices: int, dry: bool):
    """(exit code, one-line message) if this process cannot run as one of `world` ranks with `n_devices` visible GPUs."""
    if dry:
        return None
    if n_devices == 0:
        return 3, "bench.py: no GPU visible; the render path has no CPU fallback (use --dry-run to rehearse the rank plumbing)"
    if world > n_devices:
        return 2, (f"bench.py: --gpus {world} needs {world} GPUs, {n_devices} visible: one rank per GPU "
                   f"(RCCL refuses two ranks on one device)")
    return None


def build_scene(m, cfg: dict):
    w, h = cfg["width"], cfg["height"]
    if cfg["scene"] == "layer_scene":                     # Layer::new + set_global_data under the default fly camera
        rp = m.RenderParams(camera=m.FlyCameraController.default().renderer_camera(), viewport_size=(w, h))
        layer = m.Layer.new([w, h], rp)
        layer.set_global_data()
        return layer.scene_data()
    scene, cam = m.scenes.CONFIGS[cfg["scene"]]()
    mats, texels = m.flatten_materials(scene.materials)
    gcam = m.GpuCamera.new(cam, (w, h))
    return m.SceneData(gcam.c, [s.to_c() for s in scene.spheres], mats, texels)


def base_params(m, cfg: dict, flags: int = 0):
    mode = m.MIRT_MODE_PT if cfg["mode"] == "pt" else m.MIRT_MODE_PARITY
    return m.make_params(cfg["width"], cfg["height"], cfg["spp"], mode=mode, num_bounces=BOUNCES, flags=flags)


def host_cores() -> int:
    """CPUs this process may actually use: its affinity mask, capped by the cgroup's CPU quota where one is set (a GPU box hands a
    one-GPU job a share of a 256-thread host: 256 OpenMP threads on a 16-CPU quota measure the scheduler, not the code)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:                                                    # cgroup v2
        q, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:                                                # cgroup v1
            q = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            period = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if q > 0 and period > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def cpu_baseline(m, sd, cfg: dict, target_seconds: float = 12.0, repeats: int = 3) -> dict:
    """Time the CPU oracle on a bounded sample of the SAME workload (same frame, fewer spp)."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_binding as ob     # checker / baseline only

    cores = host_cores()
    w, h, full = cfg["width"], cfg["height"], cfg["spp"]
    if cfg["mode"] == "parity":
        lr = layer_rs_baseline(m, ob, sd, w, h, spp=full if full < 64 else 2)
        best = lr["clean_all_cores"]
        return {"value": best["msamples_per_s"], "unit": "Msamples/s", "cores": cores, "kind": "port",
                "sample": f"Layer::scene {w}x{h} at {lr['spp']} spp (the reference's default), clean variant on all cores; see layer_rs",
                "layer_rs": lr}
    # calibrate on a small sample, then REPEATS runs of ~target_seconds / REPEATS each: round 3's single sample on 256 threads (the
    # host's logical CPUs; the job's cgroup quota is 16) wandered by +-14 % from run to run.  host_cores() now honours the quota,
    # and the best of N is what is reported.  (Binding the threads with OMP_PROC_BIND was tried: it pins the MAIN thread too, after
    # which the affinity mask says one CPU.)
    spp, secs = 4, 0.0
    per_run = target_seconds / repeats
    for _ in range(3):
        p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=BOUNCES)
        ob.render(sd, p, n_threads=cores)
        secs = ob.stats()["kernel_ms"] / 1e3
        if secs >= 0.5 * per_run or spp >= full:
            break
        spp = int(max(spp + 1, min(full, round(spp * per_run / max(secs, 1e-3)))))
    p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=BOUNCES)
    runs = []
    for _ in range(repeats):
        ob.render(sd, p, n_threads=cores)
        runs.append(ob.stats()["kernel_ms"] / 1e3)
    rates = sorted(w * h * spp / t / 1e6 for t in runs)
    best, median = rates[-1], rates[len(rates) // 2]
    return {"value": round(best, 3), "unit": "Msamples/s", "cores": cores,
            "kind": "port", "repeats": repeats, "median": round(median, 3), "min": round(rates[0], 3),
            "sample": f"same scene and frame ({w}x{h}, {BOUNCES} bounces) at {spp} spp instead of {full}, best of {repeats} runs "
                      f"({min(runs):.1f} s each; rate is spp-independent); oracle = C restatement, OpenMP over rows, one thread per CPU of the cgroup's quota; "
                      f"the Rust reference cannot be built (no toolchain)"}


def layer_rs_baseline(m, ob, sd_layer, w: int, h: int, spp: int = 2) -> dict:
    """The reference's own CPU loop (`Layer::set_data`, layer.rs:264-282 and callees) restated in C and timed on the
    host: `faithful` reproduces the two world.clone() allocation rounds per sample (layer.rs:332,359), `clean` does not
    allocate.  spp = 2 is the reference's default (mod.rs:605-613); parity-mode work is not proportional to spp (the
    loop returns at the first terminating sample), so the rate is quoted at that default only.  One thread renders the
    bench frame itself; for all cores the same camera renders 4x4 times the pixels, otherwise a 2-spp 1080p frame is
    over before the threads have started (2 M pixels on 256 threads)."""
    cores = host_cores()
    out = {"spp": spp, "scene": "Layer::scene (6 spheres, layer.rs:90-123)", "cores": cores}
    big = m.SceneData(m.GpuCamera.new(m.FlyCameraController.default().renderer_camera(), (4 * w, 4 * h)).c,
                      sd_layer.spheres, sd_layer.materials, sd_layer.texels)
    for name, variant in (("faithful", ob.FAITHFUL), ("clean", ob.CLEAN)):
        for label, nt, sd, fw, fh in (("1_thread", 1, sd_layer, w, h), ("all_cores", cores, big, 4 * w, 4 * h)):
            p = m.make_params(fw, fh, spp)
            reps, secs = 0, 0.0
            while secs < 0.5 and reps < 50:
                ob.render(sd, p, n_threads=nt, variant=variant)
                secs += ob.stats()["kernel_ms"] / 1e3
                reps += 1
            out[f"{name}_{label}"] = {"msamples_per_s": round(fw * fh * spp * reps / secs / 1e6, 3), "threads": nt,
                                      "frame": f"{fw}x{fh}", "repeats": reps}
    return out


def fast_math_line(m, torch, ctx, base, exact_frame, total_samples: int, flops: float, launches: int = 5) -> dict:
    """The OPT-IN fast-math build (MIRT_FLAG_FAST_MATH) on the same workload, outside the timed region and never part
    of `value`: kernel time from HIP events, the same algorithmic flops, and how far its frame strays from the exact
    build's (max |delta| over the channels of a pixel -> number of pixels)."""
    import copy
    import numpy as np
    p = copy.copy(base)
    p.flags |= m.MIRT_FLAG_FAST_MATH
    out = torch.empty_like(exact_frame)
    stream = torch.cuda.current_stream().cuda_stream
    ctx.render_device(p, out.data_ptr(), out.numel(), stream)            # warm-up
    torch.cuda.synchronize()
    ctx.stats()
    for _ in range(launches):
        ctx.render_device(p, out.data_ptr(), out.numel(), stream)
    torch.cuda.synchronize()
    st = ctx.stats()
    ms = st["kernel_ms_total"] / max(1, st["launches"])
    d = (out.to(torch.int16)[..., :3] - exact_frame.to(torch.int16)[..., :3]).abs().amax(dim=-1)
    vals, counts = torch.unique(d, return_counts=True)
    hist = {str(int(v)): int(c) for v, c in zip(vals.tolist(), counts.tolist())}
    tflops = flops / (ms * 1e-3) / 1e12
    return {"kernel": ctx.last_kernel(), "kernel_ms_avg": round(ms, 4), "value_from_kernel_time": round(total_samples / ms / 1e3, 2),
            "unit": "Msamples/s", "roofline_frac": round(tflops / PEAK_FP32_VECTOR_TFLOPS, 4),
            "max_abs_delta_per_pixel_vs_exact": hist,
            "within_1_pct": round(100.0 * (int(hist.get("0", 0)) + int(hist.get("1", 0))) / d.numel(), 4),
            "note": "opt-in build: hardware rcp/rsq/sqrt/sin/cos/exp/log + contraction; no parity claim; not the reported value"}


def texel_tiles_line(m, torch, ctx, base, default_frame, total_samples: int, launches: int = 5) -> dict:
    """The OPT-IN LDS texel-tile build (MIRT_FLAG_TEXEL_TILES; BASELINE configs[3] "LDS texel tiles") on the same workload,
    outside the timed region and never part of `value`: kernel time from HIP events, that its frame equals the default
    build's, and the tile hit rates from one counting launch at a tenth of the samples."""
    import copy
    p = copy.copy(base)
    p.flags |= m.MIRT_FLAG_TEXEL_TILES
    out = torch.empty_like(default_frame)
    stream = torch.cuda.current_stream().cuda_stream
    ctx.render_device(p, out.data_ptr(), out.numel(), stream)            # warm-up
    torch.cuda.synchronize()
    ctx.stats()
    for _ in range(launches):
        ctx.render_device(p, out.data_ptr(), out.numel(), stream)
    torch.cuda.synchronize()
    st = ctx.stats()
    ms = st["kernel_ms_total"] / max(1, st["launches"])
    kernel = ctx.last_kernel()
    pc = copy.copy(p)
    pc.flags |= m.MIRT_FLAG_COUNT_WORK
    pc.spp = max(48, p.spp // 10)
    scratch = torch.empty_like(default_frame)
    ctx.render_device(pc, scratch.data_ptr(), scratch.numel(), stream)
    torch.cuda.synchronize()
    cs = ctx.stats()
    f, t = cs["texel_fetches"], cs["texel_tile_hits"]
    return {"kernel": kernel, "kernel_ms_avg": round(ms, 4), "value_from_kernel_time": round(total_samples / ms / 1e3, 2), "unit": "Msamples/s",
            "frame_identical_to_default_build": bool(torch.equal(out, default_frame)),
            "image_texel_fetches_per_sample": round((f[0] + f[1]) / max(1, cs["samples"]), 4),
            "tile_hit_rate_pct": {"camera_ray_hits": round(100.0 * t[0] / max(1, f[0]), 2), "later_bounces": round(100.0 * t[1] / max(1, f[1]), 2),
                                  "all": round(100.0 * (t[0] + t[1]) / max(1, f[0] + f[1]), 2)},
            "note": "opt-in build: a 16x2-texel LDS window per wave and strip; same texel values, so the same image; not the reported value"}


def steady_state_line(m, torch, ctx, base, w: int, h: int, spp: int = 1000, launches: int = 3) -> dict:
    """Config 2's frame lasts 1.4 ms: ramp-up and tail weigh.  The same scene at 1000 spp, outside the timed region, separates the
    kernel's steady-state rate from that (default schedule for that sample count; flops from a counting launch of its own)."""
    import copy
    p = copy.copy(base)
    p.spp = spp
    out = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ctx.render_device(p, out.data_ptr(), out.numel(), stream)            # warm-up
    torch.cuda.synchronize()
    ctx.stats()
    for _ in range(launches):
        ctx.render_device(p, out.data_ptr(), out.numel(), stream)
    torch.cuda.synchronize()
    st = ctx.stats()
    ms = st["kernel_ms_total"] / max(1, st["launches"])
    kernel = ctx.last_kernel()
    pc = copy.copy(p)
    pc.flags |= m.MIRT_FLAG_COUNT_WORK
    ctx.render_device(pc, out.data_ptr(), out.numel(), stream)
    torch.cuda.synchronize()
    flops = algorithmic_flops(ctx.stats(), mode="pt")
    tf = flops / (ms * 1e-3) / 1e12
    return {"spp": spp, "kernel": kernel, "kernel_ms_avg": round(ms, 4), "value_from_kernel_time": round(w * h * spp / ms / 1e3, 2),
            "unit": "Msamples/s", "roofline_frac": round(tf / PEAK_FP32_VECTOR_TFLOPS, 4),
            "note": "same scene and frame at 1000 spp, outside the timed region: the rate without the ramp-up and tail of a 1 ms launch"}


def parity_schedules_line(m, torch, ctx, base, w: int, h: int, launches: int = 20) -> dict:
    """Parity mode's two schedules on this workload, outside the timed region: lane = pixel (default below 64 spp, the shape of the
    reference's 2-spp operating point) and lane = sample (MIRT_FLAG_KERNEL_STRIP), kernel time from HIP events, frames compared."""
    import copy
    out, frames = {}, []
    stream = torch.cuda.current_stream().cuda_stream
    for name, extra in (("default", 0), ("lane_per_sample_forced", m.MIRT_FLAG_KERNEL_STRIP)):
        p = copy.copy(base)
        p.flags |= extra
        buf = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
        ctx.render_device(p, buf.data_ptr(), buf.numel(), stream)
        torch.cuda.synchronize()
        ctx.stats()
        for _ in range(launches):
            ctx.render_device(p, buf.data_ptr(), buf.numel(), stream)
        torch.cuda.synchronize()
        st = ctx.stats()
        ms = st["kernel_ms_total"] / max(1, st["launches"])
        frames.append(buf)
        out[name] = {"kernel": ctx.last_kernel(), "kernel_us_avg": round(ms * 1e3, 2), "msamples_per_s": round(w * h * base.spp / ms / 1e3, 1)}
    out["frames_identical"] = bool(torch.equal(frames[0], frames[1]))
    return out


def verify_rows(m, sd, cfg: dict, frame_host, budget_seconds: float = 40.0, cpu_msamples_per_s: float = None, oracle_rows=None,
                self_rows=None, against: str = "auto") -> list:
    """Rows of the TIMED frame against checker rows at the full sample count; the checker runs after the clock stopped.

    Two checkers.  `against = "oracle"`: the CPU oracle (`oracle_rows(first, last)` replaces it in the dry run).  It parallelises
    over rows, so every probe is a BAND of rows (up to 16, one per host thread) around a probe row -- the middle of the frame first;
    `cpu_msamples_per_s` (the cpu_baseline leg's rate on all cores) sizes the number of bands to the budget before anything is
    rendered, and the budget is checked again between bands.  `against = "rank0_single_gpu_render"`: `self_rows(first, last)` =
    rank 0 rendering those rows ITSELF through the whole-frame path (row_begin / row_end, no tiles, same sample count).  That is
    not oracle parity -- the kernel's parity is established by the N = 1 line and the GPU tests -- it checks what an N-rank job
    adds: partition + RCCL gather + de-interleave, at the job's full size.  `"auto"` takes the oracle where one of its rows fits
    the budget and the self-render otherwise (BASELINE configs[4]: 3840x2160 x 4000 spp, ten minutes per oracle row), so the line
    of the one job that is DEFINED on 8 GPUs never says "skipped".  Every entry names its checker in `against`."""
    import numpy as np

    w, h = cfg["width"], cfg["height"]
    cores = host_cores()
    band = max(1, min(16, cores, h))
    probes = list(VERIFY_ROWS.get(h, (0, h // 2, h - 1)))
    probes.sort(key=lambda r: abs(r - h // 2))
    checker = against
    note = None
    if checker == "auto":
        checker = "oracle"
    if checker == "oracle" and oracle_rows is None:
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_binding as ob

        def oracle_rows(first, last):
            p = base_params(m, cfg)
            p.row_begin, p.row_end = first, last
            return ob.render(sd, p, n_threads=cores)
        if not cpu_msamples_per_s:                       # no cpu_baseline leg in this run (N > 1): a one-row, 4-spp probe of the oracle's rate
            p = base_params(m, cfg)
            p.spp, p.row_begin, p.row_end = min(4, cfg["spp"]), h // 2, h // 2 + 1
            tp = time.perf_counter()
            ob.render(sd, p, n_threads=1)
            cpu_msamples_per_s = cores * w * p.spp / max(time.perf_counter() - tp, 1e-6) / 1e6
        row_seconds = w * cfg["spp"] / (cpu_msamples_per_s / cores * 1e6)              # one row on one thread
        if row_seconds > 3.0 * budget_seconds:           # e.g. config 5 at 4000 spp: ten minutes per row
            note = (f"one oracle row of this workload takes ~{row_seconds:.0f} s on a host thread (budget {budget_seconds:.0f} s): "
                    f"bands checked against rank 0's own single-GPU render instead")
            if against == "oracle" or self_rows is None:
                return [{"row": None, "against": "oracle", "skipped": note}]
            checker = "rank0_single_gpu_render"
        else:
            probes = probes[:max(1, int(budget_seconds / max(row_seconds, 1e-6)))]
    if checker == "rank0_single_gpu_render":
        if self_rows is None:
            return [{"row": None, "against": checker, "skipped": "no self-render available"}]
        rows_of = self_rows
    else:
        rows_of = oracle_rows
    t0, out = time.perf_counter(), []
    for r in probes:
        if out and time.perf_counter() - t0 > budget_seconds:
            break
        first = max(0, min(r - band // 2, h - band))
        want = rows_of(first, first + band)
        got = frame_host[first:first + band]
        rows_equal = [bool(np.array_equal(got[i], want[i])) for i in range(band)]
        entry = {"row": r, "rows": [first, first + band], "spp": cfg["spp"], "equal": all(rows_equal),
                 "rows_equal": sum(rows_equal), "against": checker}
        if note:
            entry["note"] = note
        out.append(entry)
    return out


def self_render_rows(m, torch, ctx, base, first: int, last: int):
    """Rows [first, last) of the frame rendered by THIS rank alone through the whole-frame path: row_begin / row_end, no tile
    interleave, the same sample count, seed and flags as the timed steps (the checker of `verify_rows(against="rank0_single_gpu_render")`)."""
    import copy
    p = copy.copy(base)
    p.tile_rows, p.n_parts, p.part = 0, 0, 0
    p.row_begin, p.row_end = first, last
    buf = torch.empty((last - first, base.width, 4), dtype=torch.uint8, device=torch.device("cuda", ctx.device))
    ctx.render_device(p, buf.data_ptr(), buf.numel(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return buf.cpu().numpy()


def verify_whole_frame_self(m, torch, ctx, base, frame_dev, est_ms: float, budget_ms: float = 5000.0) -> dict:
    """N > 1: the COMPLETE gathered frame against rank 0's own single-GPU render of the whole frame (after the clock has stopped),
    compared on the device, when that render is affordable (`est_ms` = the ranks' kernel time summed).  Checks partition + gather +
    de-interleave for every pixel of the job; says nothing about the kernel's parity (see verify_rows)."""
    if est_ms > budget_ms:
        return {"against": "rank0_single_gpu_render", "skipped": f"a single-GPU render of the whole frame would take ~{est_ms:.0f} ms (budget {budget_ms:.0f} ms)"}
    import copy
    p = copy.copy(base)
    p.tile_rows, p.n_parts, p.part = 0, 0, 0
    buf = torch.empty_like(frame_dev)
    ctx.render_device(p, buf.data_ptr(), buf.numel(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    diff_rows = (buf != frame_dev).flatten(1).any(dim=1)
    return {"against": "rank0_single_gpu_render", "rows": [0, int(frame_dev.shape[0])], "equal": not bool(diff_rows.any()),
            "rows_differing": int(diff_rows.sum()), "render_ms": round(ctx.stats()["kernel_ms"], 3)}


# ------------------------------------------------------------------------------------------------
# --dry-run: the N-rank path without a GPU (pattern renderer, gloo)
# ------------------------------------------------------------------------------------------------

def pattern_rows(rows, width: int):
    """RGBA8 rows whose bytes are a function of the ABSOLUTE (row, x) only — what a correct partition must reassemble."""
    import numpy as np
    rows = np.asarray(rows, dtype=np.uint32)[:, None]
    xs = np.arange(width, dtype=np.uint32)[None, :]
    img = np.empty((rows.shape[0], width, 4), dtype=np.uint8)
    img[..., 0] = (rows * 7 + xs) & 255
    img[..., 1] = (rows ^ (xs >> 2)) & 255
    img[..., 2] = ((rows >> 3) + (xs >> 5)) & 255
    img[..., 3] = 255
    return img


class PatternContext:
    """Stand-in for `Context` in --dry-run: fills a rank's compact buffer with pattern_rows() of the rows the
    params select, and de-interleaves on the host with the same `assemble_host` the CPU tests use."""

    def __init__(self, m):
        self.m, self.device, self.launches = m, 0, 0

    @staticmethod
    def _view(ptr: int, shape):
        import numpy as np
        n = int(np.prod(shape))
        return np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(ptr)).reshape(shape)

    def render_device(self, params, d_ptr: int, nbytes: int, stream=None) -> None:
        m = self.m
        rows = [m.params_out_row_index(params, i) for i in range(m.params_out_rows(params))]
        assert nbytes >= len(rows) * params.width * 4
        self._view(d_ptr, (len(rows), params.width, 4))[:] = pattern_rows(rows, params.width)
        self.launches += 1

    def deinterleave_device(self, params, d_parts: int, part_stride: int, d_out: int, out_nbytes: int, stream=None) -> None:
        m = self.m
        world, w = params.n_parts, params.width
        parts = self._view(d_parts, (world, part_stride // (w * 4), w, 4))
        band = m.multi_gpu.band_rows(params)
        self._view(d_out, (band, w, 4))[:] = m.multi_gpu.assemble_host(parts, params, world, params.tile_rows)

    def stats(self) -> dict:
        n, self.launches = self.launches, 0
        return {"kernel_ms_total": 0.0, "launches": n, "kernel_ms": 0.0}

    def close(self) -> None:
        pass


# ------------------------------------------------------------------------------------------------
# launcher: N > 1 from a bare shell
# ------------------------------------------------------------------------------------------------

class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when a communicator comes up; rank 0's stdout is reserved for the ONE JSON line.  File
    descriptor 1 points at stderr while the process group and its first collective are set up."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv) -> int:
    """Start N fresh rank processes through torch.distributed.run and wait for them.  This process has not
    imported torch and never touches the GPU; the ranks are ordinary children (no exec of a GPU process)."""
    env = dict(os.environ)
    if not args.dry_run:
        # one rank per GPU: count the devices in a child (this process never imports torch) and refuse here, with one line,
        # what RCCL would refuse inside init_process_group ("Duplicate GPU detected")
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], env=env, capture_output=True, text=True)
        n = int(r.stdout.strip() or 0) if r.returncode == 0 and r.stdout.strip().isdigit() else 0
        err = rank_device_error(args.gpus, n, False)
        if err:
            print(err[1], file=sys.stderr)
            return err[0]
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve())] + argv
    return subprocess.run(cmd, env=env).returncode


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="3")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs (cpu_baseline, verified_rows)")
    ap.add_argument("--tile-rows", type=int, default=4)
    ap.add_argument("--spp", type=int, default=0, help="override the config's samples per pixel (e.g. --config 5 --gpus 8 --spp 4000 = "
                                                        "BASELINE configs[4] in full); the workload string says so")
    ap.add_argument("--size", default="", help="override the config's frame, WxH (e.g. --config parity --size 800x600 --spp 2 = the "
                                               "reference's own operating point, main.rs:28 / mod.rs:605-613); the workload string says so")
    ap.add_argument("--rehearse-collectives", action="store_true",
                    help="with --gpus 1: form a ONE-rank process group (RCCL; gloo with --dry-run) and take every N > 1 branch -- gather, "
                         "all-reduces, barriers, gather_ms, verification of the gathered frame -- on a box with a single GPU; measures nothing new")
    ap.add_argument("--preroll-ms", type=float, default=100.0,
                    help="untimed steps before the warm-up steps until the GPU has been busy this long (clocks settle); 0 = none")
    ap.add_argument("--dry-run", action="store_true", help="rehearse the N-rank path on CPU (gloo, pattern renderer); measures nothing")
    ap.add_argument("--verify-against", choices=("auto", "oracle", "self"), default="auto",
                    help="checker of verified_rows: the CPU oracle, rank 0's own single-GPU render of the probe bands (`self`: checks "
                         "partition + gather + de-interleave, not kernel parity), or `auto` = the oracle where one of its rows fits the "
                         "budget, the self-render otherwise (BASELINE configs[4] at 4000 spp)")
    return ap.parse_args(argv)


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)                       # BEFORE torch / the GPU are touched in this process
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    dry = args.dry_run
    err = rank_device_error(world, torch.cuda.device_count(), dry)       # before init_process_group, before the GPU is touched
    if err:
        if rank == 0:
            print(err[1], file=sys.stderr)
        return err[0]
    if not dry:
        torch.cuda.set_device(local_rank)                                 # one rank per GPU
    multi = world > 1 or args.rehearse_collectives            # the code below takes its N > 1 branches
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():
            if dry:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    elif multi:                                                # a one-rank group of its own
        init = f"tcp://127.0.0.1:{free_port()}"
        with stdout_to_stderr():
            if dry:
                dist.init_process_group(backend="gloo", init_method=init, rank=0, world_size=1)
            else:
                dist.init_process_group(backend="nccl", init_method=init, rank=0, world_size=1, device_id=torch.device("cuda", local_rank))

    import weekend_raytracer_wgpu_amd as m

    cfg = dict(CONFIGS[args.config])
    if args.spp > 0 and args.spp != cfg["spp"]:
        cfg["workload"] += f" [spp overridden: {args.spp} instead of {cfg['spp']}]"
        cfg["spp"] = args.spp
    if args.size:
        sw, sh = (int(v) for v in args.size.lower().split("x"))
        if (sw, sh) != (cfg["width"], cfg["height"]):
            cfg["workload"] += f" [frame overridden: {sw}x{sh} instead of {cfg['width']}x{cfg['height']}]"
            cfg["width"], cfg["height"] = sw, sh
    if dry:                                               # small frame: the pattern renderer is numpy
        cfg.update(width=192, height=108, spp=1)
    w, h, spp = cfg["width"], cfg["height"], cfg["spp"]
    device = torch.device("cpu") if dry else torch.device("cuda", local_rank)

    # everything below is queued on ONE non-default stream: renders (through the C ABI), the RCCL gather (which
    # orders itself against torch's current stream) and the de-interleave
    if not dry:
        torch.cuda.set_stream(torch.cuda.Stream(device=local_rank))

    sd = None
    if dry:
        ctx = PatternContext(m)
    else:
        sd = build_scene(m, cfg)
        ctx = m.Context(local_rank)
        ctx.set_scene(sd)                                   # inputs resident in HBM before the timed region
    base = base_params(m, cfg, flags=int(os.environ.get("MIRT_BENCH_FLAGS", "0"), 0))   # e.g. 0x10 strip / 0x20 pool (A/B runs)
    # N > 1: the gather of frame i overlaps the render of frame i+1 (double-buffered parts, async collective);
    # every frame is complete on rank 0 before the timed region ends (flush).  MIRT_BENCH_PIPELINE=0: one frame at a time.
    pipelined = multi and os.environ.get("MIRT_BENCH_PIPELINE", "1") != "0"
    # ... and consecutive frames RENDER on the context's two frame streams (mirt_ctx_frame_stream: different hardware queues): a rank's
    # share of a frame is a short launch whose ramp and tail then overlap its neighbours' (one rank's share of config 3 at N = 8 on one
    # GPU: 89.5 -> 95.4 % of the ideal eighth; profiles/r04_part_overlap.txt).  MIRT_BENCH_FRAMES_IN_FLIGHT=1: one stream.
    in_flight = 2 if pipelined and not dry and os.environ.get("MIRT_BENCH_FRAMES_IN_FLIGHT", "2") != "1" else 1
    def make_frame(pl: bool, fl: int):
        return m.multi_gpu.TiledFrame(ctx, base, rank, world, tile_rows=args.tile_rows, pipelined=pl, device=device,
                                      _rehearse_single_rank=multi and world == 1, frames_in_flight=fl)

    # N > 1: the overlapped schedules (asynchronous gather, two frames in flight on two streams) have only ever met RCCL with ONE rank
    # (--rehearse-collectives), so a schedule must QUALIFY before it is timed: three untimed frames, then rank 0 compares the frame it
    # assembled with its own single-GPU render of the whole frame (the pattern in a dry run), every rank learns the answer, and a schedule
    # that does not reproduce the frame is replaced by the next more conservative one -- two frames in flight -> one stream, pipelined
    # gather -> one frame at a time, blocking gather.  The line says which schedule ran and why (`schedule_check`).
    schedule_check = []
    if multi:
        candidates = [(pipelined, in_flight)]
        for c in ((pipelined, 1), (False, 1)):
            if c not in candidates:
                candidates.append(c)
        fault = os.environ.get("MIRT_BENCH_FAULT_FIRST_SCHEDULE", "0") == "1"        # tests: the first candidate "fails"
        frame = None
        for k, (pl, fl) in enumerate(candidates):
            frame = make_frame(pl, fl)
            with stdout_to_stderr():                       # (the first collective of a communicator prints RCCL's banner)
                for _ in range(3):
                    frame.step()
                frame.flush()
                if not dry:
                    torch.cuda.synchronize()
            ok = True
            if rank == 0:
                if dry:
                    import numpy as np
                    ok = bool(np.array_equal(frame.frame.numpy(), pattern_rows(range(h), w)))
                else:
                    ok = bool(verify_whole_frame_self(m, torch, ctx, base, frame.frame, 0.0)["equal"])
                if fault and k == 0:
                    ok = False
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=torch.device("cpu") if dry else torch.device("cuda"))
            dist.broadcast(flag, src=0)
            ok = bool(int(flag.item()))
            schedule_check.append({"pipelined_gather": bool(pl), "frames_in_flight": fl, "frame_equals_rank0_single_gpu_render": ok})
            if ok:
                break
        pipelined, in_flight = candidates[len(schedule_check) - 1]
    else:
        frame = make_frame(pipelined, in_flight)

    def gather_now():
        """ONE gather of the current part buffers to rank 0 (the rehearsal's one-rank group included)."""
        if world == 1:
            dist.gather(frame.local, list(frame.parts.unbind(0)), dst=0)
            return frame.parts
        return m.multi_gpu.gather_parts(frame.local, rank, world, dst=0, out=frame.parts)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    def barrier():
        if multi:
            dist.barrier()

    if multi:
        # communicator set-up, not a step: RCCL opens its point-to-point channels on first use, so push one
        # gather of the (still empty) buffers through before anything is timed
        with stdout_to_stderr():
            gather_now()
            sync()
            barrier()

    # Clock pre-roll (untimed, before the W warm-up steps): the GPU's clocks take tens of milliseconds of load to settle after an idle
    # gap.  Three warm-up steps of a 23 ms kernel cover that; three of config 2's 0.9 ms kernel (or of a 20 us 2-spp frame) do not --
    # its line read 226 Gsamples/s with the timed region on ramping clocks and 242 after 40 steps.  So: one step to learn its duration
    # (max over ranks: every rank must run the same number of collectives), then as many as fill PREROLL_S.
    preroll_steps = 0
    if not dry and args.preroll_ms > 0:
        sync()
        tp = time.perf_counter()
        frame.step()
        frame.flush()
        sync()
        one = max(time.perf_counter() - tp, 1e-6)
        if multi:
            t = torch.tensor([one], dtype=torch.float64, device=torch.device("cuda"))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            one = float(t.item())
        if multi:                                         # a count every rank agrees on
            preroll_steps = 1 + min(5000, int(args.preroll_ms * 1e-3 / one))
            for _ in range(preroll_steps - 1):
                frame.step()
        else:                                             # by the clock (`one` of a 20 us frame is mostly the synchronize): batches of ~10 ms
            preroll_steps = 1
            batch = max(1, min(256, int(0.01 / one)))
            while time.perf_counter() - tp < args.preroll_ms * 1e-3 and preroll_steps < 100000:
                for _ in range(batch):
                    frame.step()
                sync()
                preroll_steps += batch
        frame.flush()
        sync()
    for _ in range(args.warmup):
        frame.step()
    frame.flush()
    sync()
    ctx.stats()                                         # drain the event pool: the timed region starts clean
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame.step()
    frame.flush()                                       # all K frames assembled on rank 0
    sync()
    barrier()
    elapsed = time.perf_counter() - t0
    st = ctx.stats()                                    # HIP-event time of the K kernels of the timed region
    kernel_ms = st["kernel_ms_total"] / max(1, st["launches"])
    step_ms_this_rank = elapsed / max(1, args.steps) * 1e3
    red_dev = torch.device("cpu") if dry else torch.device("cuda")
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        per_rank = torch.zeros(world, dtype=torch.float64, device=red_dev)     # every rank's kernel time, so that
        per_rank[rank] = kernel_ms                                              # imbalance is visible
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
        kernel_ms_per_rank = [round(float(x), 4) for x in per_rank.tolist()]
        kernel_ms_max = max(kernel_ms_per_rank)
    else:
        kernel_ms_max = kernel_ms
        kernel_ms_per_rank = [round(kernel_ms, 4)]
    partition = ("whole frame" if not multi else
                 f"{args.tile_rows}-row tiles interleaved over {world} ranks + 1 gather per frame"
                 + (" (overlapping the next frame's render)" if pipelined else "")
                 + ("; consecutive frames render on the context's two frame streams (two frames in flight)" if in_flight == 2 else ""))
    # Two frames in flight: a launch's HIP-event duration includes the time it shares the GPU with its neighbour, so the roofline of such a
    # line is priced on WALL time per step instead -- this rank's for `achieved`, the job's (max over ranks) for `achieved_whole_job`; both
    # contain whatever the rank waits for (the gather), i.e. they are lower bounds of what the kernel does.
    roof_ms = step_ms_this_rank if in_flight == 2 else kernel_ms
    roof_ms_job = elapsed / max(1, args.steps) * 1e3 if in_flight == 2 else kernel_ms_max
    total_samples = w * h * spp

    # N > 1: one gather + de-interleave on its own, after the clock has stopped (the timed region overlaps it with the next
    # frame's render, so it cannot be read off the step time): rank 0's wall time of [gather, assemble], ranks released together
    gather_ms = None
    if multi:
        reps, acc = 5, 0.0
        for _ in range(reps):
            barrier()
            sync()
            tg = time.perf_counter()
            parts = gather_now()
            if rank == 0:
                frame._assemble(parts, frame._stream())
            sync()
            acc += time.perf_counter() - tg
        gather_ms = round(acc / reps * 1e3, 4)

    if dry:
        ok = True
        if rank == 0:
            import numpy as np
            got = frame.frame.numpy()
            ok = bool(np.array_equal(got, pattern_rows(range(h), w)))
            # the verification leg of a real N-rank run, with the pattern standing in for the oracle: same function, same fields
            # (--verify-against self: the pattern context renders the probe bands through row_begin / row_end, as rank 0's GPU would)
            def dry_self_rows(first, last):
                import copy
                import numpy as np
                p = copy.copy(base)
                p.tile_rows, p.n_parts, p.part, p.row_begin, p.row_end = 0, 0, 0, first, last
                buf = np.zeros((last - first, w, 4), dtype=np.uint8)
                ctx.render_device(p, buf.ctypes.data, buf.nbytes)
                return buf
            rows = verify_rows(m, None, cfg, got, oracle_rows=lambda first, last: pattern_rows(range(first, last), w),
                               self_rows=dry_self_rows, against=VERIFY_CHECKER[args.verify_against])
            ok = ok and all(r.get("equal") for r in rows)
            print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "frames_verified": ok, "verified_rows": rows, "verified_frame": "gathered on rank 0" if multi else "whole frame",
                              "gather_ms": gather_ms, "launches_rank0": st["launches"], "backend": "gloo" if multi else "none",
                              "schedule_check": schedule_check,
                              "kernel_ms_per_rank": kernel_ms_per_rank,
                              "config": {"workload": f"DRY RUN ({w}x{h} pattern frame, no GPU, nothing measured)", "partition": partition}}),
                  flush=True)
        if multi:
            dist.barrier()
            dist.destroy_process_group()
        return 0 if ok else 4

    # one counting launch (outside the timed region) gives the exact work of this rank's launch; for grid builds
    # (many-sphere scenes) it counts the kernel that actually runs, not the flat scan
    kernel_name = ctx.last_kernel()
    uses_grid = kernel_uses_grid(kernel_name)
    pc = m.multi_gpu.part_params(base, rank, world, args.tile_rows)
    pc.flags |= m.MIRT_FLAG_COUNT_WORK | (m.MIRT_FLAG_COUNT_GRID if uses_grid else 0)
    scratch = torch.empty((frame.max_rows if multi else frame.frame.shape[0], w, 4), dtype=torch.uint8, device=device)
    ctx.render_device(pc, scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    work = ctx.stats()
    counting_kernel = ctx.last_kernel()
    flops = algorithmic_flops(work, mode=cfg["mode"], grid=uses_grid)      # of THIS rank's share
    flops_all = flops
    if multi:
        t = torch.tensor([flops], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        flops_all = float(t.item())

    if rank == 0:
        value = total_samples * args.steps / elapsed / 1e6
        traffic = profiled_traffic(kernel_name, cfg["workload"]) if not multi else None
        achieved_tflops = flops / (roof_ms * 1e-3) / 1e12
        out_bytes = frame.rows * w * 4
        algo_bytes = algorithmic_bytes(sd, cfg["mode"], out_bytes)
        lane_slots = 64 * work["wave_iterations"]
        # MIRT_FLAG_COUNT_WORK can force another schedule than the timed one (the pool's counting build, lane = sample instead
        # of lane = pixel): the flop counts do not depend on the schedule, lane utilisation does
        same_schedule = kernel_schedule(counting_kernel) == kernel_schedule(kernel_name)
        result = {
            "metric": "Msamples/sec (pixels x spp) at 1920x1080, 1000 spp" if args.config == "3" else "Msamples/sec (pixels x spp)",
            "value": round(value, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "preroll_steps": preroll_steps,               # untimed steps before the W warm-up steps (clock pre-roll, --preroll-ms)
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": cfg["workload"], "width": w, "height": h, "spp": spp, "num_bounces": BOUNCES,
                       "mode": cfg["mode"], "partition": partition},
            "roofline": {
                "bound": "valu_fp32",
                "achieved": round(achieved_tflops, 3),
                "peak": PEAK_FP32_VECTOR_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved_tflops / PEAK_FP32_VECTOR_TFLOPS, 4),
                "traffic": traffic[0] if traffic else None,
                "traffic_source": (f"profiles/{traffic[1]} (rocprofv3 PMC, N=1 run of this workload)" if traffic else None),
                "valu_busy_pct_profiled": (round(traffic[2], 1) if traffic and traffic[2] is not None else None),
                "valu_lane_utilization_pct_profiled": (round(traffic[3], 1) if traffic and traffic[3] is not None else None),
                "scope": ("whole frame" if not multi else
                          f"rank 0's share (its tiles: {frame.rows} of {h} rows); achieved_whole_job = all ranks' flops / slowest rank's kernel time" if in_flight != 2 else
                          f"rank 0's share (its tiles: {frame.rows} of {h} rows), two frames in flight: achieved = its flops per step / its WALL time per step "
                          "(launches overlap, so their HIP-event durations -- kernel_ms_* below -- include shared time); achieved_whole_job = all ranks' flops / the job's wall time per step"),
                "frames_in_flight": in_flight,
                "achieved_whole_job": round(flops_all / (roof_ms_job * 1e-3) / 1e12, 3),
                "kernel": kernel_name,
                "kernel_ms_avg": round(kernel_ms, 4),
                "kernel_ms_max_over_ranks": round(kernel_ms_max, 4),
                "kernel_ms_per_rank": kernel_ms_per_rank,
                "algorithmic_gflop_per_launch": round(flops / 1e9, 3),
                "flop_per_sample": round(flops / max(1, work["samples"]), 2),
                "grays_per_s": round(work["rays"] / (kernel_ms * 1e-3) / 1e9, 3),
                "gtests_per_s": round(work["sphere_tests"] / (kernel_ms * 1e-3) / 1e9, 3),
                "counting_kernel": counting_kernel,
                "lane_utilization": round(work["lane_iterations"] / lane_slots, 4) if lane_slots and same_schedule else None,
                "lane_utilization_note": None if same_schedule else "the counting launch ran another schedule than the timed kernel: not reported",
                "grid_walk_lane_utilization": (round(work["grid_cells"] / (64 * work["grid_wave_cells"]), 4)
                                               if work.get("grid_wave_cells") and same_schedule else None),
                "hbm": {"algorithmic_bytes_per_launch": algo_bytes,
                        "achieved_gbs": round(algo_bytes / (kernel_ms * 1e-3) / 1e9, 4),
                        "peak_gbs": PEAK_HBM_GBS},
                "note": "fp32 vector-ALU roofline (no MFMA on this path; HBM traffic is the framebuffer + the scene tables once); "
                        "rocprofv3 summaries under profiles/",
            },
        }
        if not multi and cfg["mode"] == "pt":
            result["fast_math"] = fast_math_line(m, torch, ctx, base, frame.frame, total_samples, flops)
            if args.config == "4":
                result["texel_tiles"] = texel_tiles_line(m, torch, ctx, base, frame.frame, total_samples)
            if args.config == "2" and not args.no_cpu_baseline:   # (not in the profiler passes: since round 4 it runs the timed kernel itself)
                result["steady_state"] = steady_state_line(m, torch, ctx, base, w, h)
        if not multi and cfg["mode"] == "parity":
            result["parity_schedules"] = parity_schedules_line(m, torch, ctx, base, w, h)
        if multi:
            result["schedule_check"] = schedule_check           # the schedules tried before the timed region, most overlapped first; the last one ran
            result["gather_ms"] = gather_ms
            result["gather_note"] = "one gather + de-interleave alone on rank 0 (no render), after the timed region; the timed steps overlap it with the next frame's render"
        cpu_rate = None
        if not multi and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(m, sd, cfg)
            result["gpu_over_cpu"] = round(value / result["cpu_baseline"]["value"], 1)
            cpu_rate = result["cpu_baseline"]["value"] if cfg["mode"] == "pt" else None
            if cfg["mode"] == "pt":                          # the reference's own CPU loop beside it (north_star's last sentence)
                sys.path.insert(0, str(ROOT / "tests"))
                import oracle_binding as ob
                result["cpu_baseline"]["layer_rs"] = layer_rs_baseline(m, ob, build_scene(m, CONFIGS["parity"]), 1920, 1080)
        if not args.no_cpu_baseline:
            # the frame the timed region left on rank 0 -- for N > 1 the one gathered over RCCL -- against oracle rows
            result["verified_rows"] = verify_rows(m, sd, cfg, frame.frame.cpu().numpy(), cpu_msamples_per_s=cpu_rate,
                                                  self_rows=lambda first, last: self_render_rows(m, torch, ctx, base, first, last),
                                                  against=VERIFY_CHECKER[args.verify_against])
            if multi:       # every pixel of the gathered frame against rank 0's own render of the whole frame, when that is a few seconds
                result["verified_whole_frame"] = verify_whole_frame_self(m, torch, ctx, base, frame.frame, sum(kernel_ms_per_rank))
            result["verified_frame"] = "whole frame" if not multi else f"gathered from {world} rank{'s' if world > 1 else ' (rehearsal: a one-rank group)'} on rank 0"
        print(json.dumps(result), flush=True)

    ctx.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
