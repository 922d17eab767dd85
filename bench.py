#!/usr/bin/env python3
"""bench.py — Msamples/s of the per-pixel ray-trace path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by `python -m torch.distributed.run --nproc-per-node N ...` (one rank per GPU,
  RCCL); RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment.

Workload (BASELINE.json configs[2], the configuration `metric` is quoted on): path-traced mode,
three-sphere diffuse+metal+dielectric scene (reference src/main.rs:539-541), 1920x1080, 1000 spp,
8 bounces, synthetic scene tables already resident in HBM.  One "step" = one full frame:
every rank renders its row tiles, ONE gather brings them to rank 0, rank 0 assembles the RGBA8
frame.  value = pixels x spp x K / wall seconds / 1e6 (whole job, all ranks; total work is fixed
as N grows -> "strong" scaling).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     fp32 vector-ALU roofline of the render kernel (north_star: "scalar-ray fp32 FMA,
               no MFMA"): achieved = ALGORITHMIC flops per launch (work counters of one counting
               launch x the per-unit figures of SURVEY §8d / DESIGN.md) / the kernel's average
               duration measured with HIP events on the launch stream inside the timed region.
  cpu_baseline the CPU oracle (a port: the reference is Rust and cannot be built here) timed on
               the host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path[:0] = [str(ROOT)]

WIDTH, HEIGHT, SPP, BOUNCES = 1920, 1080, 1000, 8
PEAK_FP32_VECTOR_TFLOPS = 157.3     # MI355X_MICROARCH.md "Peak FP32 (vector)": 256 CU x 256 flop/clk x 2.4 GHz
PEAK_HBM_GBS = 8000.0

# algorithmic flops per unit of work (DESIGN.md "Algorithmic work")
FLOPS = {
    "sample": 17 + 12 + 6,          # ray generation + thin-lens offset + throughput*colour & fixed-point convert
    "test": 23,                     # ray-sphere test up to the discriminant
    "root": 4,                      # sqrt, negate, add/sub, scale
    "hit": 21,                      # p, n, u/v
    "scatter": [45, 35, 50, 54, 35],  # lambertian, metal, dielectric, checkerboard(+9 over lambertian), missing
    "sky_gradient": 18,
    "sky_hosek": 360,
}


def algorithmic_flops(st: dict, hosek: bool = False) -> float:
    f = FLOPS["sample"] * st["samples"] + FLOPS["test"] * st["sphere_tests"] + FLOPS["root"] * st["roots"]
    f += FLOPS["hit"] * st["hits"] + sum(w * n for w, n in zip(FLOPS["scatter"], st["scatter"]))
    f += (FLOPS["sky_hosek"] if hosek else FLOPS["sky_gradient"]) * st["sky_misses"]
    return float(f)


def profiled_traffic(kernel: str):
    """HBM bytes per launch of the render kernel from the newest committed rocprofv3 PMC summary
    (profiles/*_pmc_summary.json, written by tools/summarize_profile.py: separate --pmc passes,
    FETCH_SIZE x2 gfx950 correction).  None if no summary for this kernel exists."""
    best = None
    for f in sorted((ROOT / "profiles").glob("*_pmc_summary.json")):
        try:
            d = json.loads(f.read_text())
        except (OSError, ValueError):
            continue
        if d.get("kernel_id") == kernel and "hbm_bytes_per_launch" in d.get("derived", {}):
            best = (d["derived"]["hbm_bytes_per_launch"], f.name, d["derived"].get("valu_issue_busy_pct"),
                    d["derived"].get("valu_lane_utilization_pct"))
    return best


def build_scene(m):
    scene, cam = m.scenes.three_spheres()
    mats, texels = m.flatten_materials(scene.materials)
    gcam = m.GpuCamera.new(cam, (WIDTH, HEIGHT))
    return m.SceneData(gcam.c, [s.to_c() for s in scene.spheres], mats, texels)


def cpu_baseline(m, sd, target_seconds: float = 15.0) -> dict:
    """Time the CPU oracle on a bounded sample of the SAME workload (same frame, fewer spp)."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_binding as ob     # checker / baseline only

    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    spp, secs = 4, 0.0
    for _ in range(4):                                   # grow the sample until it is worth ~target_seconds of CPU work
        p = m.make_params(WIDTH, HEIGHT, spp, mode=m.MIRT_MODE_PT, num_bounces=BOUNCES)
        ob.render(sd, p, n_threads=cores)
        secs = ob.stats()["kernel_ms"] / 1e3
        if secs >= 0.6 * target_seconds or spp >= SPP:
            break
        spp = int(max(spp + 1, min(SPP, round(spp * target_seconds / max(secs, 1e-3)))))
    return {"value": round(WIDTH * HEIGHT * spp / secs / 1e6, 3), "unit": "Msamples/s", "cores": cores,
            "kind": "port",
            "sample": f"same scene and frame ({WIDTH}x{HEIGHT}, {BOUNCES} bounces) at {spp} spp instead of {SPP} "
                      f"({secs:.1f} s; rate is spp-independent); oracle = C restatement, OpenMP over rows; "
                      f"the Rust reference cannot be built (no toolchain)"}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tile-rows", type=int, default=4)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the render path has no CPU fallback", file=sys.stderr)
        return 3
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import weekend_raytracer_wgpu_amd as m

    # everything below is queued on ONE non-default stream: renders (through the C ABI), the RCCL gather (which
    # orders itself against torch's current stream) and the de-interleave
    torch.cuda.set_stream(torch.cuda.Stream(device=local_rank))

    sd = build_scene(m)
    ctx = m.Context(local_rank)
    ctx.set_scene(sd)                                   # inputs resident in HBM before the timed region
    base = m.make_params(WIDTH, HEIGHT, SPP, mode=m.MIRT_MODE_PT, num_bounces=BOUNCES,
                         flags=int(os.environ.get("MIRT_BENCH_FLAGS", "0"), 0))      # e.g. 0x10 strip / 0x20 pool (A/B runs)
    # N > 1: the gather of frame i overlaps the render of frame i+1 (double-buffered parts, async collective);
    # every frame is complete on rank 0 before the timed region ends (flush).  MIRT_BENCH_PIPELINE=0: one frame at a time.
    pipelined = world > 1 and os.environ.get("MIRT_BENCH_PIPELINE", "1") != "0"
    frame = m.multi_gpu.TiledFrame(ctx, base, rank, world, tile_rows=args.tile_rows, pipelined=pipelined)

    def barrier():
        if world > 1:
            dist.barrier()

    if world > 1:
        # communicator set-up, not a step: RCCL opens its point-to-point channels on first use, so push one
        # gather of the (still empty) buffers through before anything is timed
        m.multi_gpu.gather_parts(frame.local, rank, world, dst=0, out=frame.parts)
        torch.cuda.synchronize()
        barrier()

    for _ in range(args.warmup):
        frame.step()
    frame.flush()
    torch.cuda.synchronize()
    ctx.stats()                                         # drain the event pool: the timed region starts clean
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame.step()
    frame.flush()                                       # all K frames assembled on rank 0
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    st = ctx.stats()                                    # HIP-event time of the K kernels of the timed region
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        k = torch.tensor([st["kernel_ms_total"] / max(1, st["launches"])], dtype=torch.float64, device="cuda")
        kmax = k.clone()
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
        kernel_ms_max = float(kmax.item())
        per_rank = torch.zeros(world, dtype=torch.float64, device="cuda")     # every rank's kernel time, so that
        per_rank[rank] = k[0]                                                  # imbalance is visible
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
        kernel_ms_per_rank = [round(float(x), 4) for x in per_rank.tolist()]
    else:
        kernel_ms_max = st["kernel_ms_total"] / max(1, st["launches"])
        kernel_ms_per_rank = [round(kernel_ms_max, 4)]
    kernel_ms = st["kernel_ms_total"] / max(1, st["launches"])

    # one counting launch (outside the timed region) gives the exact work of this rank's launch
    pc = m.multi_gpu.part_params(base, rank, world, args.tile_rows)
    pc.flags |= m.MIRT_FLAG_COUNT_WORK
    ctx.render_device(pc, frame.local.data_ptr() if world > 1 else frame.frame.data_ptr(),
                      frame.rows * WIDTH * 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    work = ctx.stats()

    result = None
    if rank == 0:
        total_samples = WIDTH * HEIGHT * SPP
        value = total_samples * args.steps / elapsed / 1e6
        flops = algorithmic_flops(work)
        kernel_name = "render_pt_pool_kernel<256,112,false,false>"     # default schedule for spp >= 48 (mirt_api.hip)
        if base.flags & m.MIRT_FLAG_KERNEL_STRIP:
            kernel_name = "render_pt_strip_kernel<false,false>"
        traffic = profiled_traffic(kernel_name) if world == 1 else None
        achieved_tflops = flops / (kernel_ms * 1e-3) / 1e12
        out_bytes = frame.rows * WIDTH * 4
        in_bytes = 32 * len(sd.spheres) + 32 * len(sd.materials) + 96
        result = {
            "metric": "Msamples/sec (pixels x spp) at 1920x1080, 1000 spp",
            "value": round(value, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: path-traced 3-sphere scene, checker-diffuse ground + glass + metal "
                            "(src/main.rs:539-541), 1920x1080, 1000 spp, 8 bounces, gradient sky, seed 0",
                "width": WIDTH, "height": HEIGHT, "spp": SPP, "num_bounces": BOUNCES, "mode": "pt",
                "partition": "whole frame" if world == 1 else f"{args.tile_rows}-row tiles interleaved over {world} ranks + 1 gather per frame"
                             + (" (overlapping the next frame's render)" if pipelined else ""),
            },
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
                "kernel": kernel_name,
                "kernel_ms_avg": round(kernel_ms, 4),
                "kernel_ms_max_over_ranks": round(kernel_ms_max, 4),
                "kernel_ms_per_rank": kernel_ms_per_rank,
                "algorithmic_gflop_per_launch": round(flops / 1e9, 3),
                "flop_per_sample": round(flops / work["samples"], 2),
                "grays_per_s": round(work["rays"] / (kernel_ms * 1e-3) / 1e9, 3),
                "gtests_per_s": round(work["sphere_tests"] / (kernel_ms * 1e-3) / 1e9, 3),
                "lane_utilization": round(work["lane_iterations"] / max(1, 64 * work["wave_iterations"]), 4),
                "hbm": {"algorithmic_bytes_per_launch": out_bytes + in_bytes,
                        "achieved_gbs": round((out_bytes + in_bytes) / (kernel_ms * 1e-3) / 1e9, 4),
                        "peak_gbs": PEAK_HBM_GBS},
                "note": "fp32 vector-ALU roofline (no MFMA on this path; HBM traffic is ~8 MB per launch); "
                        "rocprofv3 summaries under profiles/",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(m, sd)
            result["gpu_over_cpu"] = round(value / result["cpu_baseline"]["value"], 1)
        print(json.dumps(result), flush=True)

    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
