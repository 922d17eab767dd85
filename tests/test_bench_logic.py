"""bench.py's pure host logic: the algorithmic-flop formula of SURVEY §8d / DESIGN.md §4.3 and the
lookup of profiled HBM traffic (no GPU, no oracle)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def test_algorithmic_flops_formula():
    st = {"samples": 10, "sphere_tests": 100, "roots": 7, "hits": 5, "scatter": [1, 1, 1, 1, 1], "sky_misses": 9}
    want = 35 * 10 + 23 * 100 + 4 * 7 + 21 * 5 + (45 + 35 + 50 + 54 + 35) + 18 * 9
    assert bench.algorithmic_flops(st) == want
    assert bench.algorithmic_flops(st, hosek=True) == want + (360 - 18) * 9


def test_profiled_traffic_lookup_matches_committed_summaries():
    # the headline kernel under the name mirt_ctx_last_kernel reports (this round's summaries) ...
    new = bench.profiled_traffic("render_pt_pool_kernel<256,112,6,false,false,3,false,false>", bench.CONFIGS["3"]["workload"])
    assert new is not None and new[1].startswith("r04b_c3_pmc") and 8.29e6 <= new[0] <= 2 * 8.3e6     # the newest committed summary OF THIS WORKLOAD wins (not the 2-spp runs of the same kernel family)
    # (the pooled kernel gained an eighth template argument in round 4 -- FLATY, a grid one cell high -- so its profiles were taken again: r04b_*)
    old = bench.profiled_traffic("render_pt_pool_kernel<256,112,6,false,false,3,false>", bench.CONFIGS["3"]["workload"])
    assert old is not None and old[1].startswith("r04_c3_pmc")
    # ... and round 1's summaries under the name bench.py used then
    got = bench.profiled_traffic("render_pt_pool_kernel<256,112,false,false>")
    assert got is not None
    traffic, name, valu_busy, lane_use = got
    assert 0.0 < valu_busy <= 100.0 and 0.0 < lane_use <= 100.0
    d = json.loads((ROOT / "profiles" / name).read_text())
    assert d["derived"]["hbm_bytes_per_launch"] == traffic
    # HBM traffic per launch is the 8.3 MB framebuffer plus the strip dispenser's atomics: well under 2x algorithmic
    assert 8.29e6 <= traffic <= 2 * 8.3e6
    assert bench.profiled_traffic("no_such_kernel") is None


def test_workload_constants_are_the_baseline_config():
    cfg = json.loads((ROOT / "BASELINE.json").read_text())
    assert "1920" in cfg["metric"] and "1000 spp" in cfg["metric"]
    assert (bench.WIDTH, bench.HEIGHT, bench.SPP) == (1920, 1080, 1000)
    assert bench.PEAK_FP32_VECTOR_TFLOPS == 157.3


def test_parity_and_grid_flop_formulas():
    st = {"samples": 10, "sphere_tests": 100, "roots": 7, "hits": 5, "scatter": [0, 3, 0, 0, 0], "sky_misses": 9,
          "lane_iterations": 4, "rays": 6, "grid_cells": 50}
    assert bench.algorithmic_flops(st, mode="parity") == 17 * 4 + 23 * 100 + 4 * 7 + 21 * 5 + 20 * 3
    flat = bench.algorithmic_flops(st)
    assert bench.algorithmic_flops(st, grid=True) == flat + 45 * 6 + 10 * 50


def test_kernel_name_parsing():
    assert bench.kernel_uses_grid("render_pt_pool_kernel<256,112,3,false,false,5,true>")
    assert not bench.kernel_uses_grid("render_pt_pool_kernel<256,112,6,false,false,3,false>")
    assert bench.kernel_uses_grid("render_pt_pool_kernel<1024,152,4,false,false,1,true,true>")             # round 4: + FLATY (a grid one cell high)
    assert bench.kernel_uses_grid("render_pt_pool_kernel<1024,152,4,false,false,1,true,false>")
    assert not bench.kernel_uses_grid("render_pt_pool_kernel<256,112,6,false,false,3,false,false>")
    assert bench.kernel_uses_grid("render_pt_strip_kernel<false,false,true,false>")
    assert not bench.kernel_uses_grid("render_pt_strip_kernel<false,false,false,true>")
    assert not bench.kernel_uses_grid("render_parity_kernel<false,false>") and not bench.kernel_uses_grid("")


def test_every_baseline_config_has_a_bench_workload():
    cfg = json.loads((ROOT / "BASELINE.json").read_text())
    assert len(cfg["configs"]) == 5                       # configs[0] is the CPU-only plumbing case (a parity test)
    assert set(bench.CONFIGS) == {"2", "3", "4", "5", "parity"}
    assert bench.parse_args([]).config == "3" and bench.parse_args([]).gpus == 1
    for c in bench.CONFIGS.values():
        assert {"scene", "width", "height", "spp", "mode", "workload"} <= set(c)


def _bare_env():
    import os
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_bench_launches_its_own_ranks_from_a_bare_shell():
    """`python bench.py --gpus 2` with no WORLD_SIZE must start 2 fresh ranks itself (before touching the GPU),
    reach init_process_group, run the pipelined partition -> gather -> assemble path and print ONE JSON line from
    rank 0.  --dry-run swaps the HIP kernel for a pattern renderer and RCCL for gloo; everything else is the code
    the driver's SCALE run executes."""
    import subprocess
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"],
                       env=_bare_env(), capture_output=True, text=True, timeout=600, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] and d["n_gpus"] == 2 and d["frames_verified"] and d["launches_rank0"] == 3
    assert len(d["kernel_ms_per_rank"]) == 2 and "2 ranks + 1 gather per frame" in d["config"]["partition"]
    # the verification leg of a real N-rank line: bands of rows of the frame GATHERED on rank 0 against the checker (here the
    # pattern), and the stand-alone gather + assemble time
    assert d["verified_frame"] == "gathered on rank 0"
    assert len(d["verified_rows"]) >= 3 and all(r["equal"] and r["rows_equal"] == r["rows"][1] - r["rows"][0] for r in d["verified_rows"])
    assert d["gather_ms"] is not None and d["gather_ms"] > 0.0


def test_an_overlapped_schedule_must_reproduce_the_frame_before_it_is_timed():
    """N > 1: before the timed region every schedule runs three untimed frames and rank 0 compares what it assembled with its own
    render of the whole frame (`schedule_check`); a schedule that fails is replaced by the next more conservative one -- here the first
    candidate is made to fail (MIRT_BENCH_FAULT_FIRST_SCHEDULE=1): the pipelined gather gives way to one frame at a time, every rank
    follows (the decision is broadcast), and the line verifies."""
    import subprocess
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"],
                       env=dict(_bare_env(), MIRT_BENCH_FAULT_FIRST_SCHEDULE="1"), capture_output=True, text=True, timeout=600, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    sc = d["schedule_check"]
    assert [c["frame_equals_rank0_single_gpu_render"] for c in sc] == [False, True]
    assert sc[0]["pipelined_gather"] and not sc[1]["pipelined_gather"] and sc[1]["frames_in_flight"] == 1
    assert d["frames_verified"] and "overlapping the next frame's render" not in d["config"]["partition"]
    # without the fault the first (most overlapped) schedule qualifies
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"],
                       env=_bare_env(), capture_output=True, text=True, timeout=600, cwd="/tmp")
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert len(d["schedule_check"]) == 1 and d["schedule_check"][0]["frame_equals_rank0_single_gpu_render"] and d["schedule_check"][0]["pipelined_gather"]
    assert "overlapping the next frame's render" in d["config"]["partition"]


def test_bench_rejects_a_world_size_mismatch():
    import os
    import subprocess
    env = dict(_bare_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True,
                       timeout=300, cwd="/tmp")
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


def test_one_rank_per_gpu_is_checked_before_the_process_group():
    """RCCL refuses two ranks on one device inside init_process_group ("Duplicate GPU detected"): bench.py says so itself, in
    one line, with exit code 2 -- in the launcher and again in every rank -- and 3 when there is no GPU at all."""
    assert bench.rank_device_error(2, 1, False)[0] == 2 and "2 GPUs, 1 visible" in bench.rank_device_error(2, 1, False)[1]
    assert bench.rank_device_error(8, 4, False)[0] == 2
    assert bench.rank_device_error(1, 0, False)[0] == 3 and bench.rank_device_error(2, 0, False)[0] == 3
    assert bench.rank_device_error(1, 1, False) is None and bench.rank_device_error(8, 8, False) is None
    assert bench.rank_device_error(2, 0, True) is None                     # --dry-run needs no GPU
    import subprocess
    # a bare `--gpus 2` on this GPU-less machine: refused by the launcher, nothing is spawned
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], env=_bare_env(), capture_output=True, text=True, timeout=300, cwd="/tmp")
    import torch
    if torch.cuda.device_count() == 0:
        assert r.returncode == 3 and "no GPU visible" in r.stderr and "Traceback" not in r.stderr
    # ... and as a rank of a 2-rank job (what the driver's torch.distributed.run starts), before init_process_group
    env = dict(_bare_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode in (2, 3) and "Traceback" not in r.stderr and r.stdout.strip() == ""


def test_algorithmic_bytes_count_only_what_a_launch_reads():
    """config 3 = framebuffer + camera + 3 spheres + 5 materials = 8 294 752 B -- not the texel table, which that
    scene never reads; config 4 adds the 1024 x 512 earth texture once; parity mode two texels."""
    import weekend_raytracer_wgpu_amd as m

    def sd_of(name, w=1920, h=1080):
        return bench.build_scene(m, dict(bench.CONFIGS[name]))
    fb = 1920 * 1080 * 4
    sd3 = sd_of("3")
    assert bench.algorithmic_bytes(sd3, "pt", fb) == 8294752
    sd4 = sd_of("4")
    assert bench.algorithmic_bytes(sd4, "pt", fb) == fb + 96 + 32 * len(sd4.spheres) + 32 * len(sd4.materials) + 12 * 1024 * 512
    sdp = sd_of("parity")
    assert bench.algorithmic_bytes(sdp, "parity", fb) == fb + 96 + 32 * 6 + 32 * 5 + 24


def test_kernel_schedule_names():
    assert bench.kernel_schedule("render_pt_strip_kernel<false,false,false,true>") == "strip/pixel"
    assert bench.kernel_schedule("render_pt_strip_kernel<true,false,false,false>") == "strip/sample"
    assert bench.kernel_schedule("render_pt_stream_kernel<false>") == "strip/pixel-stream" and not bench.kernel_uses_grid("render_pt_stream_kernel<false>")
    assert bench.kernel_schedule("render_pt_pool_kernel<256,112,6,false,false,3,false>") == bench.kernel_schedule("render_pt_pool_kernel<256,112,1,true,false,5,false>") == "pool"
    assert bench.kernel_schedule("render_parity_kernel<false,true>") == "parity/pixel" != bench.kernel_schedule("render_parity_kernel<true,false>")


def test_the_rehearsal_of_the_n_rank_branches_runs_without_a_gpu():
    """`bench.py --rehearse-collectives --dry-run`: a one-rank gloo group through every N > 1 branch (the GPU suite runs the same
    over RCCL)."""
    import subprocess
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--dry-run", "--rehearse-collectives", "--steps", "2", "--warmup", "1"],
                       env=_bare_env(), capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["frames_verified"] and d["verified_frame"] == "gathered on rank 0" and d["backend"] == "gloo" and d["gather_ms"] > 0.0


def test_verify_rows_falls_back_to_rank0s_own_render_when_the_oracle_is_out_of_reach(monkeypatch):
    """BASELINE configs[4] -- 3840 x 2160 x 4000 spp, the one job DEFINED on 8 GPUs -- costs the oracle ten minutes per row.  Its
    line must not say "skipped": verify_rows then compares the probe bands of the gathered frame with rank 0's own single-GPU render
    of those rows (partition + gather + de-interleave at full size) and names that checker in every entry."""
    import numpy as np
    cfg = dict(bench.CONFIGS["5"], spp=4000)
    w, h = cfg["width"], cfg["height"]
    frame = np.zeros((h, 8, 4), np.uint8)                       # 8 columns stand in for 3840: verify_rows only slices rows
    frame[..., 0] = (np.arange(h) % 251)[:, None]
    calls = []

    def self_rows(first, last):
        calls.append((first, last))
        return frame[first:last].copy()

    class _NoOracle:                                             # the oracle must not be asked for a row
        @staticmethod
        def render(*a, **k):
            raise AssertionError("the oracle was called for a workload it cannot afford")
    import sys as _sys
    monkeypatch.setitem(_sys.modules, "oracle_binding", _NoOracle)
    monkeypatch.setattr(bench, "host_cores", lambda: 256)       # the GPU box's host
    # cpu_msamples_per_s as the N = 1 line measured it for RTIOW: ~6 Msamples/s on all cores -> ~650 s per row and thread
    rows = bench.verify_rows(None, None, cfg, frame, cpu_msamples_per_s=6.0, self_rows=self_rows, against="auto")
    assert len(rows) == len(bench.VERIFY_ROWS[2160]) == len(calls)
    assert all(r["against"] == "rank0_single_gpu_render" and r["equal"] and r["spp"] == 4000 and "skipped" not in r for r in rows)
    assert all("oracle row" in r["note"] for r in rows)
    # a wrong row in the gathered frame is caught
    bad = frame.copy()
    bad[1081, 3, 1] ^= 1
    rows = bench.verify_rows(None, None, cfg, bad, cpu_msamples_per_s=6.0, self_rows=self_rows, against="auto")
    assert [r["equal"] for r in rows].count(False) == 1 and [r for r in rows if not r["equal"]][0]["rows_equal"] == 15
    # forced oracle on that workload still reports the skip (and says which checker skipped); forced self never needs the rate
    rows = bench.verify_rows(None, None, cfg, frame, cpu_msamples_per_s=6.0, self_rows=self_rows, against="oracle")
    assert rows == [dict(rows[0])] and rows[0]["against"] == "oracle" and "skipped" in rows[0]
    rows = bench.verify_rows(None, None, cfg, frame, self_rows=self_rows, against="rank0_single_gpu_render")
    assert all(r["against"] == "rank0_single_gpu_render" and r["equal"] for r in rows)
    # an affordable workload keeps the oracle (here: the dry run's pattern stands in for it)
    cfg3 = dict(bench.CONFIGS["3"])
    f3 = bench.pattern_rows(range(1080), 16)
    rows = bench.verify_rows(None, None, cfg3, f3, oracle_rows=lambda a, b: bench.pattern_rows(range(a, b), 16), self_rows=self_rows, against="auto")
    assert len(rows) == 5 and all(r["against"] == "oracle" and r["equal"] for r in rows)


def test_the_dry_run_takes_the_self_render_branch_of_the_verification():
    """`bench.py --gpus 2 --dry-run --verify-against self`: the gathered frame's probe bands against bands rank 0 renders itself through
    row_begin / row_end (the pattern context here, the GPU in a real run)."""
    import subprocess
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1", "--verify-against", "self"],
                       env=_bare_env(), capture_output=True, text=True, timeout=600, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["frames_verified"] and len(d["verified_rows"]) >= 3
    assert all(v["against"] == "rank0_single_gpu_render" and v["equal"] for v in d["verified_rows"])
    # ... and the one-rank rehearsal too
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--dry-run", "--rehearse-collectives", "--steps", "1", "--warmup", "0", "--verify-against", "self"],
                       env=_bare_env(), capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert all(v["against"] == "rank0_single_gpu_render" and v["equal"] for v in d["verified_rows"])


def test_the_complete_frame_test_never_compares_nothing():
    """tests/test_gpu_baseline_configs.py::test_complete_frames_...: the whole frame when the oracle's pass fits the budget, else as
    many 16-row bands as do -- at least one -- spread over the frame."""
    from helpers import full_frame_bands
    assert full_frame_bands(1080, 0.9, 256, 120.0) == [(0, 1080)]                       # the GPU box: 20 s for config 3
    slow = full_frame_bands(1080, 20.0, 8, 120.0)                                       # this container
    assert len(slow) == 1 and slow[0][1] - slow[0][0] == 16 and slow[0][0] < 540 < slow[0][1]
    hopeless = full_frame_bands(1080, 5000.0, 2, 120.0)
    assert len(hopeless) == 1 and hopeless[0][1] - hopeless[0][0] == 16                 # never zero
    mid = full_frame_bands(2160, 5.0, 16, 120.0)
    assert 2 <= len(mid) < 2160 // 16 and mid[0] == (0, 16) and mid[-1] == (2144, 2160)
    assert all(b - a == 16 for a, b in mid) and all(mid[i][1] <= mid[i + 1][0] for i in range(len(mid) - 1))
    assert full_frame_bands(8, 1000.0, 4, 1.0) == [(0, 8)]                              # a frame smaller than a band
