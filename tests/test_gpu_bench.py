"""bench.py end to end on the GPU (short runs): the ONE JSON line must carry the driver's contract fields, the roofline
object, the kernel's own name and -- with the CPU legs on -- cpu_baseline and verified_rows that are all true."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu

CONTRACT = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline"}


def _bench(*args):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]          # stdout is the ONE JSON line, nothing else (no RCCL banner)
    return json.loads(lines[0])


def test_default_line_has_the_contract_fields_and_a_verified_frame():
    d = _bench("--steps", "2", "--warmup", "1")
    assert CONTRACT <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "Msamples/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and "1000 spp" in d["metric"]
    assert d["config"]["width"] == 1920 and d["config"]["spp"] == 1000 and "configs[2]" in d["config"]["workload"]
    r = d["roofline"]
    assert r["kernel"].startswith("render_pt_pool_kernel<") and r["peak"] == 157.3 and 0.0 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(d["value"] - 1920 * 1080 * 1000 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    assert r["kernel_ms_avg"] <= d["ms_per_step"] * 1.02                       # the kernel is the step
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    assert {"faithful_1_thread", "clean_1_thread", "faithful_all_cores", "clean_all_cores"} <= set(d["cpu_baseline"]["layer_rs"])
    assert len(d["verified_rows"]) >= 4 and all(v["equal"] for v in d["verified_rows"])
    assert d["fast_math"]["kernel"].startswith("fast_build::") and d["fast_math"]["within_1_pct"] >= 99.9


@pytest.mark.parametrize("config,kernel", [("2", "render_pt_stream_kernel<false>"), ("parity", "render_parity_kernel<"), ("5", "render_pt_pool_kernel<1024,")])
def test_other_configs_emit_the_same_shape(config, kernel):
    d = _bench("--config", config, "--steps", "1", "--warmup", "1", "--no-cpu-baseline")
    assert CONTRACT <= set(d) and d["roofline"]["kernel"].startswith(kernel)
    assert "cpu_baseline" not in d and "verified_rows" not in d
    if config == "5":
        assert d["roofline"]["grid_walk_lane_utilization"] > 0.5


def test_config4_line_reports_the_texel_tile_build():
    d = _bench("--config", "4", "--steps", "1", "--warmup", "1", "--no-cpu-baseline")
    assert CONTRACT <= set(d) and d["roofline"]["kernel"].startswith("render_pt_pool_kernel<256,")
    t = d["texel_tiles"]
    assert t["kernel"].startswith("render_pt_pool_tile_kernel<256,") and t["frame_identical_to_default_build"] is True
    assert t["tile_hit_rate_pct"]["camera_ray_hits"] > 50.0 > t["tile_hit_rate_pct"]["later_bounces"]


def test_the_n_rank_branches_run_over_rccl_on_one_gpu():
    """`--rehearse-collectives`: a ONE-rank RCCL group, and every branch an N > 1 run takes -- tiled frame with the pipelined async
    gather, the all-reduces of elapsed time / kernel times / flops, gather_ms, verification of the GATHERED frame against oracle rows,
    barrier and teardown.  The driver's 2 / 4 / 8-GPU runs execute exactly this code with more ranks."""
    d = _bench("--config", "2", "--rehearse-collectives", "--steps", "3", "--warmup", "1")
    assert CONTRACT <= set(d) and d["n_gpus"] == 1
    assert "gather per frame" in d["config"]["partition"] and "overlapping" in d["config"]["partition"]
    assert d["gather_ms"] > 0.0 and d["verified_frame"].startswith("gathered from 1 rank")
    assert len(d["verified_rows"]) >= 4 and all(v["equal"] for v in d["verified_rows"])
    assert d["roofline"]["scope"].startswith("rank 0's share") and d["roofline"]["achieved_whole_job"] > 0.0
    assert "cpu_baseline" not in d and len(d["roofline"]["kernel_ms_per_rank"]) == 1
    assert all(v["against"] == "oracle" for v in d["verified_rows"])
    assert d["verified_whole_frame"]["equal"] is True and d["verified_whole_frame"]["rows"] == [0, 1080]
    # the schedule qualified before it was timed: two frames in flight + the pipelined gather reproduce rank 0's own render
    assert d["schedule_check"] == [{"pipelined_gather": True, "frames_in_flight": 2, "frame_equals_rank0_single_gpu_render": True}]
    assert d["roofline"]["frames_in_flight"] == 2


def test_the_8_gpu_headline_job_verifies_its_gathered_frame_without_the_oracle():
    """BASELINE configs[4] exactly as named -- 3840 x 2160 x 4000 spp -- through the N-rank branches (one-rank RCCL group).  The line
    never says "skipped": the probe bands of the gathered frame are held against rank 0's own single-GPU render of those rows
    (`--verify-against self`; with `auto` the oracle is taken only where one of its rows fits the budget -- minutes per row here), and
    the complete gathered frame against rank 0's render of the whole frame."""
    d = _bench("--config", "5", "--spp", "4000", "--rehearse-collectives", "--steps", "1", "--warmup", "0", "--preroll-ms", "0", "--verify-against", "self")
    assert d["config"]["spp"] == 4000 and d["config"]["width"] == 3840
    rows = d["verified_rows"]
    assert len(rows) >= 3 and all("skipped" not in v and v["equal"] and v["against"] == "rank0_single_gpu_render" and v["spp"] == 4000 for v in rows)
    assert d["verified_whole_frame"]["equal"] is True and d["verified_whole_frame"]["rows"] == [0, 2160]


def test_cpu_baseline_is_a_best_of_n():
    d = _bench("--config", "2", "--steps", "2", "--warmup", "1")
    cb = d["cpu_baseline"]
    assert cb["repeats"] >= 3 and cb["min"] <= cb["median"] <= cb["value"] and cb["cores"] >= 2
    assert all(v["against"] == "oracle" and v["equal"] for v in d["verified_rows"])
