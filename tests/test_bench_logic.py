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
