#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh run (gpurun_out/prof_<tag>/) into committed artefacts:

    profiles/<tag>_bench.json         the plain (un-profiled) bench line of the same command
    profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
    profiles/<tag>_pmc_summary.json   per-launch averages of every collected counter for the render
                                      kernel + derived metrics (VALU issue busy %, lane utilisation,
                                      clock, HBM bytes with the gfx950 FETCH_SIZE x2 correction)
    profiles/<tag>_resource_usage.txt the compiler's register / LDS / spill report for that kernel
                                      (from `make -C weekend-raytracer-wgpu_amd/csrc asm`, if present)

The render kernel is taken from the bench line (`roofline.kernel`, i.e. mirt_ctx_last_kernel).
Usage: python tools/summarize_profile.py TAG
"""
from __future__ import annotations

import collections
import csv
import glob
import json
import re
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
N_CU, SIMD_PER_CU = 256, 4


def rocprof_spelling(kernel_id: str) -> str:
    """'render_pt_pool_kernel<256,112,6,false,false,3,false>' -> the substring rocprofv3 prints
    ('render_pt_pool_kernel<256u, 112u, 6u, false, false, 3u, false>')."""
    name, targs = kernel_id[:kernel_id.index("<")], kernel_id[kernel_id.index("<") + 1:-1].split(",")
    if "::" not in name:
        name = "exact_build::" + name            # the default build's namespace (the opt-in one reports "fast_build::...")
    return name + "<" + ", ".join(a + "u" if a.isdigit() else a for a in targs) + ">"


def mangled_fragment(kernel_id: str) -> str:
    name, targs = kernel_id[:kernel_id.index("<")], kernel_id[kernel_id.index("<") + 1:-1].split(",")
    name = name.split("::")[-1]
    enc = "".join(f"Lj{a}E" if a.isdigit() else f"Lb{1 if a == 'true' else 0}E" for a in targs)
    return f"{name}I{enc}E"


def main() -> int:
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = ROOT / "gpurun_out" / f"prof_{tag}"
    dst = ROOT / "profiles"
    dst.mkdir(exist_ok=True)
    bench_line = [ln for ln in (src / "bench.json").read_text().splitlines() if ln.startswith("{")][-1]
    bench = json.loads(bench_line)
    kernel_id = bench["roofline"]["kernel"]
    needle = rocprof_spelling(kernel_id)
    (dst / f"{tag}_bench.json").write_text(bench_line + "\n")
    import os
    # gpurun MERGES a visit's files into gpurun_out/: a workload profiled twice has both visits' files side by side -- the NEWEST one per
    # pass is the build being summarised
    stats = sorted(glob.glob(str(src / "trace" / "**" / "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    if not stats:
        print("no kernel_stats.csv under", src)
        return 1
    shutil.copy(stats[0], dst / f"{tag}_kernel_stats.csv")
    kernel_ns, calls = None, 0
    for row in csv.DictReader(open(stats[0])):
        if needle in row["Name"]:
            kernel_ns = float(row["AverageNs"])
            calls = int(row["Calls"])
    if kernel_ns is None:
        print("kernel", needle, "not in", stats[0])
        return 1
    counters: dict[str, list[float]] = collections.defaultdict(list)
    meta = {}
    newest: dict[str, str] = {}
    for f in glob.glob(str(src / "pmc*" / "**" / "*_counter_collection.csv"), recursive=True):
        pass_dir = Path(f).relative_to(src).parts[0]
        if pass_dir not in newest or os.path.getmtime(f) > os.path.getmtime(newest[pass_dir]):
            newest[pass_dir] = f
    for f in sorted(newest.values()):
        for row in csv.DictReader(open(f)):
            if needle in row["Kernel_Name"]:
                counters[row["Counter_Name"]].append(float(row["Counter_Value"]))
                meta = {"grid_size": int(row["Grid_Size"]), "workgroup_size": int(row["Workgroup_Size"])}
    avg = {k: sum(v) / len(v) for k, v in counters.items()}
    d: dict[str, float] = {}
    secs = kernel_ns * 1e-9
    if "GRBM_GUI_ACTIVE" in avg:
        cycles = avg["GRBM_GUI_ACTIVE"] / 8.0                   # rocprofv3 sums the 8 XCDs
        d["clock_ghz"] = cycles / secs / 1e9                     # NB: profiled pass; kernel time from the trace pass
        simd_cycles = cycles * N_CU * SIMD_PER_CU
        if "SQ_INSTS_VALU" in avg:
            # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles at full rate
            d["valu_issue_busy_pct"] = 100.0 * avg["SQ_INSTS_VALU"] * 2.0 / simd_cycles
        if "SQ_ACTIVE_INST_VALU" in avg:
            # SQ_ACTIVE_INST_* count quad-cycles per wave; two waves' VALU ops overlap on one SIMD
            d["valu_active_pct_of_pipe"] = 100.0 * avg["SQ_ACTIVE_INST_VALU"] * 4.0 / (2.0 * simd_cycles)
    if "SQ_THREAD_CYCLES_VALU" in avg and "SQ_ACTIVE_INST_VALU" in avg:
        d["valu_lane_utilization_pct"] = 100.0 * avg["SQ_THREAD_CYCLES_VALU"] / (avg["SQ_ACTIVE_INST_VALU"] * 64.0)
    if "SQ_WAVE_CYCLES" in avg:
        for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if k in avg:
                d[k.lower() + "_share_of_wave_cycles_pct"] = 100.0 * avg[k] / avg["SQ_WAVE_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in avg and avg.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_share_pct"] = 100.0 * avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"]
    samples = bench["config"]["width"] * bench["config"]["height"] * bench["config"]["spp"]
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
        if k in avg:
            d[k.lower() + "_per_wave_sample"] = avg[k] / (samples / 64.0)     # wave-instructions per 64 samples
    if "FETCH_SIZE" in avg:
        d["hbm_read_bytes"] = avg["FETCH_SIZE"] * 1024.0 * 2.0   # gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads
    if "WRITE_SIZE" in avg:
        d["hbm_write_bytes"] = avg["WRITE_SIZE"] * 1024.0
    if "hbm_read_bytes" in d and "hbm_write_bytes" in d:
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
        d["hbm_gbs"] = d["hbm_bytes_per_launch"] / secs / 1e9
    if "TCC_HIT_sum" in avg and "TCC_MISS_sum" in avg:
        d["l2_hit_rate_pct"] = 100.0 * avg["TCC_HIT_sum"] / max(1.0, avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])
    out = {"tag": tag, "kernel": needle, "kernel_id": kernel_id, "workload": bench["config"]["workload"],
           "kernel_avg_ms_trace_pass": kernel_ns / 1e6, "kernel_avg_ms_hip_events_unprofiled": bench["roofline"]["kernel_ms_avg"],
           "calls_in_trace": calls, "dispatch": meta, "counters_per_launch_avg": avg, "derived": d,
           "roofline_from_bench_line": {k: bench["roofline"][k] for k in ("achieved", "peak", "frac", "algorithmic_gflop_per_launch", "flop_per_sample")},
           "notes": ["each --pmc group was collected in its own run (tools/profile_bench.sh)",
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of coalesced reads); WRITE_SIZE taken as is",
                     "kernel time is from the un-counted --kernel-trace --stats pass",
                     "registers / LDS / spills: see the _resource_usage.txt beside this file (rocprofv3's dispatch fields are not reliable here)"]}
    (dst / f"{tag}_pmc_summary.json").write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
    ru = ROOT / "weekend-raytracer-wgpu_amd" / "csrc" / "build" / "resource_usage.txt"
    if ru.exists():
        frag = mangled_fragment(kernel_id)
        lines, keep = [], False
        for ln in ru.read_text().splitlines():
            if "Function Name:" in ln:
                keep = frag in ln
            if keep:
                lines.append(re.sub(r"^remark: [^ ]+ +", "", ln).replace(" [-Rpass-analysis=kernel-resource-usage]", ""))
        if lines:
            (dst / f"{tag}_resource_usage.txt").write_text("\n".join(lines) + "\n")
    print(json.dumps(d, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
