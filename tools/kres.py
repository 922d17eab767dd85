#!/usr/bin/env python3
"""Resource usage of every built kernel, one line each (after `make -C weekend-raytracer-wgpu_amd/csrc asm`):
    python tools/kres.py [name-substring]      VGPRs / SGPRs / waves per SIMD / spills / scratch from hipcc's -Rpass-analysis remarks"""
import re
import subprocess
import sys
from pathlib import Path

f = Path(__file__).resolve().parent.parent / "weekend-raytracer-wgpu_amd/csrc/build/resource_usage.txt"
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for b in re.split(r"(?=remark: [^\n]*Function Name)", f.read_text()):
    m = re.search(r"Function Name: (\S+)", b)
    if not m:
        continue
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().replace("mirt::", "").replace("(RenderArgs)", "")
    name = name.replace("void ", "")
    if pat not in name:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    occ, scratch = g(r"Occupancy \[waves/SIMD\]"), g(r"ScratchSize \[bytes/lane\]")
    print(f"{name:92s} VGPR {g('VGPRs'):>3} SGPR {g('TotalSGPRs'):>3} waves {occ} spill s{g('SGPRs Spill')} v{g('VGPRs Spill')} scratch {scratch}")
