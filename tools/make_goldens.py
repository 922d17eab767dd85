#!/usr/bin/env python3
"""Generate tests/golden/images_v1.npz from the CPU oracle (the reference holds no images).

The reference has no golden vectors for this path (parity unpinned), so these
fixtures pin OUR oracle: tests/test_golden.py fails if a later edit changes the oracle's output,
and the GPU tests check the HIP path against the same bytes without needing the oracle.
Run:  python tools/make_goldens.py
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]

import weekend_raytracer_wgpu_amd as m  # noqa: E402
import oracle_binding as ob  # noqa: E402
from helpers import GOLDEN, layer_scene_data, scene_data  # noqa: E402

# name -> (kind, scene, w, h, spp, flags, seed)
CASES = {
    "parity_layer_200x150_spp2": ("parity", "layer", 200, 150, 2, 0, 0),
    "parity_layer_200x150_spp21": ("parity", "layer", 200, 150, 21, 0, 0),
    "parity_single_96x96_spp1": ("parity", "single_sphere", 96, 96, 1, 0, 0),
    "pt_single_96x64_spp8": ("pt", "single_sphere", 96, 64, 8, 0, 0),
    "pt_three_128x72_spp16": ("pt", "three_spheres", 128, 72, 16, 0, 0),
    "pt_three_128x72_spp16_linear_seed7": ("pt", "three_spheres", 128, 72, 16, m.MIRT_FLAG_NO_TONEMAP | m.MIRT_FLAG_NO_SRGB, 7),
    "pt_earth_128x72_spp16": ("pt", "earth", 128, 72, 16, 0, 0),
    "pt_mainrs_96x54_spp70": ("pt", "main_rs_scene", 96, 54, 70, 0, 0),
    "pt_rtiow_64x36_spp4": ("pt", "rtiow_final", 64, 36, 4, 0, 0),
}


def case_inputs(name: str):
    kind, scene, w, h, spp, flags, seed = CASES[name]
    sd = layer_scene_data(w, h) if scene == "layer" else scene_data(scene, w, h)
    mode = m.MIRT_MODE_PARITY if kind == "parity" else m.MIRT_MODE_PT
    return sd, m.make_params(w, h, spp, mode=mode, num_bounces=8, flags=flags, seed=seed)


def main() -> int:
    out = {}
    for name in CASES:
        sd, p = case_inputs(name)
        out[name] = ob.render(sd, p)
        print(name, out[name].shape, out[name][..., :3].reshape(-1, 3).mean(0).round(2))
    np.savez_compressed(GOLDEN / "images_v1.npz", **out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
