#!/usr/bin/env python3
"""Decode the reference's two albedo maps into raw RGB8 texel fixtures.

Runs ONLY in the authoring container (it reads /root/reference/assets); the outputs are
committed as package data (weekend-raytracer-wgpu_amd/assets/) so that nothing at test/bench time needs the reference tree.

    assets/earthmap.jpeg  (1024x512 baseline JPEG)    -> weekend-raytracer-wgpu_amd/assets/earthmap_1024x512_rgb8.npz
    assets/moon.jpeg      (1024x512 progressive JPEG) -> weekend-raytracer-wgpu_amd/assets/moon_1024x512_rgb8.npz

Image credits (reference README.md:123-132): the earth and moon maps ship with
linuxing3/weekend-raytracer-wgpu (MIT), which credits NASA / Solar System Scope textures.

DECODER-UNPINNED: the reference decodes with the `image` crate 0.24.6 (jpeg-decoder); this
script uses Pillow (libjpeg-turbo).  IDCT/upsampling differences of +-1 LSB per texel are
possible and nothing in the reference pins either result.  The fixture is the INPUT of our
parity tests (both oracle and HIP path read the same texels), so the difference cannot
produce a false parity result; it only means a Rust host may upload slightly different texels.
"""
from __future__ import annotations

import hashlib
import sys
from pathlib import Path

import numpy as np
from PIL import Image

REF = Path("/root/reference/assets")
OUT = Path(__file__).resolve().parent.parent / "weekend-raytracer-wgpu_amd" / "assets"


def main() -> int:
    if not REF.is_dir():
        print("reference assets not present; fixtures are already committed", file=sys.stderr)
        return 1
    OUT.mkdir(parents=True, exist_ok=True)
    for name in ("earthmap", "moon"):
        img = Image.open(REF / f"{name}.jpeg").convert("RGB")   # == into_rgba8() minus alpha (texture.rs:27-28)
        arr = np.asarray(img, dtype=np.uint8)                    # [h][w][3], top row first
        h, w, _ = arr.shape
        dst = OUT / f"{name}_{w}x{h}_rgb8.npz"
        np.savez_compressed(dst, rgb8=arr)
        print(dst.name, arr.shape, "sha256", hashlib.sha256(arr.tobytes()).hexdigest()[:16],
              "texel(0,0)", arr[0, 0].tolist())
    return 0


if __name__ == "__main__":
    sys.exit(main())
