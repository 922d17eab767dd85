#!/usr/bin/env python3
"""Mutation fuzzing of the library's JPEG decoder (csrc/mirt_jpeg.cpp) under AddressSanitizer + UBSan, on the CPU.

Builds the decoder alone into tools/_scratch/jpegfuzz/libjpegfuzz.so with -fsanitize=address,undefined, then feeds it
baseline / progressive, 4:4:4 / 4:2:2 / 4:2:0 files (made with Pillow) with 1-5 random byte edits, deletions and insertions
each.  Any out-of-bounds access or undefined operation aborts the process.

    python tools/jpeg_fuzz.py [--seeds 8] [--iterations 4000]
(re-executes itself with libasan preloaded).  Last run: 8 seeds x 4000 files, no finding.
"""
import argparse
import ctypes as C
import io
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
WORK = ROOT / "tools" / "_scratch" / "jpegfuzz"
SRC = ROOT / "weekend-raytracer-wgpu_amd" / "csrc" / "mirt_jpeg.cpp"


def build() -> Path:
    WORK.mkdir(parents=True, exist_ok=True)
    out = WORK / "libjpegfuzz.so"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=undefined", "-o", str(out), str(SRC)], check=True)
    return out


def fuzz(lib_path: str, seed: int, iterations: int) -> int:
    import numpy as np
    from PIL import Image
    lib = C.CDLL(lib_path)
    lib.mirt_jpeg_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.mirt_jpeg_decode_rgb8.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(seed)
    seeds = []
    for prog in (False, True):
        for sub in (0, 1, 2):
            a = (rng.random((40, 56, 3)) * 255).astype(np.uint8)
            buf = io.BytesIO()
            Image.fromarray(a).save(buf, "JPEG", quality=80, subsampling=sub, progressive=prog, restart_marker_blocks=(2 if sub == 0 else 0))
            seeds.append(buf.getvalue())
    decoded = 0
    for it in range(iterations):
        b = bytearray(seeds[it % len(seeds)])
        for _ in range(int(rng.integers(1, 6))):
            kind, pos = int(rng.integers(0, 4)), int(rng.integers(2, len(b)))
            if kind == 0:
                b[pos] = int(rng.integers(0, 256))
            elif kind == 1:
                b[pos] ^= 1 << int(rng.integers(0, 8))
            elif kind == 2:
                del b[pos:pos + int(rng.integers(1, 20))]
            else:
                b[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 8)), dtype=np.uint8))
        b = bytes(b)
        w, h = C.c_uint32(), C.c_uint32()
        if lib.mirt_jpeg_info(b, len(b), C.byref(w), C.byref(h)) != 0 or w.value * h.value > 4_000_000:
            continue
        out = np.empty((h.value, w.value, 3), np.uint8)
        decoded += lib.mirt_jpeg_decode_rgb8(b, len(b), out.ctypes.data_as(C.c_void_p), out.nbytes) == 0
    return decoded


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--iterations", type=int, default=4000)
    ap.add_argument("--child", type=int, default=-1)
    a = ap.parse_args()
    if a.child >= 0:
        print(f"seed {a.child}: {fuzz(str(WORK / 'libjpegfuzz.so'), a.child, a.iterations)} of {a.iterations} mutated files decoded, no sanitizer finding")
        sys.exit(0)
    build()
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    for s in range(1, a.seeds + 1):
        r = subprocess.run([sys.executable, __file__, "--child", str(s), "--iterations", str(a.iterations)], env=env)
        if r.returncode != 0:
            sys.exit(f"seed {s}: sanitizer finding (exit {r.returncode})")
