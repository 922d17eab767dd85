"""The Rust FFI crate ships as SOURCE ONLY (no Rust toolchain here).  What CAN be checked without a
compiler: it declares every entry point of include/mirt.h, and its #[repr(C)] structs list the same
fields in the same order as the (layout-tested) ctypes mirror."""
import re
from pathlib import Path

import weekend_raytracer_wgpu_amd as m
from weekend_raytracer_wgpu_amd import _abi

ROOT = Path(__file__).resolve().parent.parent
RS = (ROOT / "rust" / "mirt-sys" / "src" / "lib.rs").read_text()
HEADER = (ROOT / "include" / "mirt.h").read_text()


def test_every_header_symbol_is_declared():
    declared = set(re.findall(r"\b(mirt_[a-z0-9_]+)\s*\(", HEADER))
    in_rust = set(re.findall(r"pub fn (mirt_[a-z0-9_]+)\s*\(", RS))
    assert declared == in_rust, declared ^ in_rust


def test_struct_fields_match_the_ctypes_mirror():
    for name in ("MirtSphere", "MirtTextureDescriptor", "MirtMaterial", "MirtGpuCamera", "MirtSkyState", "MirtCamera",
                 "MirtSamplingParams", "MirtScene", "MirtParams", "MirtStats"):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % name, RS, re.S).group(1)
        rust_fields = re.findall(r"pub (\w+):", body)
        py_fields = [f[0] for f in getattr(_abi, name)._fields_]
        assert rust_fields == py_fields, name


def test_flag_constants_match():
    for const in ("MIRT_MODE_PARITY", "MIRT_MODE_PT", "MIRT_FLAG_SKY_HOSEK", "MIRT_FLAG_NO_TONEMAP", "MIRT_FLAG_NO_SRGB",
                  "MIRT_FLAG_COUNT_WORK", "MIRT_FLAG_KERNEL_STRIP", "MIRT_FLAG_KERNEL_POOL"):
        expr = re.search(r"pub const %s: u32 = ([^;]+);" % const, RS).group(1)
        assert eval(expr) == getattr(_abi, const), const
