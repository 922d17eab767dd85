"""K8 + boundary: struct layouts equal the reference's #[repr(C)] types; libmirt.so loads and
exports every symbol include/mirt.h declares; without a GPU the render path fails loudly."""
import ctypes as C
import re
from pathlib import Path

import pytest

import weekend_raytracer_wgpu_amd as m
from weekend_raytracer_wgpu_amd import _abi

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "mirt.h").read_text()


def test_wire_struct_sizes_and_offsets():
    # Sphere 32 B (mod.rs:418-421)
    assert C.sizeof(_abi.MirtSphere) == 32
    assert (_abi.MirtSphere.center.offset, _abi.MirtSphere.radius.offset, _abi.MirtSphere.material_idx.offset) == (0, 16, 20)
    # TextureDescriptor 12 B (mod.rs:869-876)
    assert C.sizeof(_abi.MirtTextureDescriptor) == 12
    # GpuMaterial 32 B: id@0 desc1@4 desc2@16 x@28 (mod.rs:757-765)
    assert C.sizeof(_abi.MirtMaterial) == 32
    M = _abi.MirtMaterial
    assert (M.id.offset, M.desc1.offset, M.desc2.offset, M.x.offset) == (0, 4, 16, 28)
    # GpuCamera 96 B (mod.rs:681-697)
    G = _abi.MirtGpuCamera
    assert C.sizeof(G) == 96
    assert (G.eye.offset, G.horizontal.offset, G.vertical.offset, G.u.offset, G.v.offset,
            G.lens_radius.offset, G.lower_left_corner.offset) == (0, 16, 32, 48, 64, 76, 80)
    # GpuSkyState 144 B: params@0 radiances@108 padding@120 sun@128 (mod.rs:888-896)
    S = _abi.MirtSkyState
    assert C.sizeof(S) == 144
    assert (S.params.offset, S.radiances.offset, S._padding.offset, S.sun_direction.offset) == (0, 108, 120, 128)
    assert C.sizeof(_abi.MirtCamera) == 48
    assert C.sizeof(_abi.MirtSamplingParams) == 12


def test_empty_texture_descriptor():
    d = m.TextureDescriptor.empty()          # mod.rs:878-886
    assert (d.width, d.height, d.offset) == (0, 0, 0xFFFFFFFF)


def test_library_exports_every_declared_symbol():
    declared = set(re.findall(r"\b(mirt_[a-z0-9_]+)\s*\(", HEADER))
    assert declared, "no prototypes found in include/mirt.h"
    lib = m.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libmirt.so does not export {name}"
    # and the Python binding table covers the header exactly
    assert declared == set(_abi.SYMBOLS), declared ^ set(_abi.SYMBOLS)


def test_status_strings_match_header_enum():
    lib = m.lib()
    for name, value in re.findall(r"(MIRT_(?:OK|ERR_[A-Z_]+))\s*=\s*(-?\d+)", HEADER):
        assert lib.mirt_status_string(int(value)).decode() == name
        assert _abi.STATUS[int(value)] == name


def test_version():
    assert m.lib().mirt_version() == (0 << 16) | (3 << 8) | 0     # 0.3.0: MirtParams.frame_begin, MIRT_ERR_SPP_RANGE, MIRT_MAX_SPP_PER_CALL


def test_no_gpu_means_loud_failure_not_fallback():
    """In a container without a GPU the render path must refuse, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(m.MirtError) as e:
        m.Context(0)
    assert e.value.status == _abi.MIRT_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing in the package may import, link or load it."""
    pkg = ROOT / "weekend-raytracer-wgpu_amd"
    for path in list(pkg.rglob("*.py")) + list(pkg.rglob("*.h")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.cpp")) + list(pkg.rglob("Makefile")):
        text = path.read_text()
        assert "oracle" not in text.lower(), f"{path} mentions the oracle"
