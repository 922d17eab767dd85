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
    assert m.lib().mirt_version() == (0 << 16) | (4 << 8) | 0     # 0.4.0: mirt_grid_plan + MirtGridPlan, mirt_ctx_set_timing, mirt_ctx_frame_stream (0.3.0: MirtParams.frame_begin, MIRT_ERR_SPP_RANGE, MIRT_MAX_SPP_PER_CALL)


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


# ---- mirt_grid_plan: the grid mirt_ctx_set_scene would build, host-only (no device) ----

def _grid_plan(spheres, lds=0):
    arr = (_abi.MirtSphere * len(spheres))(*spheres)
    out = _abi.MirtGridPlan()
    assert m.lib().mirt_grid_plan(C.cast(arr, C.c_void_p), len(spheres), lds, C.byref(out)) == 0, m.lib().mirt_last_error()
    return out


from helpers import bimodal_soup  # noqa: E402


def _entries_at(cs, rs, factor):
    """Cell entries the binning of csrc/mirt_api.hip::build_grid produces at `factor` median radii (same arithmetic, in numpy)."""
    import numpy as np
    cs, rs = cs.astype(np.float64), rs.astype(np.float64)
    cell = factor * np.sort(rs)[len(rs) // 2]
    lo, hi = (cs - rs[:, None]).min(0), (cs + rs[:, None]).max(0)
    while np.prod(np.maximum(1, np.ceil((hi - lo) / cell + 1e-6))) > 4096:
        cell *= 1.26
    dims = np.maximum(1, np.ceil((hi - lo) / cell + 1e-6)).astype(int)
    eps = 1e-3 * cell
    c0 = np.clip(np.floor((cs - rs[:, None] - eps - lo) / cell), 0, dims - 1)
    c1 = np.clip(np.floor((cs + rs[:, None] + eps - lo) / cell), 0, dims - 1)
    return int(np.prod(c1 - c0 + 1, axis=1).sum())


def test_grid_plan_of_the_rtiow_scene():
    scene, _ = m.scenes.rtiow_final()
    g = _grid_plan([s.to_c() for s in scene.spheres])
    assert g.cell_factor == 2.5 and g.n_big == 4 and g.pool_slots == 152            # ground + the three r = 1 spheres stay outside the grid
    assert 0 < g.blob_bytes < 26 * 1024 and g.n_cells <= 4096 and 0 < g.n_entries < 65536         # 24 640 B: two u32 per cell since round 4
    # 45 x 1 x 45 cells of 0.5: the small spheres rest on the ground (y in [0, 0.4]), so the grid is ONE cell high -- what the kernels'
    # two-dimensional walk (grid_walk<COUNT, FLATY>, RenderArgs.grid_flat_y) is for; a second layer would be 4 050 cells
    assert g.n_cells == 45 * 45
    assert _grid_plan([s.to_c() for s in scene.spheres[:20]]).cell_factor == 0.0      # fewer than 32 spheres: no grid


def test_grid_plan_coarsens_past_an_entry_overflow():
    """Round 3's finer default cell must not cost a scene the grid it had at round 2's cell of 4: when the lists overflow
    65 535 entries the builder retries coarser (x 1.26 per step, capped at 4.0 -- never 5.0) instead of giving up."""
    cs, rs = bimodal_soup()
    assert _entries_at(cs, rs, 2.5) > 65535 > _entries_at(cs, rs, 4.0)
    g = _grid_plan([m.Sphere.new(tuple(float(x) for x in c), float(r), 0).to_c() for c, r in zip(cs, rs)])
    assert g.cell_factor == 4.0 and g.n_entries == _entries_at(cs, rs, 4.0)
    assert 0 < g.blob_bytes <= 120 * 1024 - 240                                         # fits the grid layout (include/mirt.h)
    # spheres of ONE size coarsen only as far as LDS asks: never past 4.0
    import numpy as np
    rng = np.random.default_rng(5)
    uni = [m.Sphere.new(tuple(float(x) for x in rng.uniform(-3, 3, 3)), 0.05, 0).to_c() for _ in range(3000)]
    assert 2.5 <= _grid_plan(uni).cell_factor <= 4.0


def test_parked_cell_indices_come_apart_exactly():
    """A cut grid walk parks its LINEAR cell index (16 bit); the kernel takes it apart with float reciprocals computed on the host
    (GridHeader.inv_dim_x / inv_dim_xy): floor((n + 0.5) * fl(1 / d)) == n // d for every n, d <= 8192 (kGridMaxCells is 4 096)."""
    import numpy as np
    n = np.arange(8192, dtype=np.float32) + np.float32(0.5)
    whole = np.arange(8192) 
    for d in range(1, 8193):
        q = (n * (np.float32(1.0) / np.float32(d))).astype(np.uint32)
        assert np.array_equal(q, (whole // d).astype(np.uint32)), d
