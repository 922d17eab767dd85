"""C-ABI behaviour on the GPU: error codes (same as the oracle's for the same inputs), context
lifecycle, device-memory rendering into a torch tensor on torch's stream, device de-interleave,
one-shot mirt_render."""
import ctypes as C
import os

import numpy as np
import pytest

import weekend_raytracer_wgpu_amd as m
from weekend_raytracer_wgpu_amd import _abi
from helpers import assert_images_equal, layer_scene_data, scene_data, simple_camera

pytestmark = pytest.mark.gpu


def _status(fn):
    with pytest.raises(m.MirtError) as e:
        fn()
    return e.value.status


def test_render_before_set_scene():
    ctx = m.Context(0)
    assert _status(lambda: ctx.render(m.make_params(8, 8, 1))) == _abi.MIRT_ERR_NO_SCENE
    ctx.close()


@pytest.mark.parametrize("kw,status", [
    (dict(width=0), "MIRT_ERR_VIEWPORT_SIZE"), (dict(height=0), "MIRT_ERR_VIEWPORT_SIZE"),
    (dict(spp=0), "MIRT_ERR_SPP_ZERO"), (dict(mode=7), "MIRT_ERR_BAD_MODE"),
    (dict(spp=(1 << 24) + 1), "MIRT_ERR_SPP_RANGE"), (dict(spp=2, sample_begin=0xffffffff), "MIRT_ERR_SPP_RANGE"),
    (dict(row_begin=10, row_end=5), "MIRT_ERR_BAD_ROWS"), (dict(row_end=99), "MIRT_ERR_BAD_ROWS"),
    (dict(tile_rows=4, n_parts=2, part=2), "MIRT_ERR_BAD_ROWS"),
])
def test_param_errors_match_oracle(gpu_ctx, oracle, kw, status):
    sd = layer_scene_data(32, 24)
    gpu_ctx.set_scene(sd)
    args = dict(width=32, height=24, spp=2)
    args.update(kw)
    p = m.make_params(args.pop("width"), args.pop("height"), args.pop("spp"), **args)
    got = _status(lambda: gpu_ctx.render(p)) if p.width and p.height else None
    if got is None:      # numpy cannot even allocate a 0-sized frame the same way; call the ABI directly
        buf = (C.c_uint8 * 16)()
        got = m.lib().mirt_ctx_render(gpu_ctx._h, C.byref(p), buf, 16)
    assert _abi.STATUS[got] == status
    assert _abi.STATUS[oracle.render_status(sd.as_c(), p)] == status


def test_scene_errors_match_oracle(gpu_ctx, oracle):
    w, h = 16, 16
    cam = simple_camera(w, h)
    one = [m.Sphere.new((0, 0, 0), 1.0, 0).to_c()]
    # parity mode needs material_data[2] (layer.rs:345-349)
    two_mats, tex = m.flatten_materials([m.Material.Metal(m.Texture.new_from_color((1, 1, 1)), 0.1)] * 2)
    sd = m.SceneData(cam, one, two_mats, tex)
    gpu_ctx.set_scene(sd)
    p = m.make_params(w, h, 1)
    assert _abi.STATUS[_status(lambda: gpu_ctx.render(p))] == "MIRT_ERR_MATERIAL_INDEX"
    assert _abi.STATUS[oracle.render_status(sd.as_c(), p)] == "MIRT_ERR_MATERIAL_INDEX"
    # PT: sphere pointing past the material table
    sd2 = m.SceneData(cam, [m.Sphere.new((0, 0, 0), 1.0, 5).to_c()], two_mats, tex)
    gpu_ctx.set_scene(sd2)
    p = m.make_params(w, h, 1, mode=m.MIRT_MODE_PT)
    assert _abi.STATUS[_status(lambda: gpu_ctx.render(p))] == "MIRT_ERR_MATERIAL_INDEX"
    assert _abi.STATUS[oracle.render_status(sd2.as_c(), p)] == "MIRT_ERR_MATERIAL_INDEX"
    # texture descriptor reaching past the texel table
    bad = list(two_mats)
    bad[0] = _abi.MirtMaterial(0, _abi.MirtTextureDescriptor(4, 4, 0), m.TextureDescriptor.empty(), 0.0)
    sd3 = m.SceneData(cam, one, bad, tex)
    gpu_ctx.set_scene(sd3)
    assert _abi.STATUS[_status(lambda: gpu_ctx.render(p))] == "MIRT_ERR_TEXEL_RANGE"
    assert _abi.STATUS[oracle.render_status(sd3.as_c(), p)] == "MIRT_ERR_TEXEL_RANGE"
    # Hosek flag without a sky blob
    sd4 = scene_data("single_sphere", w, h)
    gpu_ctx.set_scene(sd4)
    p = m.make_params(w, h, 1, mode=m.MIRT_MODE_PT, flags=m.MIRT_FLAG_SKY_HOSEK)
    assert _abi.STATUS[_status(lambda: gpu_ctx.render(p))] == "MIRT_ERR_SKY"


def test_scene_too_large_and_null_pointers(gpu_ctx):
    cam = simple_camera(8, 8)
    mats, tex = m.flatten_materials([m.Material.Dielectric(1.5)] * 3)
    many = [m.Sphere.new((0, 0, -5), 0.1, 0).to_c()] * 5000          # 160 KB > the 120 KB LDS budget of the flat kernels, and more
                                                                       # than the 4095 spheres a grid build can index
    assert _abi.STATUS[_status(lambda: gpu_ctx.set_scene(m.SceneData(cam, many, mats, tex)))] == "MIRT_ERR_SCENE_TOO_LARGE"
    lib = m.lib()
    assert lib.mirt_ctx_set_scene(gpu_ctx._h, None) == _abi.MIRT_ERR_NULL_POINTER
    assert lib.mirt_ctx_render(gpu_ctx._h, None, None, 0) == _abi.MIRT_ERR_NULL_POINTER
    assert lib.mirt_ctx_create(99, C.byref(C.c_void_p())) == _abi.MIRT_ERR_NO_DEVICE
    assert b"out of range" in lib.mirt_last_error()


def test_out_buffer_too_small(gpu_ctx):
    sd = layer_scene_data(32, 24)
    gpu_ctx.set_scene(sd)
    p = m.make_params(32, 24, 2)
    buf = (C.c_uint8 * 100)()
    assert m.lib().mirt_ctx_render(gpu_ctx._h, C.byref(p), buf, 100) == _abi.MIRT_ERR_OUT_BUFFER


def test_one_shot_render(oracle):
    w, h = 64, 48
    sd = layer_scene_data(w, h)
    p = m.make_params(w, h, 2)
    out = np.zeros((h, w, 4), np.uint8)
    c = sd.as_c()
    assert m.lib().mirt_render(C.byref(c), C.byref(p), 0, out.ctypes.data_as(C.c_void_p), out.nbytes) == 0
    assert_images_equal(out, oracle.render(sd, p), "mirt_render")


def test_render_into_torch_tensor_and_device_deinterleave(gpu_ctx):
    """The bench path: frames stay in HBM; kernels run on torch's current stream."""
    import torch
    w, h, spp = 96, 50, 24
    sd = scene_data("three_spheres", w, h)
    gpu_ctx.set_scene(sd)
    base = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT)
    want = gpu_ctx.render(base)
    gpu_ctx.stats()                     # drain: count only the launches below
    frame = m.multi_gpu.TiledFrame(gpu_ctx, base, 0, 1)
    got = frame.step()
    torch.cuda.synchronize()
    assert_images_equal(got.cpu().numpy(), want, "render_device world=1")
    # emulate 4 ranks on one GPU: render each part into its slot, de-interleave on the device
    world, tr = 4, 4
    max_rows = m.multi_gpu.max_part_rows(base, world, tr)
    parts = torch.zeros((world, max_rows, w, 4), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for r in range(world):
        pr = m.multi_gpu.part_params(base, r, world, tr)
        rows = m.params_out_rows(pr)
        gpu_ctx.render_device(pr, parts[r].data_ptr(), rows * w * 4, stream)
    out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    gpu_ctx.deinterleave_device(m.multi_gpu.part_params(base, 0, world, tr), parts.data_ptr(), max_rows * w * 4,
                                out.data_ptr(), out.numel(), stream)
    torch.cuda.synchronize()
    assert_images_equal(out.cpu().numpy(), want, "device de-interleave")
    st = gpu_ctx.stats()
    assert st["launches"] == world + 1 and st["kernel_ms_total"] >= st["kernel_ms"] > 0


def test_collective_path_on_a_single_rank_rccl_group(gpu_ctx):
    """One GPU per box, so the N > 1 bench path cannot run here; what can is its plumbing: an RCCL process group
    of one rank drives TiledFrame's collective code -- synchronous gather, and the pipelined variant (async
    gather into double buffers, stream-level waits, flush) -- and every frame must arrive intact, also when the
    content changes from step to step."""
    import socket
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    try:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", gpu_ctx.device))
    except Exception as e:  # pragma: no cover - environment without RCCL
        pytest.skip(f"RCCL process group unavailable: {e}")
    try:
        w, h, spp = 160, 90, 16
        gpu_ctx.set_scene(scene_data("three_spheres", w, h))
        side = torch.cuda.Stream()
        # default stream, then a side stream; last: two frames in flight (renders on the context's frame streams, what bench.py's N > 1 runs do)
        for pipelined, stream, in_flight in ((False, None, 1), (True, None, 1), (False, side, 1), (True, side, 1), (True, None, 2), (True, side, 2)):
            with torch.cuda.stream(stream):
                base = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8)
                fr = m.multi_gpu.TiledFrame(gpu_ctx, base, 0, 1, pipelined=pipelined, _rehearse_single_rank=True, frames_in_flight=in_flight)
                for seed in range(5):
                    fr.params.seed = seed
                    fr.step()
                fr.flush()
                t = torch.tensor([2.5], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dist.barrier()
            torch.cuda.synchronize()
            want = gpu_ctx.render(m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, seed=4))
            assert_images_equal(fr.frame.cpu().numpy(), want, f"last of five frames, pipelined={pipelined}, side stream={stream is not None}, {in_flight} in flight")
            assert float(t.item()) == 2.5
    finally:
        dist.destroy_process_group()


def test_set_camera_only(gpu_ctx, oracle):
    w, h = 64, 48
    sd = layer_scene_data(w, h)
    gpu_ctx.set_scene(sd)
    cam2 = simple_camera(w, h, eye=(0, 2, 9), vfov=40.0)
    gpu_ctx.set_camera(cam2)
    p = m.make_params(w, h, 2)
    sd2 = m.SceneData(cam2, list(sd.spheres), list(sd.materials), sd.texels)
    assert_images_equal(gpu_ctx.render(p), oracle.render(sd2, p), "set_camera")


def test_many_contexts_and_reuse(oracle):
    w, h = 32, 24
    sd = layer_scene_data(w, h)
    p = m.make_params(w, h, 2)
    want = oracle.render(sd, p)
    for _ in range(3):
        with m.Context(0) as ctx:
            ctx.set_scene(sd)
            for _ in range(70):          # wraps the 64-entry event pool
                img = ctx.render(p)
            assert_images_equal(img, want, "reuse")
            assert ctx.stats()["launches"] == 70


def test_fast_sqrt_rcp_are_correctly_rounded(gpu_ctx):
    """The kernels' 5-instruction sqrt and 3-instruction reciprocal must equal the IEEE results for
    EVERY binary32 bit pattern on this device — the bit-parity claim against the CPU rests on it."""
    assert gpu_ctx.selftest_math() == (0, 0)


def test_large_scene_with_one_material_per_sphere(gpu_ctx, oracle):
    """2000 spheres with 2000 materials (64 KB + 96 KB of tables): beyond what fits LDS with the materials,
    so only the grid build (materials in global memory) can run it — and the counting/flat builds must refuse
    cleanly or fall back, never fault."""
    rng = np.random.default_rng(5)
    T = m.Texture
    mats, spheres = [], []
    for i in range(2000):
        mats.append(m.Material.Lambertian(T.new_from_color(rng.random(3))) if i % 3 else m.Material.Metal(T.new_from_color(rng.random(3)), 0.2))
        c = rng.normal(size=3) * 6.0
        c[1] = abs(c[1]) * 0.2 + 0.1
        spheres.append(m.Sphere.new(c, float(rng.uniform(0.05, 0.2)), i).to_c())
    gm, tex = m.flatten_materials(mats)
    w, h = 64, 36
    sd = m.SceneData(simple_camera(w, h, eye=(0, 2, 14), vfov=40, focus=14), spheres, gm, tex)
    gpu_ctx.set_scene(sd)
    p = m.make_params(w, h, 4, mode=m.MIRT_MODE_PT, num_bounces=4)
    assert_images_equal(gpu_ctx.render(p), oracle.render(sd, p), "2000 spheres / 2000 materials")
    for bad in (m.make_params(w, h, 4, mode=m.MIRT_MODE_PT, flags=m.MIRT_FLAG_NO_GRID), m.make_params(w, h, 4, mode=m.MIRT_MODE_PT, flags=m.MIRT_FLAG_COUNT_WORK)):
        with pytest.raises(m.MirtError) as e:
            gpu_ctx.render(bad)
        assert e.value.status_name == "MIRT_ERR_SCENE_TOO_LARGE"


def test_set_camera_is_per_launch_and_needs_no_sync(gpu_ctx, oracle):
    """`Layer::update_camera` / `set_render_params` change the camera every interactive frame (layer.rs:188-193,
    mod.rs:353-388).  The camera travels by value with each launch: launches queued BEFORE a set_camera keep the
    old camera, launches after it see the new one, and nothing synchronises in between."""
    import torch
    w, h, spp = 128, 72, 32
    sd_a = scene_data("three_spheres", w, h)
    cam_b = simple_camera(w, h, eye=(-3.0, 2.5, 6.0), direction=(0.4, -0.2, -1.0), vfov=40.0, aperture=0.2, focus=7.0)
    sd_b = m.SceneData(cam_b, sd_a.spheres, sd_a.materials, sd_a.texels)
    p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT)
    gpu_ctx.set_scene(sd_a)
    s = torch.cuda.Stream()
    bufs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(4)]
    with torch.cuda.stream(s):
        big = m.make_params(640, 360, 256, mode=m.MIRT_MODE_PT)            # keeps the stream busy while the host runs ahead
        scratch = torch.zeros((360, 640, 4), dtype=torch.uint8, device="cuda")
        gpu_ctx.render_device(big, scratch.data_ptr(), scratch.numel(), s.cuda_stream)
        gpu_ctx.render_device(p, bufs[0].data_ptr(), bufs[0].numel(), s.cuda_stream)      # camera A
        gpu_ctx.set_camera(cam_b)
        gpu_ctx.render_device(p, bufs[1].data_ptr(), bufs[1].numel(), s.cuda_stream)      # camera B
        gpu_ctx.set_camera(sd_a.camera)
        gpu_ctx.render_device(p, bufs[2].data_ptr(), bufs[2].numel(), s.cuda_stream)      # camera A again
        gpu_ctx.set_camera(cam_b)
        gpu_ctx.render_device(p, bufs[3].data_ptr(), bufs[3].numel(), s.cuda_stream)
    s.synchronize()
    want_a, want_b = oracle.render(sd_a, p), oracle.render(sd_b, p)
    assert not np.array_equal(want_a, want_b)
    for i, want in enumerate((want_a, want_b, want_a, want_b)):
        assert_images_equal(bufs[i].cpu().numpy(), want, f"launch {i}")
    gpu_ctx.set_camera(sd_a.camera)


def test_launches_of_one_context_overlap_on_two_streams(gpu_ctx, oracle):
    """Every launch owns its strip dispenser and work counters: two renders of ONE context queued on different
    streams may run at the same time without handing out a strip twice or skipping one."""
    import torch
    w, h = 320, 200
    sd = scene_data("three_spheres", w, h)
    gpu_ctx.set_scene(sd)
    pa = m.make_params(w, h, 96, mode=m.MIRT_MODE_PT, seed=1)
    pb = m.make_params(w, h, 64, mode=m.MIRT_MODE_PT, seed=2, flags=m.MIRT_FLAG_KERNEL_STRIP)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(3):
        a = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        b = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.current_stream().synchronize()         # the fills run on torch's current stream, the renders on s1 / s2: order them
        gpu_ctx.render_device(pa, a.data_ptr(), a.numel(), s1.cuda_stream)
        gpu_ctx.render_device(pb, b.data_ptr(), b.numel(), s2.cuda_stream)
        outs.append((a, b))
    torch.cuda.synchronize()
    want_a, want_b = oracle.render(sd, pa), oracle.render(sd, pb)
    for a, b in outs:
        assert_images_equal(a.cpu().numpy(), want_a, "stream 1 (pool kernel)")
        assert_images_equal(b.cpu().numpy(), want_b, "stream 2 (strip kernel)")


def test_accum_resolve_waits_for_adds_on_a_caller_stream(gpu_ctx, oracle):
    import torch
    w, h = 160, 90
    sd = scene_data("three_spheres", w, h)
    gpu_ctx.set_scene(sd)
    p = m.make_params(w, h, 16, mode=m.MIRT_MODE_PT)
    s = torch.cuda.Stream()
    gpu_ctx.accum_reset(p)
    for _ in range(6):
        gpu_ctx.accum_add(p, s.cuda_stream)
    got = gpu_ctx.accum_resolve(p)                       # no explicit sync of `s` by the caller
    sums = gpu_ctx.accum_read(p)
    want = m.make_params(w, h, 96, mode=m.MIRT_MODE_PT)
    assert_images_equal(got, oracle.render(sd, want), "6 x 16 spp on a side stream")
    assert np.array_equal(sums, oracle.render_pt_sums(sd, want))


def test_accumulation_on_the_default_stream_then_resolve(gpu_ctx, oracle):
    """The reference's progressive loop queued on torch's CURRENT stream when no side stream is set -- handle 0, which the binding passes
    as hipStreamLegacy -- then resolved: frames of 2 samples per pixel drawn from the shader's per-frame RNG stream, resolve and read after
    every few of them.  (Round 4: accum_resolve made the context's stream wait for an event recorded on the legacy stream with
    hipStreamWaitEvent, which crashes inside the HIP runtime; the host waits for the event now.)"""
    import torch
    w, h = 160, 90
    sd = scene_data("main_rs_scene", w, h)
    gpu_ctx.set_scene(sd)
    st = torch.cuda.current_stream().cuda_stream
    assert st == 0
    p = m.make_params(w, h, 2, mode=m.MIRT_MODE_PT, frame_spp=2)
    gpu_ctx.accum_reset(p)
    for frames in (1, 4, 16):
        while gpu_ctx.accum_samples() < 2 * frames:
            gpu_ctx.accum_add(p, st)
        got = gpu_ctx.accum_resolve(p)                   # no explicit sync by the caller
        sums = gpu_ctx.accum_read(p)
        want = m.make_params(w, h, 2 * frames, mode=m.MIRT_MODE_PT, frame_spp=2)
        assert_images_equal(got, oracle.render(sd, want), f"{frames} frames of 2 spp on the default stream")
        assert np.array_equal(sums, oracle.render_pt_sums(sd, want))


def test_frame_streams_keep_two_frames_in_flight(oracle):
    """mirt_ctx_frame_stream: the context's two streams on different hardware queues.  Frames alternate between them and between two
    framebuffers -- dispensed, one-unit-per-wave and pooled launches, timing on and off -- and every frame checked is the oracle's;
    mirt_ctx_synchronize waits for both streams; the handles are stable, differ, and index 2 is refused."""
    import torch
    w, h = 320, 180
    sd = scene_data("three_spheres", w, h)
    ctx = m.Context(0)
    try:
        ctx.set_scene(sd)
        s0, s1 = ctx.frame_stream(0), ctx.frame_stream(1)
        assert s0 != s1 and s0 != 0 and s1 != 0 and (ctx.frame_stream(0), ctx.frame_stream(1)) == (s0, s1)
        assert _abi.STATUS[_status(lambda: ctx.frame_stream(2))] == "MIRT_ERR_BAD_ROWS"
        bufs = [torch.empty((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
        for timing in (True, False):
            ctx.set_timing(timing)
            for spp, flags in ((2, 0), (8, 0), (24, 0), (48, m.MIRT_FLAG_KERNEL_POOL)):
                ps = [m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, seed=k, flags=flags) for k in range(2)]
                want = [oracle.render(sd, p) for p in ps]
                for b in bufs:
                    b.fill_(0xAB)
                torch.cuda.synchronize()
                ctx.stats()
                for i in range(40):
                    ctx.render_device(ps[i % 2], bufs[i % 2].data_ptr(), bufs[0].numel(), (s0, s1)[i % 2])
                ctx.synchronize()                              # no torch synchronisation: the context waits for both of its streams
                for k in range(2):
                    assert not _frames_differ(bufs[k].cpu().numpy(), want[k]), (timing, spp, k, _frames_differ(bufs[k].cpu().numpy(), want[k]))
                assert ctx.stats()["launches"] == 40
    finally:
        ctx.close()


def test_set_scene_is_failure_atomic(gpu_ctx):
    """A set_scene that fails (here: rejected up front) or half-fails leaves the context without a scene
    rather than with stale tables: the next render must answer MIRT_ERR_NO_SCENE or render the OLD scene whole."""
    w, h = 32, 24
    sd = layer_scene_data(w, h)
    gpu_ctx.set_scene(sd)
    p = m.make_params(w, h, 2)
    want = gpu_ctx.render(p)
    cam = simple_camera(8, 8)
    mats, tex = m.flatten_materials([m.Material.Dielectric(1.5)] * 3)
    many = [m.Sphere.new((0, 0, -5), 0.1, 0).to_c()] * 5000
    assert _abi.STATUS[_status(lambda: gpu_ctx.set_scene(m.SceneData(cam, many, mats, tex)))] == "MIRT_ERR_SCENE_TOO_LARGE"
    assert_images_equal(gpu_ctx.render(p), want, "old scene intact after a rejected set_scene")


def _soup(n_small, n_large, spread):
    """n_small spheres of radius 0.01 on a jittered lattice + n_large of radius 1 (> 4 median radii: 'big' for the grid builder)."""
    rng = np.random.default_rng(5)
    side = int(np.ceil(np.sqrt(max(1, n_small))))
    sph = []
    for i in range(n_small):
        sph.append(m.Sphere.new((spread * (i % side) / side - spread / 2 + 0.001 * rng.random(), 0.0, spread * (i // side) / side - spread / 2), 0.01, 0).to_c())
    for i in range(n_large):
        sph.append(m.Sphere.new((float(i) * 3.0, 5.0, -20.0), 1.0, 0).to_c())
    return sph


def _frames_differ(got, want):
    """'' if the frames are equal, else where and how they differ (for an assertion message that can be diagnosed from one failure)."""
    got, want = np.asarray(got), np.asarray(want)
    if got.shape == want.shape and np.array_equal(got, want):
        return ""
    if got.shape != want.shape:
        return f"shape {got.shape} vs {want.shape}"
    bad = np.argwhere((got != want).any(axis=-1))
    untouched = int((got[tuple(bad.T)] == 0xAB).all(axis=-1).sum())
    rows = np.unique(bad[:, 0])
    return (f"{len(bad)} of {got.shape[0] * got.shape[1]} pixels differ ({untouched} still hold the 0xAB fill), rows {rows[:6].tolist()}"
            f"{'...' if len(rows) > 6 else ''}, first at {bad[0].tolist()}: got {got[tuple(bad[0])].tolist()} want {want[tuple(bad[0])].tolist()}")


def _context_with_dispensed_units():
    """A context whose lane-per-pixel launches take their units from the dispenser (rounds 2-3's schedule; round 4 runs one unit per
    wave there): the tuning knobs are read once, in mirt_ctx_create."""
    import os
    saved = os.environ.get("MIRT_STATIC_UNITS")
    os.environ["MIRT_STATIC_UNITS"] = "0"
    try:
        return m.Context(0)
    finally:
        if saved is None:
            del os.environ["MIRT_STATIC_UNITS"]
        else:
            os.environ["MIRT_STATIC_UNITS"] = saved


def test_dispenser_slots_are_clean_across_the_event_pool(oracle):
    """A launch takes its work units from dispenser words that belong to its slot of the event ring and start at zero (no memset
    node in front of the kernel); a slot's words are re-zeroed behind the kernel that used them, on the context's own stream, and
    the slot's next user -- 64 launches later -- waits for that.  150 dispensed launches in a row -- the ring wraps twice inside
    mirt_ctx_render_device -- must all produce the oracle's frame, in the strip and in the pooled kernel."""
    import torch
    w, h = 800, 400                  # 20 000 units of 16 pixels: more than the waves of a launch, so most units come from the dispenser
    sd = scene_data("three_spheres", w, h)
    ctx = _context_with_dispensed_units()
    ctx.set_scene(sd)
    out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for spp, flags in ((4, 0), (28, m.MIRT_FLAG_KERNEL_POOL)):
        p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, flags=flags)
        # the pooled kernel's frame against the strip kernel's (itself held against the oracle by the first case and by test_gpu_pt.py)
        want = oracle.render(sd, p) if flags == 0 else ctx.render(m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, flags=m.MIRT_FLAG_KERNEL_STRIP))
        ctx.stats()
        for i in range(150):
            ctx.render_device(p, out.data_ptr(), out.numel(), stream)
            if i % 37 == 0 or i == 149:
                torch.cuda.synchronize()
                assert np.array_equal(out.cpu().numpy(), want), (spp, i)
        assert ctx.stats()["launches"] == 150 and ctx.last_kernel().startswith("render_pt_pool" if flags else "render_pt_str")
    ctx.close()


def test_the_event_ring_wraps_with_launches_on_two_streams(oracle):
    """The ring of 64 launch slots with launches of ONE context alternating between two caller streams (they overlap on the GPU):
    200 launches -- three wraps -- of two different dispensed workloads, no synchronisation in between except every 61st pair, and
    every frame checked.  A slot handed to a new launch before its previous user had finished, or before its dispenser words were
    zero again, would skip units (pixels left at the 0xAB fill) or hand one out twice."""
    import torch
    w, h = 640, 360
    sd = scene_data("three_spheres", w, h)
    ctx = _context_with_dispensed_units()
    ctx.set_scene(sd)
    pa = m.make_params(w, h, 8, mode=m.MIRT_MODE_PT, seed=3)                                   # strip kernel, lane = pixel, dispensed units
    pb = m.make_params(w, h, 32, mode=m.MIRT_MODE_PT, seed=4, flags=m.MIRT_FLAG_KERNEL_POOL)   # pooled kernel
    want_a, want_b = oracle.render(sd, pa), oracle.render(sd, pb)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    a = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
    b = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
    ctx.stats()
    for i in range(100):
        check = i % 61 == 0 or i == 99
        if check:
            torch.cuda.synchronize()
            a.fill_(0xAB)
            b.fill_(0xAB)
            torch.cuda.synchronize()
        ctx.render_device(pa, a.data_ptr(), a.numel(), s1.cuda_stream)
        ctx.render_device(pb, b.data_ptr(), b.numel(), s2.cuda_stream)
        if check:
            torch.cuda.synchronize()
            assert not _frames_differ(a.cpu().numpy(), want_a), ("stream 1", i, _frames_differ(a.cpu().numpy(), want_a))
            assert not _frames_differ(b.cpu().numpy(), want_b), ("stream 2", i, _frames_differ(b.cpu().numpy(), want_b))
    st = ctx.stats()
    assert st["launches"] == 200 and st["kernel_ms_total"] > 0.0
    # a statically dealt launch (2 spp: the reference's interactive frame) between dispensed ones leaves its slot's words alone
    p2 = m.make_params(w, h, 2, mode=m.MIRT_MODE_PT)
    want2 = oracle.render(sd, p2)
    for i in range(70):
        ctx.render_device(p2 if i % 2 else pa, a.data_ptr(), a.numel(), s1.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(a.cpu().numpy(), want2)
    assert ctx.stats()["launches"] == 70
    ctx.close()


def test_untimed_launches_render_the_same_frames_and_report_no_kernel_time(oracle):
    """mirt_ctx_set_timing(0): launches carry no event unless the context must learn that they have finished (dispensed units: their
    dispenser words are re-zeroed; counting launches).  Same frames; get_stats counts the launches and reports 0 ms; the ring still
    wraps cleanly with dispensed launches on two streams; synchronize covers launches that carried no event at all."""
    import torch
    w, h = 640, 360
    sd = scene_data("three_spheres", w, h)
    ctx = _context_with_dispensed_units()
    ctx.set_scene(sd)
    ctx.set_timing(False)
    p2 = m.make_params(w, h, 2, mode=m.MIRT_MODE_PT)                                           # one unit per wave: no event at all (default context below)
    p8 = m.make_params(w, h, 8, mode=m.MIRT_MODE_PT, seed=3)                                   # dispensed units: an end event only
    pp = m.make_params(w, h, 32, mode=m.MIRT_MODE_PT, seed=4, flags=m.MIRT_FLAG_KERNEL_POOL)   # pooled kernel, dispensed
    want2, want8, wantp = oracle.render(sd, p2), oracle.render(sd, p8), oracle.render(sd, pp)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    a = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
    b = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
    c2 = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
    ctx.stats()
    for i in range(100):                                   # three wraps of the ring, two caller streams, dispensed launches without timing
        check = i % 61 == 0 or i == 99
        if check:
            torch.cuda.synchronize()
            a.fill_(0xAB)
            b.fill_(0xAB)
            torch.cuda.synchronize()
        ctx.render_device(p8, a.data_ptr(), a.numel(), s1.cuda_stream)
        ctx.render_device(pp, b.data_ptr(), b.numel(), s2.cuda_stream)
        if check:
            torch.cuda.synchronize()
            assert not _frames_differ(a.cpu().numpy(), want8), ("stream 1", i, _frames_differ(a.cpu().numpy(), want8))
            assert not _frames_differ(b.cpu().numpy(), wantp), ("stream 2", i, _frames_differ(b.cpu().numpy(), wantp))
    st = ctx.stats()
    assert st["launches"] == 200 and st["kernel_ms_total"] == 0.0 and st["kernel_ms"] == 0.0
    c2.fill_(0xAB)
    torch.cuda.synchronize()
    plain = m.Context(0)                                   # the default schedule: a 2-spp frame runs one unit per wave, no dispenser
    plain.set_scene(sd)
    plain.set_timing(False)
    for _ in range(150):                                   # the reference's interactive frame: no event, no slot bookkeeping
        plain.render_device(p2, c2.data_ptr(), c2.numel(), s1.cuda_stream)
    plain.synchronize()                                    # no torch synchronisation: the context waits for the caller's stream itself
    assert np.array_equal(c2.cpu().numpy(), want2)
    st = plain.stats()
    assert st["launches"] == 150 and st["kernel_ms_total"] == 0.0
    plain.close()
    # counting launches keep their end event (their counters are read afterwards), timing on again reports times again
    pc = m.make_params(w, h, 8, mode=m.MIRT_MODE_PT, seed=3, flags=m.MIRT_FLAG_COUNT_WORK)
    assert np.array_equal(ctx.render(pc), want8)
    assert ctx.stats()["rays"] == oracle_counts(oracle, sd, pc)["rays"]
    ctx.set_timing(True)
    assert np.array_equal(ctx.render(p8), want8)
    st = ctx.stats()
    assert st["launches"] == 1 and st["kernel_ms_total"] > 0.0
    ctx.close()


def oracle_counts(oracle, sd, p):
    oracle.render(sd, p)
    return oracle.stats()


def test_scene_limits_at_the_boundary(gpu_ctx):
    """include/mirt.h, MIRT_ERR_SCENE_TOO_LARGE: the flat layout holds 96 + 32 n + 48 m + 144 <= 122 880 bytes -- 3 831 spheres
    with one material --, the grid layout at most 4 095 spheres; a scene that fits only the grid layout renders in path-traced
    mode and refuses the launches that need the flat scan."""
    mats, tex = m.flatten_materials([m.Material.Lambertian(albedo=m.Texture.new_from_color((0.5, 0.5, 0.5)))])
    cam = simple_camera(32, 16)
    # 100 large spheres make the grid builder give up (more than 64 'big' ones): only the flat layout is left
    ok = m.SceneData(cam, _soup(3731, 100, 4.0), mats, tex)
    gpu_ctx.set_scene(ok)                                           # 3 831 spheres: fits exactly
    assert gpu_ctx.render(m.make_params(32, 16, 1, mode=m.MIRT_MODE_PT, num_bounces=2)).shape == (16, 32, 4)
    with pytest.raises(m.MirtError) as e:
        gpu_ctx.set_scene(m.SceneData(cam, _soup(3732, 100, 4.0), mats, tex))       # one more: 32 bytes over
    assert e.value.status == _abi.MIRT_ERR_SCENE_TOO_LARGE
    # a failed set_scene leaves the previous scene in place (failure-atomic)
    assert gpu_ctx.render(m.make_params(32, 16, 1, mode=m.MIRT_MODE_PT, num_bounces=2)).shape == (16, 32, 4)
    # 4 000 small spheres: too many for the flat layout, fine for the grid layout
    grid_only = m.SceneData(cam, _soup(4000, 0, 8.0), mats, tex)
    gpu_ctx.set_scene(grid_only)
    assert gpu_ctx.render(m.make_params(32, 16, 2, mode=m.MIRT_MODE_PT, num_bounces=2)).shape == (16, 32, 4)
    assert "true>" in gpu_ctx.last_kernel() or ",true," in gpu_ctx.last_kernel()          # a grid build ran
    for p in (m.make_params(32, 16, 2, mode=m.MIRT_MODE_PT, flags=m.MIRT_FLAG_NO_GRID), m.make_params(32, 16, 2, mode=m.MIRT_MODE_PT, flags=m.MIRT_FLAG_COUNT_WORK)):
        with pytest.raises(m.MirtError) as e:
            gpu_ctx.render(p)
        assert e.value.status == _abi.MIRT_ERR_SCENE_TOO_LARGE
    # 4 096 spheres: past the 12-bit sphere ids of the grid layout as well
    with pytest.raises(m.MirtError) as e:
        gpu_ctx.set_scene(m.SceneData(cam, _soup(4096, 0, 8.0), mats, tex))
    assert e.value.status == _abi.MIRT_ERR_SCENE_TOO_LARGE


_SEQ0 = int(os.environ.get("MIRT_SEQ_FIRST_SEED", "0"))


@pytest.mark.parametrize("seed", range(_SEQ0, _SEQ0 + int(os.environ.get("MIRT_SEQ_SEEDS", "6"))))      # soaks: MIRT_SEQ_SEEDS=200
def test_random_call_sequences_keep_the_context_consistent(oracle, seed):
    """The boundary as a state machine: ~70 random calls per seed against ONE context -- set_scene, set_camera, blocking renders, device
    renders queued on the context's stream / the legacy default stream / two side streams (checked only at the next synchronisation point, so
    they overlap), progressive accumulation (reset / add / resolve / read), get_stats, set_timing, synchronize -- in both modes, every
    schedule the sample count and flags select, dispensed and one-unit-per-wave launches, whole frames, bands of rows and single ranks' tiles
    of N-way partitions.  Every frame and every sum must be the oracle's
    for the scene and camera that were current WHEN THE CALL WAS ISSUED."""
    import torch
    rng = np.random.default_rng(9000 + seed)
    w, h = int(rng.integers(40, 120)), int(rng.integers(24, 70))
    names = ["three_spheres", "main_rs_scene", "single_sphere", "rtiow_final", "layer"]

    def load(name):
        return layer_scene_data(w, h) if name == "layer" else scene_data(name, w, h)

    def with_camera(sd, cam):
        return m.SceneData(cam, sd.spheres, sd.materials, sd.texels, sd.sky)

    ctx = _context_with_dispensed_units() if rng.random() < 0.5 else m.Context(0)
    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    streams = [None, 0, side[0].cuda_stream, side[1].cuda_stream]
    sd = load(names[int(rng.integers(len(names)))])
    ctx.set_scene(sd)
    pending = []                                   # (device buffer, expected frame, what) of renders not yet synchronised
    acc = None                                     # (params of the epoch, stream of the epoch, scene at reset) of the running accumulation

    def params():
        pt = len(sd.materials) < 3 or rng.random() < 0.75           # (the parity loop reads material_data[2], layer.rs:345-349)
        spp = int(rng.choice([1, 2, 3, 5, 8, 16, 24, 40, 70]))
        if len(sd.spheres) > 100:
            spp = min(spp, 24)                                       # (keeps the checker's flat scan of a many-sphere scene short)
        flags = 0
        if pt:
            flags = int(rng.choice([0, 0, 0, m.MIRT_FLAG_KERNEL_STRIP, m.MIRT_FLAG_KERNEL_POOL, m.MIRT_FLAG_NO_TONEMAP, m.MIRT_FLAG_NO_SRGB]))
            fs = int(rng.choice([0, 0, 1, spp]))
            return m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=int(rng.integers(1, 9)), flags=flags, seed=int(rng.integers(0, 1 << 40)),
                                 frame_spp=fs)
        return m.make_params(w, h, min(spp, 24), mode=m.MIRT_MODE_PARITY, flags=int(rng.choice([0, m.MIRT_FLAG_KERNEL_STRIP])))

    def partitioned(p):
        """Sometimes a band of rows or one rank's tiles of an N-way partition (MirtParams.row_begin/row_end, tile_rows/n_parts/part)."""
        r = rng.random()
        if r < 0.15:
            a, b = sorted(int(x) for x in rng.integers(0, h + 1, size=2))
            if a < b:
                p.row_begin, p.row_end = a, b
        elif r < 0.3:
            p.tile_rows, p.n_parts = int(rng.choice([1, 4, 16])), int(rng.integers(2, 9))
            p.part = int(rng.integers(p.n_parts))
            if m.params_out_rows(p) == 0:
                p.tile_rows = p.n_parts = p.part = 0
        return p

    def drain():
        torch.cuda.synchronize()
        for buf, want, what in pending:
            assert_images_equal(buf.cpu().numpy()[:want.shape[0]], want, f"seed {seed}: {what}")
        pending.clear()

    try:
        for step in range(70):
            op = rng.choice(["scene", "camera", "render", "device", "device", "device", "reset", "add", "add", "resolve", "read", "stats", "timing", "sync"])
            if op == "scene":
                name = names[int(rng.integers(len(names)))]
                sd = load(name)
                ctx.set_scene(sd)                                  # waits for what is in flight (launches keep the tables they were issued with)
                acc = None
            elif op == "camera":
                fc = m.FlyCameraController.default()
                cam = fc.renderer_camera()
                cam.eye_pos = cam.eye_pos + rng.normal(size=3).astype(np.float32) * 0.3
                cam.aperture = float(rng.choice([0.0, cam.aperture, 0.3]))
                gcam = m.GpuCamera.new(cam, (w, h)).c
                ctx.set_camera(gcam)
                sd = with_camera(sd, gcam)
                acc = None                                         # (the sums of another view are not this one's: the host resets, as Raytracer does)
            elif op == "render":
                p = partitioned(params())
                assert_images_equal(ctx.render(p), oracle.render(sd, p), f"seed {seed} step {step}: blocking render")
            elif op == "device":
                p = partitioned(params())
                buf = torch.full((h, w, 4), 0xAB, dtype=torch.uint8, device="cuda")
                torch.cuda.current_stream().synchronize()          # the fill must land before a side stream's kernel writes the buffer
                st = streams[int(rng.integers(len(streams)))]
                ctx.render_device(p, buf.data_ptr(), buf.numel(), st)
                pending.append((buf, oracle.render(sd, p), f"step {step}: device render on stream {st}, {ctx.last_kernel()}"))
                if len(pending) >= 6:
                    drain()
            elif op == "reset":
                p = m.make_params(w, h, int(rng.choice([1, 2, 4, 16])), mode=m.MIRT_MODE_PT, num_bounces=int(rng.integers(1, 9)),
                                  frame_spp=0, seed=int(rng.integers(0, 1 << 20)))
                if rng.random() < 0.5:
                    p.frame_spp = p.spp
                ctx.accum_reset(p)
                acc = (p, streams[int(rng.integers(len(streams)))], sd)
            elif op == "add" and acc is not None:
                ctx.accum_add(acc[0], acc[1])                      # one stream per epoch: adds must be ordered among themselves (mirt.h)
            elif op in ("resolve", "read") and acc is not None and ctx.accum_samples() > 0:
                p, _, sd0 = acc
                total = m.make_params(w, h, ctx.accum_samples(), mode=m.MIRT_MODE_PT, num_bounces=p.num_bounces, frame_spp=p.frame_spp, seed=p.seed)
                if op == "resolve":
                    assert_images_equal(ctx.accum_resolve(p), oracle.render(sd0, total), f"seed {seed} step {step}: resolve after {ctx.accum_samples()} samples")
                else:
                    assert np.array_equal(ctx.accum_read(p), oracle.render_pt_sums(sd0, total)), f"seed {seed} step {step}: sums after {ctx.accum_samples()} samples"
            elif op == "stats":
                st = ctx.stats()
                assert st["launches"] >= 0 and st["kernel_ms_total"] >= 0.0
            elif op == "timing":
                ctx.set_timing(bool(rng.integers(2)))
            elif op == "sync":
                ctx.synchronize()
                drain()
        drain()
    finally:
        torch.cuda.synchronize()
        ctx.close()
