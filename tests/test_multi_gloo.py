"""N > 1 path on CPU: two processes over gloo run the SAME partition -> gather -> assemble code
bench.py uses on GPUs (weekend-raytracer-wgpu_amd/multi_gpu.py), with the oracle standing in for
the HIP kernel as the per-rank renderer, and must reproduce the single-process frame bit for bit."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, tile_rows: int, out_path: str) -> None:
    sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import weekend_raytracer_wgpu_amd as m
    import oracle_binding as ob
    from helpers import scene_data

    w, h, spp = 48, 37, 6
    sd = scene_data("three_spheres", w, h)
    base = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8)
    p = m.multi_gpu.part_params(base, rank, world, tile_rows)
    part = ob.render(sd, p, n_threads=1)                                  # stand-in for ctx.render_device
    local = torch.zeros((m.multi_gpu.max_part_rows(base, world, tile_rows), w, 4), dtype=torch.uint8)
    local[:part.shape[0]] = torch.from_numpy(part)
    parts = m.multi_gpu.gather_parts(local, rank, world, dst=0)           # the ONE collective
    if rank == 0:
        frame = m.multi_gpu.assemble_host(parts.numpy(), base, world, tile_rows)
        np.save(out_path, frame)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tile_rows", [4, 16])
def test_two_rank_gather_reassembles_the_frame(tmp_path, tile_rows, oracle):
    import weekend_raytracer_wgpu_amd as m
    from helpers import assert_images_equal, scene_data

    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), tile_rows, out), nprocs=2, join=True)
    w, h, spp = 48, 37, 6
    want = oracle.render(scene_data("three_spheres", w, h), m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8))
    assert_images_equal(np.load(out), want, "2-rank frame")


class _OracleContext:
    """Stand-in for `Context` in the pipelined TiledFrame: the oracle renders into the rank's CPU buffer and the
    host `assemble_host` de-interleaves — the collective code path (double-buffered parts, async gather, retire
    order, flush) is the product's."""

    def __init__(self, sd):
        self.sd, self.device = sd, 0

    @staticmethod
    def _view(ptr, shape):
        import ctypes
        n = int(np.prod(shape))
        return np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(ptr)).reshape(shape)

    def render_device(self, params, d_ptr, nbytes, stream=None):
        import oracle_binding as ob
        img = ob.render(self.sd, params, n_threads=1)
        assert nbytes >= img.nbytes
        self._view(d_ptr, img.shape)[:] = img

    def deinterleave_device(self, params, d_parts, part_stride, d_out, out_nbytes, stream=None):
        import weekend_raytracer_wgpu_amd as m
        world, w = params.n_parts, params.width
        parts = self._view(d_parts, (world, part_stride // (w * 4), w, 4))
        self._view(d_out, (m.multi_gpu.band_rows(params), w, 4))[:] = m.multi_gpu.assemble_host(parts, params, world, params.tile_rows)


def _pipelined_worker(rank: int, world: int, port: int, tile_rows: int, pipelined: bool, out_path: str) -> None:
    sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import weekend_raytracer_wgpu_amd as m
    from helpers import scene_data

    w, h = 40, 30
    sd = scene_data("three_spheres", w, h)
    frames = []
    # three DIFFERENT frames in a row (seed = frame number): the double buffering must not mix them up
    tf = None
    for k in range(3):
        base = m.make_params(w, h, 5, mode=m.MIRT_MODE_PT, num_bounces=8, seed=k + 1)
        if tf is None:
            tf = m.multi_gpu.TiledFrame(_OracleContext(sd), base, rank, world, tile_rows=tile_rows, pipelined=pipelined,
                                        device=torch.device("cpu"))
        tf.base, tf.params = base, m.multi_gpu.part_params(base, rank, world, tile_rows)
        out = tf.step()
        if rank == 0 and pipelined and k >= 1:
            frames.append(out.numpy().copy())              # pipelined: step k returns frame k-1
        elif rank == 0 and not pipelined:
            frames.append(out.numpy().copy())
    out = tf.flush()
    if rank == 0 and pipelined:
        frames.append(out.numpy().copy())
    if rank == 0:
        np.save(out_path, np.stack(frames))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("pipelined", [True, False])
def test_tiled_frame_pipelined_two_ranks(tmp_path, pipelined, oracle):
    """TiledFrame.step()/flush() as bench.py drives it for N > 1 (pipelined: async gather of frame k overlapping
    the render of frame k+1), two ranks over gloo: every assembled frame is bit-identical to the single-process one."""
    import weekend_raytracer_wgpu_amd as m
    from helpers import assert_images_equal, scene_data

    out = str(tmp_path / "frames.npy")
    mp.spawn(_pipelined_worker, args=(2, _free_port(), 4, pipelined, out), nprocs=2, join=True)
    got = np.load(out)
    assert got.shape[0] == 3
    w, h = 40, 30
    sd = scene_data("three_spheres", w, h)
    for k in range(3):
        want = oracle.render(sd, m.make_params(w, h, 5, mode=m.MIRT_MODE_PT, num_bounces=8, seed=k + 1))
        assert_images_equal(got[k], want, f"frame {k} pipelined={pipelined}")
