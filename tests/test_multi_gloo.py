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
