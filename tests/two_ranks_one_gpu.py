"""Two REAL ranks -- two processes, real HIP kernels, real streams -- through the pipelined TiledFrame on ONE GPU.

RCCL refuses two ranks on one device, so the collective here is gloo's gather on CUDA tensors (checked to work in this image); what
is exercised is everything around it with world == 2: every rank's tile partition rendered by the HIP kernels, double-buffered part
buffers, the asynchronous gather of frame k travelling while frame k+1 renders, retire order, de-interleave on the device, flush.
Every frame carries a different seed, so a frame assembled from the wrong buffer cannot pass.  Run by tests/test_gpu_multi.py as a
child process (it spawns its ranks before it has touched the GPU); prints one line, "N ranks ok: ...", on success.

    python tests/two_ranks_one_gpu.py [frames] [ranks] [frames in flight: 1 or 2]"""
import os
import socket
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def worker(rank: int, world: int, port: int, frames: int, in_flight: int) -> None:
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import weekend_raytracer_wgpu_amd as m
    from helpers import scene_data

    w, h, spp = 320, 182, 48                       # 182 rows: the last 4-row tile is ragged; 48 spp: the pooled kernel, dispensed strips
    sd = scene_data("three_spheres", w, h)
    ctx = m.Context(0)
    ctx.set_scene(sd)
    base = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8)
    torch.cuda.set_stream(torch.cuda.Stream())
    frame = m.multi_gpu.TiledFrame(ctx, base, rank, world, tile_rows=4, pipelined=True, frames_in_flight=in_flight)
    got = []
    for k in range(frames):
        frame.params.seed = 100 + k                # the frame index travels in the seed: frames differ
        f = frame.step()                           # pipelined: rank 0 gets the frame of step k - 1
        if rank == 0 and k >= 1:
            torch.cuda.synchronize()
            got.append(f.cpu().numpy().copy())
    f = frame.flush()
    if rank == 0:
        torch.cuda.synchronize()
        got.append(f.cpu().numpy().copy())
        alone = m.Context(0)                       # the same frames from ONE rank, whole-frame path
        alone.set_scene(sd)
        bad = []
        for k in range(frames):
            p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8, seed=100 + k)
            if not np.array_equal(got[k], alone.render(p)):
                bad.append(k)
        alone.close()
        assert len(got) == frames and not bad, f"frames {bad} of {frames} differ from the single-rank render"
        assert any(not np.array_equal(got[0], g) for g in got[1:]), "the frames must differ from each other"
        print(f"{world} ranks ok: {frames} pipelined frames of {w}x{h}x{spp} spp ({in_flight} in flight), each equal to the single-rank render", flush=True)
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2          # at most 4: the GPU box allows 6 processes on its card
    in_flight = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # 2: consecutive frames render on the context's two frame streams
    mp.spawn(worker, args=(world, port, frames, in_flight), nprocs=world, join=True)
