"""Multi-GPU image assembly: one process per GPU, tile-interleaved row partition, ONE gather.

The path shards naturally (pixels are independent; the RNG stream is keyed by absolute
(pixel, sample), so the assembled image is bit-identical for any GPU count).  Each rank renders
the row tiles t with t % world == rank into a compact buffer; rank 0 receives all buffers with a
single `torch.distributed.gather` (RCCL over xGMI on GPUs, gloo in the CPU tests) and
de-interleaves them into the final frame.  torch is plumbing here (device memory, streams,
the collective); every pixel is produced behind the C ABI.
"""
from __future__ import annotations

import copy
from typing import Callable, List, Optional

import numpy as np

from . import _abi
from .context import Context, params_out_row_index, params_out_rows

DEFAULT_TILE_ROWS = 4


def part_params(base: _abi.MirtParams, rank: int, world: int, tile_rows: int = DEFAULT_TILE_ROWS) -> _abi.MirtParams:
    """The MirtParams of rank `rank` of `world`: same image, its share of the row tiles."""
    p = copy.copy(base)
    if world > 1:
        p.tile_rows, p.n_parts, p.part = tile_rows, world, rank
    else:
        p.tile_rows, p.n_parts, p.part = 0, 0, 0
    return p


def max_part_rows(base: _abi.MirtParams, world: int, tile_rows: int = DEFAULT_TILE_ROWS) -> int:
    return max(params_out_rows(part_params(base, r, world, tile_rows)) for r in range(world))


def band_rows(base: _abi.MirtParams) -> int:
    end = base.row_end if base.row_end else base.height
    return end - base.row_begin


def gather_parts(local, rank: int, world: int, dst: int = 0, out=None):
    """ONE collective: every rank's padded part -> [world, max_rows, W, 4] on `dst` (None elsewhere)."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return local.unsqueeze(0)
    if rank == dst:
        if out is None:
            out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(local, list(out.unbind(0)), dst=dst)
        return out
    dist.gather(local, None, dst=dst)
    return None


def assemble_host(parts: np.ndarray, base: _abi.MirtParams, world: int, tile_rows: int = DEFAULT_TILE_ROWS) -> np.ndarray:
    """Host-side de-interleave (CPU tests / host consumers): parts [world, max_rows, W, 4] -> band image."""
    rb = base.row_begin
    out = np.zeros((band_rows(base), base.width, 4), dtype=np.uint8)
    for r in range(world):
        p = part_params(base, r, world, tile_rows)
        for i in range(params_out_rows(p)):
            out[params_out_row_index(p, i) - rb] = parts[r, i]
    return out


class TiledFrame:
    """Per-rank state of the multi-GPU render of one frame on GPUs.

    `pipelined=True` (world > 1) double-buffers the part buffers and issues the gather with `async_op=True`:
    the gather of frame i travels while frame i+1 renders, and rank 0 de-interleaves frame i after that.  Ranks
    then no longer meet once per frame; `flush()` completes the frames still in flight.

    `frames_in_flight=2` (pipelined, on a GPU) also RENDERS consecutive frames on the context's two frame streams
    (`mirt_ctx_frame_stream`: different hardware queues): a rank's share of a frame is a short launch -- 3 ms of config 3 at
    N = 8 -- whose ramp and tail then overlap the neighbouring frames' (one rank's share on one GPU: 89.5 -> 95.4 % of the ideal
    1/8 of the whole frame's time).  Buffer b's render, gather and de-interleave all live on frame stream b; an
    event chain keeps rank 0's de-interleaves in frame order."""

    def __init__(self, ctx: Context, base: _abi.MirtParams, rank: int, world: int, tile_rows: int = DEFAULT_TILE_ROWS,
                 pipelined: bool = False, _rehearse_single_rank: bool = False, device=None, frames_in_flight: int = 1):
        """`ctx` renders (`render_device`) and de-interleaves (`deinterleave_device`) into the buffers allocated
        here.  `device` defaults to the context's GPU; the CPU rehearsals of the N > 1 path (gloo; tests and
        `bench.py --dry-run`) pass torch.device("cpu") together with a stand-in context."""
        import torch

        self.ctx, self.base, self.rank, self.world, self.tile_rows = ctx, base, rank, world, tile_rows
        self.params = part_params(base, rank, world, tile_rows)
        self.rows = params_out_rows(self.params)
        self.max_rows = max_part_rows(base, world, tile_rows)
        self._rehearse = _rehearse_single_rank and world == 1       # drive the collective path with a 1-rank group
        self.pipelined = pipelined and (world > 1 or self._rehearse)
        dev = torch.device("cuda", ctx.device) if device is None else torch.device(device)
        self.device = dev
        n_buf = 2 if self.pipelined else 1
        self.locals = [torch.zeros((self.max_rows, base.width, 4), dtype=torch.uint8, device=dev) for _ in range(n_buf)]
        self.local = self.locals[0]
        self.frame = (torch.zeros((band_rows(base), base.width, 4), dtype=torch.uint8, device=dev)
                      if rank == 0 else None)
        collective = world > 1 or self._rehearse
        self.parts_bufs = ([torch.zeros((world, self.max_rows, base.width, 4), dtype=torch.uint8, device=dev) for _ in range(n_buf)]
                           if rank == 0 and collective else None)
        self.parts = self.parts_bufs[0] if self.parts_bufs else None
        self._pending = [None] * n_buf
        self._k = 0
        # two frames in flight: torch views of the context's frame streams (the collective and its wait() act on torch's CURRENT stream)
        self._frame_streams = None
        self._assembled = None                            # event after rank 0's latest de-interleave
        if frames_in_flight == 2 and self.pipelined and dev.type == "cuda":
            self._frame_streams = [torch.cuda.ExternalStream(ctx.frame_stream(i), device=dev) for i in range(2)]

    def _assemble(self, parts, stream):
        if self._rehearse:
            self.frame.copy_(parts[0][: self.frame.shape[0]])
            return
        self.ctx.deinterleave_device(part_params(self.base, 0, self.world, self.tile_rows), parts.data_ptr(),
                                     self.local.numel(), self.frame.data_ptr(), self.frame.numel(), stream)

    def _stream(self) -> int:
        import torch
        return torch.cuda.current_stream().cuda_stream if self.device.type == "cuda" else 0

    def _retire(self, b: int, stream) -> None:
        """Frame in buffer b: its gather has completed (stream-level wait); rank 0 assembles it."""
        self._pending[b].wait()
        self._pending[b] = None
        if self.rank == 0:
            self._assemble(self.parts_bufs[b], stream)

    def _retire_on_frame_stream(self, b: int) -> None:
        """Two frames in flight: the same on frame stream b, after the previous frame's de-interleave (they write one frame buffer)."""
        import torch
        fs = self._frame_streams[b]
        with torch.cuda.stream(fs):
            self._pending[b].wait()
            self._pending[b] = None
            if self.rank == 0:
                if self._assembled is not None:
                    fs.wait_event(self._assembled)
                self._assemble(self.parts_bufs[b], fs.cuda_stream)
                self._assembled = torch.cuda.Event()
                self._assembled.record(fs)

    def _step_two_in_flight(self):
        import torch
        import torch.distributed as dist
        b = self._k & 1
        fs = self._frame_streams[b]
        if self._k < 2:                                    # the caller's stream may still be filling what this frame reads or writes
            fs.wait_stream(torch.cuda.current_stream())
        if self._pending[b] is not None:                   # frame k-2 (normally retired one step ago): its gather read locals[b]
            self._retire_on_frame_stream(b)
        with torch.cuda.stream(fs):
            self.ctx.render_device(self.params, self.locals[b].data_ptr(), self.rows * self.base.width * 4, fs.cuda_stream)
            gather_list = list(self.parts_bufs[b].unbind(0)) if self.rank == 0 else None
            self._pending[b] = dist.gather(self.locals[b], gather_list, dst=0, async_op=True)
        if self._pending[b ^ 1] is not None:               # frame k-1: its gather had a whole render to complete
            self._retire_on_frame_stream(b ^ 1)
        self._k += 1
        return self.frame

    def step(self):
        """Render this rank's tiles, gather to rank 0, assemble.  Returns the frame tensor on rank 0
        (pipelined: the frame of the PREVIOUS step; call flush() after the last one)."""
        import torch.distributed as dist

        stream = self._stream()
        if self.world == 1 and not self._rehearse:
            self.ctx.render_device(self.params, self.frame.data_ptr(), self.frame.numel(), stream)
            return self.frame
        if not self.pipelined:
            self.ctx.render_device(self.params, self.local.data_ptr(), self.rows * self.base.width * 4, stream)
            if self._rehearse:
                dist.gather(self.local, list(self.parts.unbind(0)), dst=0)
                parts = self.parts
            else:
                parts = gather_parts(self.local, self.rank, self.world, dst=0, out=self.parts)
            if self.rank == 0:
                self._assemble(parts, stream)
            return self.frame
        if self._frame_streams is not None:
            return self._step_two_in_flight()
        b = self._k & 1
        if self._pending[b] is not None:                   # frame k-2 (normally retired one step ago)
            self._retire(b, stream)
        self.ctx.render_device(self.params, self.locals[b].data_ptr(), self.rows * self.base.width * 4, stream)
        gather_list = list(self.parts_bufs[b].unbind(0)) if self.rank == 0 else None
        self._pending[b] = dist.gather(self.locals[b], gather_list, dst=0, async_op=True)
        if self._pending[b ^ 1] is not None:               # frame k-1: its gather had a whole render to complete
            self._retire(b ^ 1, stream)
        self._k += 1
        return self.frame

    def flush(self):
        """Complete the frames still in flight (pipelined mode); older first."""
        stream = self._stream()
        if self.pipelined:
            for b in ((self._k & 1), (self._k & 1) ^ 1):
                if self._pending[b] is not None:
                    if self._frame_streams is not None:
                        self._retire_on_frame_stream(b)
                    else:
                        self._retire(b, stream)
            if self._frame_streams is not None:            # what follows on the caller's stream sees the finished frame
                import torch
                for fs in self._frame_streams:
                    torch.cuda.current_stream().wait_stream(fs)
        return self.frame
