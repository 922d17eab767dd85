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
    """Per-rank state of the multi-GPU render of one frame on GPUs."""

    def __init__(self, ctx: Context, base: _abi.MirtParams, rank: int, world: int, tile_rows: int = DEFAULT_TILE_ROWS):
        import torch

        self.ctx, self.base, self.rank, self.world, self.tile_rows = ctx, base, rank, world, tile_rows
        self.params = part_params(base, rank, world, tile_rows)
        self.rows = params_out_rows(self.params)
        self.max_rows = max_part_rows(base, world, tile_rows)
        dev = torch.device("cuda", ctx.device)
        self.local = torch.zeros((self.max_rows, base.width, 4), dtype=torch.uint8, device=dev)
        self.frame = (torch.zeros((band_rows(base), base.width, 4), dtype=torch.uint8, device=dev)
                      if rank == 0 else None)
        self.parts = (torch.zeros((world, self.max_rows, base.width, 4), dtype=torch.uint8, device=dev)
                      if rank == 0 and world > 1 else None)

    def step(self):
        """Render this rank's tiles, gather to rank 0, assemble.  Returns the frame tensor on rank 0."""
        import torch

        stream = torch.cuda.current_stream().cuda_stream
        if self.world == 1:
            self.ctx.render_device(self.params, self.frame.data_ptr(), self.frame.numel(), stream)
            return self.frame
        self.ctx.render_device(self.params, self.local.data_ptr(), self.rows * self.base.width * 4, stream)
        parts = gather_parts(self.local, self.rank, self.world, dst=0, out=self.parts)
        if self.rank == 0:
            self.ctx.deinterleave_device(part_params(self.base, 0, self.world, self.tile_rows), parts.data_ptr(),
                                         self.local.numel(), self.frame.data_ptr(), self.frame.numel(), stream)
        return self.frame
