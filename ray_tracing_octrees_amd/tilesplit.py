"""Screen-space split of one frame across the GPUs of a node + ONE gather for the final image.

No reference counterpart (the reference is single-GPU).  Every rank holds the whole
octree (<= 90 MB) and renders the bands `b % world == rank` of the image (bands of `band_rows` rows,
round-robin so that the lit centre rows are spread over all ranks) into a compact buffer; a single
`torch.distributed.gather` to rank 0 (RCCL over xGMI: 7 point-to-point links into the root, W*H*16/N bytes
each) delivers the buffers, and rank 0 re-interleaves them with one copy kernel (rto_assemble_device).

The class is backend-agnostic so the rank/partition/gather logic can be exercised with gloo on CPU:
  HipBackend     device buffers + the C ABI (the product path)
  any object with the same three methods (tests supply a CPU stand-in)
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import hip


def partition_rows(height: int, num_parts: int, part: int, band_rows: int) -> int:
    """Rows owned by `part` (same arithmetic as rto_partition_rows)."""
    if num_parts <= 1:
        return height
    bands = (height + band_rows - 1) // band_rows
    rows = 0
    for b in range(part, bands, num_parts):
        rows += min((b + 1) * band_rows, height) - b * band_rows
    return rows


def partition_row_map(height: int, num_parts: int, part: int, band_rows: int) -> np.ndarray:
    """Global row index of every local row of `part`, in compact-buffer order."""
    if num_parts <= 1:
        return np.arange(height)
    bands = (height + band_rows - 1) // band_rows
    rows = []
    for b in range(part, bands, num_parts):
        rows.extend(range(b * band_rows, min((b + 1) * band_rows, height)))
    return np.asarray(rows, dtype=np.int64)


class HipBackend:
    """Product backend: torch CUDA(HIP) tensors as device buffers, kernels through the C ABI on torch's
    current stream (so RCCL collectives issued by torch order themselves behind the render)."""

    def __init__(self, ctx: hip.Context):
        import torch

        self.torch = torch
        self.ctx = ctx
        self.device = torch.device("cuda", ctx.device)

    def empty(self, shape):
        return self.torch.empty(shape, dtype=self.torch.float32, device=self.device)

    def render_part(self, frame: hip.Frame, part: hip.Partition | None, out):
        self.ctx.render_device(frame, out.data_ptr(), part, self.torch.cuda.current_stream(self.device).cuda_stream)

    def assemble(self, frame: hip.Frame, part0: hip.Partition, gathered, out):
        self.ctx.assemble_device(frame, part0, gathered.data_ptr(), out.data_ptr(),
                                 self.torch.cuda.current_stream(self.device).cuda_stream)


@dataclass
class _Buffers:
    key: tuple
    local: object
    gathered: object
    frame: object


class TileSplitRenderer:
    """renders `frame` cooperatively; rank 0 gets the (H, W, 4) image, the others None.

    `stage_through_host=True` moves the gather payload through CPU tensors: only for rehearsing the multi-rank
    path with the gloo backend (e.g. several ranks sharing one GPU); the product path gathers device to device."""

    def __init__(self, backend, rank: int, world_size: int, band_rows: int = 16, group=None, stage_through_host: bool = False):
        if band_rows <= 0 or band_rows % 8:
            raise ValueError("band_rows must be a positive multiple of 8")
        self.backend = backend
        self.rank = rank
        self.world = world_size
        self.band_rows = band_rows
        self.group = group
        self.stage_through_host = stage_through_host
        self._buf: _Buffers | None = None

    def partition(self, part: int | None = None) -> hip.Partition:
        return hip.Partition(self.world, self.rank if part is None else part, self.band_rows)

    def _buffers(self, frame: hip.Frame) -> _Buffers:
        key = (frame.width, frame.height)
        if self._buf is None or self._buf.key != key:
            rows0 = partition_rows(frame.height, self.world, 0, self.band_rows)   # part 0 owns the most rows
            local = self.backend.empty((rows0, frame.width, 4))
            gathered = frm = None
            if self.rank == 0:
                frm = self.backend.empty((frame.height, frame.width, 4))
                gathered = self.backend.empty((self.world, rows0, frame.width, 4)) if self.world > 1 else None
            self._buf = _Buffers(key, local, gathered, frm)
        return self._buf

    def render(self, frame: hip.Frame):
        b = self._buffers(frame)
        if self.world == 1:
            self.backend.render_part(frame, None, b.frame)
            return b.frame
        import torch.distributed as dist

        self.backend.render_part(frame, self.partition(), b.local)
        if self.stage_through_host:
            local = b.local.cpu()
            if self.rank == 0:
                parts = [local.new_empty(local.shape) for _ in range(self.world)]
                dist.gather(local, parts, dst=0, group=self.group)
                for i, p in enumerate(parts):
                    b.gathered[i].copy_(p)
                self.backend.assemble(frame, self.partition(0), b.gathered, b.frame)
                return b.frame
            dist.gather(local, None, dst=0, group=self.group)
            return None
        if self.rank == 0:
            dist.gather(b.local, [b.gathered[i] for i in range(self.world)], dst=0, group=self.group)
            self.backend.assemble(frame, self.partition(0), b.gathered, b.frame)
            return b.frame
        dist.gather(b.local, None, dst=0, group=self.group)
        return None
