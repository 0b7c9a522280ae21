"""Screen-space split of one frame across the GPUs of a node + ONE gather for the final image.

No reference counterpart (the reference is single-GPU).  Every rank holds the whole
octree (<= 90 MB) and renders the bands `b % world == rank` of the image (bands of `band_rows` rows,
round-robin so that the lit centre rows are spread over all ranks) into a compact buffer; a single
`torch.distributed.gather` to rank 0 (RCCL over xGMI: 7 point-to-point links into the root) delivers the
buffers, and rank 0 re-interleaves them with one kernel.

Two things keep the xGMI links and the root GPU off the critical path:
  payload   "shade" (default): a part ships ONE float per pixel -- the Lambert term of the hit, -1 for a miss
            (rto_render_shade_device) -- and rank 0 finishes the colour expression while it re-interleaves
            (rto_assemble_shade_device): W*H*4/N bytes per link instead of W*H*16/N, bit-identical frame.
            "rgba": the plain RGBA32F pixels (rto_render_device / rto_assemble_device).
  pipeline  submit()/flush(): the gather of frame k runs on RCCL's stream while this rank already renders its
            part of frame k+1 (double-buffered local buffers); render() is the one-frame-at-a-time form.

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

    def __init__(self, ctx: hip.Context, triangles: bool = False, shadow: bool = True):
        """triangles=True renders BASELINE config 5's path (leaf triangles + `shadow` ray; upload or build the triangle
        buffer on every rank's context first) instead of the solid-leaf octree path."""
        import torch

        self.torch = torch
        self.ctx = ctx
        self.device = torch.device("cuda", ctx.device)
        self.triangles = triangles
        self.shadow = shadow
        self._bound = 0

    def bind_stream(self):
        """Look torch's current stream up once per batch (the lookup costs about as much as a kernel launch)."""
        self._bound = self.torch.cuda.current_stream(self.device).cuda_stream

    def _stream(self) -> int:
        return self._bound

    def empty(self, shape):
        return self.torch.empty(shape, dtype=self.torch.float32, device=self.device)

    def render_part(self, frame: hip.Frame, part: hip.Partition | None, out, payload: str = "rgba"):
        if self.triangles:
            if payload == "shade":
                self.ctx.render_triangles_shade_device(frame, out.data_ptr(), self.shadow, part, self._stream())
            else:
                self.ctx.render_triangles_device(frame, out.data_ptr(), self.shadow, part, self._stream())
        elif payload == "shade":
            self.ctx.render_shade_device(frame, out.data_ptr(), part, self._stream())
        else:
            self.ctx.render_device(frame, out.data_ptr(), part, self._stream())

    def render_parts(self, frames, part: hip.Partition, local, payload: str):
        """All frames of a batch with one call through the C ABI (octree path): local is [batch][rows][width(,4)]."""
        if self.triangles:
            for f, frame in enumerate(frames):
                self.render_part(frame, part, local[f], payload)
            return
        arr = hip.Context.frame_array(frames)
        self.ctx.render_batch_device(arr, local.data_ptr(), local.stride(0) * 4, part, payload == "shade", self._stream())

    def assemble_all(self, frames, part0: hip.Partition, gathered, out_frames, payload: str):
        arr = hip.Context.frame_array(frames)
        self.ctx.assemble_batch_all_device(arr, part0, gathered.data_ptr(), payload == "shade", out_frames.data_ptr(),
                                           out_frames.stride(0) * 4, self._stream())

    def assemble(self, frame: hip.Frame, part0: hip.Partition, gathered, out, payload: str = "rgba", batch: int = 1, index: int = 0):
        """Frame `index` of a gather that carried `batch` frames per rank: gathered is [rank][batch][rows][width(,4)]."""
        self.ctx.assemble_batch_device(frame, part0, gathered.data_ptr(), batch, index, payload == "shade", out.data_ptr(), self._stream())


@dataclass
class _Buffers:
    key: tuple
    local: list          # two compact part buffers [batch][rows][width(,4)] (double-buffered for the pipelined form)
    gathered: object     # rank 0: world x that, as the gather delivers them
    frames: object       # rank 0: the assembled RGBA32F frames [batch][H][W][4]


@dataclass
class _InFlight:
    work: object         # the gather's Work handle
    frames: list         # the hip.Frame of every image in the batch
    keep: object         # tensors that must outlive the gather


class TileSplitRenderer:
    """Renders frames cooperatively; rank 0 gets the (H, W, 4) images, the others None.

    render(frame)               one frame at a time (everything stream-ordered, no host sync)
    submit(frame) / flush()     pipelined: submit(k) returns the image of frame k-1 (None for the first call),
                                flush() the last one; a returned tensor is overwritten by the next assemble
    submit_batch(frames) / flush_batch() / render_batch(frames)
                                the same with SEVERAL consecutive frames per collective: every rank renders its part of
                                each frame of the batch, ONE gather ships them all ([rank][batch][rows][width]), rank 0
                                assembles each.  Fewer, larger collectives: the host cost of a torch.distributed call
                                (tens of microseconds, comparable to a whole 1080p frame) is paid once per batch.

    `stage_through_host=True` moves the gather payload through CPU tensors: only for rehearsing the multi-rank
    path with the gloo backend (e.g. several ranks sharing one GPU); the product path gathers device to device.
    `force_collective=True` walks the multi-rank path with a one-rank group (exercising it on a single GPU)."""

    def __init__(self, backend, rank: int, world_size: int, band_rows: int = 16, group=None, stage_through_host: bool = False,
                 payload: str = "shade", force_collective: bool = False):
        if band_rows <= 0 or band_rows % 8:
            raise ValueError("band_rows must be a positive multiple of 8")
        if payload not in ("shade", "rgba"):
            raise ValueError("payload must be 'shade' or 'rgba'")
        self.backend = backend
        self.rank = rank
        self.world = world_size
        self.band_rows = band_rows
        self.group = group
        self.stage_through_host = stage_through_host
        self.payload = payload
        self.single = world_size == 1 and not force_collective     # then frames are rendered straight into place
        self._buf: _Buffers | None = None
        self._seq = 0
        self._inflight: _InFlight | None = None
        self._part = hip.Partition(self.world, self.rank, self.band_rows)
        self._part0 = hip.Partition(self.world, 0, self.band_rows)

    def partition(self, part: int | None = None) -> hip.Partition:
        return hip.Partition(self.world, self.rank if part is None else part, self.band_rows)

    def _part_shape(self, batch: int, rows: int, width: int) -> tuple:
        return (batch, rows, width) if self.payload == "shade" else (batch, rows, width, 4)

    def _buffers(self, frame: hip.Frame, batch: int) -> _Buffers:
        key = (frame.width, frame.height, batch)
        if self._buf is None or self._buf.key != key:
            if self._inflight is not None:
                raise RuntimeError("frame size or batch size changed with frames in flight: call flush() first")
            gathered = frms = None
            local = []
            if self.rank == 0:
                frms = self.backend.empty((batch, frame.height, frame.width, 4))
            if not self.single:
                rows0 = partition_rows(frame.height, self.world, 0, self.band_rows)   # part 0 owns the most rows
                local = [self.backend.empty(self._part_shape(batch, rows0, frame.width)) for _ in range(2)]
                if self.rank == 0:
                    gathered = self.backend.empty((self.world,) + self._part_shape(batch, rows0, frame.width))
                    self._gather_list = [gathered[i] for i in range(self.world)]
            self._buf = _Buffers(key, local, gathered, frms)
        return self._buf

    # ---- the three steps of a batch ---------------------------------------------------------------
    def _issue_gather(self, b: _Buffers, local, frames) -> _InFlight:
        import torch.distributed as dist

        if self.stage_through_host:
            src = local.cpu()
            parts = [src.new_empty(src.shape) for _ in range(self.world)] if self.rank == 0 else None
            work = dist.gather(src, parts, dst=0, group=self.group, async_op=True)
            return _InFlight(work, frames, (src, parts))
        parts = self._gather_list if self.rank == 0 else None
        work = dist.gather(local, parts, dst=0, group=self.group, async_op=True)
        return _InFlight(work, frames, (local, parts))

    def _complete(self, b: _Buffers, fl: _InFlight):
        fl.work.wait()          # RCCL: the current stream waits for the gather; gloo: the host does
        if self.rank != 0:
            return None
        if self.stage_through_host:
            for i, p in enumerate(fl.keep[1]):
                b.gathered[i].copy_(p)
        n = len(fl.frames)
        if hasattr(self.backend, "assemble_all"):
            self.backend.assemble_all(fl.frames, self._part0, b.gathered, b.frames, self.payload)
        else:
            for f, frame in enumerate(fl.frames):
                self.backend.assemble(frame, self._part0, b.gathered, b.frames[f], self.payload, batch=b.key[2], index=f)
        return [b.frames[f] for f in range(n)]

    # ---- public: batches ----------------------------------------------------------------------------
    def submit_batch(self, frames):
        """Render this rank's part of every frame, complete the previous batch (its gather ran meanwhile), start this
        batch's gather.  Returns the previous batch's images on rank 0 (None on the first call / other ranks).  Every
        call must carry the same number of frames until flush_batch()."""
        frames = list(frames)
        if not frames:
            raise ValueError("submit_batch: no frames")
        b = self._buffers(frames[0], len(frames))
        if hasattr(self.backend, "bind_stream"):
            self.backend.bind_stream()
        if self.single:
            for f, frame in enumerate(frames):
                self.backend.render_part(frame, None, b.frames[f], "rgba")
            self._single_last = len(frames)
            return [b.frames[f] for f in range(len(frames))]
        local = b.local[self._seq % 2]
        self._seq += 1
        if hasattr(self.backend, "render_parts"):               # overlaps the gather in flight
            self.backend.render_parts(frames, self._part, local, self.payload)
        else:
            for f, frame in enumerate(frames):
                self.backend.render_part(frame, self._part, local[f], self.payload)
        done = None
        if self._inflight is not None:
            done = self._complete(b, self._inflight)            # ... and only now waits for it
            self._inflight = None
        self._inflight = self._issue_gather(b, local, frames)
        return done

    def flush_batch(self):
        if self.single:
            if self._buf is None:
                return None
            return [self._buf.frames[f] for f in range(getattr(self, "_single_last", 1))]
        if self._inflight is None:
            return None
        fl, self._inflight = self._inflight, None
        if hasattr(self.backend, "bind_stream"):
            self.backend.bind_stream()
        return self._complete(self._buf, fl)

    def render_batch(self, frames):
        if self._inflight is not None:
            raise RuntimeError("render with pipelined frames in flight: call flush() first")
        self.submit_batch(frames)
        return self.flush_batch()

    # ---- public: one frame per collective -------------------------------------------------------------
    def submit(self, frame: hip.Frame):
        out = self.submit_batch([frame])
        return None if out is None else out[0]

    def flush(self):
        out = self.flush_batch()
        return None if out is None else out[-1]

    def render(self, frame: hip.Frame):
        if self._inflight is not None:
            raise RuntimeError("render() with a pipelined frame in flight: call flush() first")
        self.submit(frame)
        return self.flush()
