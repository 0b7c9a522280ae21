"""TEST RIG (lives under tests/, not in the product package): the split plan of the multi-GPU path, as seen from Python --
a caller of the C ABI's planner.

The product's screen split lives below the C boundary (`rto_comm_*`, csrc/rto_comm.inc); everything its ranks must agree
on -- who renders, which part a rank owns, the column window of every frame that travels, offsets and float counts -- is
derived by ONE exported function, `rto_split_plan_make` (csrc/rto_split.inc, pure host arithmetic).  This module binds
it and adds what a rehearsal WITHOUT GPUs needs on top: numpy statements of the two data-movement kernels
(`k_pack_columns`, `k_assemble_shade_crop`) and the send/recv pattern of `comm_exchange` over any torch.distributed
process group (gloo on CPU).  Nothing here decides anything about the split: every index comes out of the C functions,
so `tests/_tilesplit_worker.py` (world 2, 3, 4, 5, 8 over gloo) exercises the planner the GPUs use.

No reference counterpart: the reference is single-GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ray_tracing_octrees_amd import hip

F = np.float32


def scene_bounds_of_nodes(nodes: np.ndarray, grid_min, voxel_size) -> hip.SceneBounds:
    """rto_scene_bounds_of_nodes: the bounds rto_upload_octree derives from a GPUNodes array (no GPU needed)."""
    nodes = np.ascontiguousarray(nodes)
    gm = (C.c_float * 3)(*[float(F(x)) for x in grid_min])
    b = hip.SceneBounds()
    rc = hip.load().rto_scene_bounds_of_nodes(nodes.ctypes.data, len(nodes), gm, float(F(voxel_size)), C.byref(b))
    if rc != hip.RTO_OK:
        raise hip.RtoError(rc, "rto_scene_bounds_of_nodes")
    return b


def make_plan(bounds: hip.SceneBounds, frames, world: int, band_rows: int = 16) -> hip.SplitPlan:
    """rto_split_plan_make for a batch of frames (a list of hip.Frame)."""
    arr = hip.Context.frame_array(list(frames))
    plan = hip.SplitPlan()
    rc = hip.load().rto_split_plan_make(C.byref(bounds), arr, len(arr), world, band_rows, C.byref(plan))
    if rc != hip.RTO_OK:
        raise hip.RtoError(rc, "rto_split_plan_make: frames of one batch share a positive width / height; band_rows % 8 == 0")
    return plan


def plan_bytes(plan: hip.SplitPlan) -> bytes:
    """The plan as the bytes every rank must hold identically."""
    return bytes(C.string_at(C.addressof(plan), C.sizeof(plan)))


def part_of_rank(plan: hip.SplitPlan, rank: int) -> int:
    return hip.load().rto_split_part_of_rank(C.byref(plan), rank)


def rows_of_part(plan: hip.SplitPlan, part: int) -> int:
    return hip.load().rto_split_rows_of_part(C.byref(plan), part)


def row_sources(plan: hip.SplitPlan):
    """(part, local_row) of every row of an assembled frame, from rto_split_row_source."""
    L = hip.load()
    part = np.empty(plan.height, np.int64)
    local = np.empty(plan.height, np.int64)
    p, r = C.c_int(), C.c_int()
    for y in range(plan.height):
        rc = L.rto_split_row_source(C.byref(plan), y, C.byref(p), C.byref(r))
        if rc != hip.RTO_OK:
            raise hip.RtoError(rc, "rto_split_row_source")
        part[y], local[y] = p.value, r.value
    return part, local


def row_map(plan: hip.SplitPlan, part: int) -> np.ndarray:
    """Global row of every local row of `part`, in compact-buffer order (the inverse of row_sources)."""
    src_part, src_local = row_sources(plan)
    rows = np.nonzero(src_part == part)[0]
    out = np.empty(len(rows), np.int64)
    out[src_local[rows]] = rows
    return out


# ---- numpy statements of the two data-movement kernels (host-staged rehearsal only) ---------------------------------
def pack_columns(plan: hip.SplitPlan, local: np.ndarray) -> np.ndarray:
    """k_pack_columns: local is [n_frames][rows_part0][width] float32 -> the packed part (pack_floats floats)."""
    if not plan.cropped:
        return np.ascontiguousarray(local, F).reshape(-1)
    out = np.zeros(plan.pack_floats, F)
    for i in range(plan.n_frames):
        x0, w, off = plan.win_x0[i], plan.win_w[i], plan.win_off[i]
        out[off: off + plan.rows_part0 * w] = local[i][:, x0: x0 + w].reshape(-1)
    return out


def shade_color(s: np.ndarray) -> np.ndarray:
    """shade_color() of the kernels in float32, one rounding per operation: vec3(1, .8, .6) * term + .1, miss -> black."""
    s = s.astype(F)
    px = np.empty(s.shape + (4,), F)
    px[..., 0] = F(1.0) * s + F(0.1)
    px[..., 1] = F(0.8) * s + F(0.1)
    px[..., 2] = F(0.6) * s + F(0.1)
    px[..., 3] = F(1.0)
    px[s < 0] = (0.0, 0.0, 0.0, 1.0)
    return px


def assemble(plan: hip.SplitPlan, gathered: np.ndarray) -> np.ndarray:
    """k_assemble_shade_crop / k_assemble_shade: gathered is [render_parts][pack_floats] float32 as rank 0 receives it ->
    [n_frames][height][width][4] RGBA32F."""
    src_part, src_local = row_sources(plan)
    W, H = plan.width, plan.height
    out = np.empty((plan.n_frames, H, W, 4), F)
    for i in range(plan.n_frames):
        shade = np.full((H, W), F(-1.0), F)                      # outside the window: background (kShadeMiss)
        if plan.cropped:
            x0, w, off = plan.win_x0[i], plan.win_w[i], plan.win_off[i]
            for y in range(H):
                a = off + src_local[y] * w
                shade[y, x0: x0 + w] = gathered[src_part[y], a: a + w]
        else:
            for y in range(H):
                a = i * plan.frame_floats + src_local[y] * W
                shade[y] = gathered[src_part[y], a: a + W]
        out[i] = shade_color(shade)
    return out


def exchange(plan: hip.SplitPlan, rank: int, packed, group=None):
    """comm_exchange over a torch.distributed group, staged through host tensors: every rendering rank sends pack_floats
    floats to rank 0; rank 0 receives part p at offset p * pack_floats.  Returns [render_parts][pack_floats] on rank 0."""
    import torch
    import torch.distributed as dist

    count = int(plan.pack_floats)
    mine = part_of_rank(plan, rank)
    gathered = None
    reqs = []
    if rank == 0:
        gathered = torch.zeros((plan.render_parts, count), dtype=torch.float32)
        for r in range(plan.world):
            p = part_of_rank(plan, r)
            if p < 0:
                continue                                          # the gatherer itself ships nothing
            if r == 0:
                gathered[p].copy_(torch.from_numpy(np.ascontiguousarray(packed, F)))
            else:
                reqs.append(dist.irecv(gathered[p], src=r, group=group))
    elif mine >= 0:
        t = torch.from_numpy(np.ascontiguousarray(packed, F))
        assert t.numel() == count, "a rank ships exactly the plan's pack_floats"
        reqs.append(dist.isend(t, dst=0, group=group))
    for q in reqs:
        q.wait()
    return None if gathered is None else gathered.numpy()
