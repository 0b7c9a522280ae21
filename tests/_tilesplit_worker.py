"""Worker for test_tilesplit_gloo.py: run under torch.distributed.run with the gloo backend (CPU, no GPU).

What is under test is the PRODUCT's split arithmetic: every rank calls the C ABI's planner (rto_split_plan_make,
rto_split_part_of_rank, rto_split_row_source -- csrc/rto_split.inc, the functions rto_comm_* itself uses) on its own copy
of the frames and the scene bounds; the ranks compare their plans byte for byte, then ship exactly the plan's float
counts to rank 0 with the send/recv pattern of comm_exchange, and rank 0 re-interleaves with the plan's offsets.  The
assembled frames must be the oracle's, bit for bit.  The pixel producer is a CPU stand-in backed by the oracle (there is
no GPU here); packing and assembly are the numpy statements of k_pack_columns / k_assemble_shade_crop in
ray_tracing_octrees_amd.tilesplit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import orc  # noqa: E402
from ray_tracing_octrees_amd import hip  # noqa: E402
import tilesplit  # noqa: E402  (tests/tilesplit.py: test rig, beside this file)

F = np.float32


def encode_shade(rgba: np.ndarray) -> np.ndarray:
    """The oracle only produces pixels; recover a Lambert term that decodes to exactly those pixels: each channel
    fl(fl(k * s) + 0.1) is monotone in s >= 0, so the terms that decode to a given channel value form an interval of float
    bit patterns (found by bisection); any term in the intersection of the three intervals will do.  Test scaffolding for
    the CPU stand-in only."""
    shape = rgba.shape[:-1]
    px = np.ascontiguousarray(rgba, F).reshape(-1, 4)
    hit = ~((px[:, 0] == 0) & (px[:, 1] == 0) & (px[:, 2] == 0))
    out = np.full(len(px), -1.0, F)
    t = px[hit]
    lo_all = np.zeros(len(t), np.int64)
    hi_all = np.full(len(t), 0x40000000, np.int64)                 # bit pattern of 2.0f: terms lie in [0, 1]
    for c, k in enumerate((F(1.0), F(0.8), F(0.6))):
        def first_where(pred):                                        # smallest bit pattern b in [0, 2.0f] with pred(float(b))
            lo = np.zeros(len(t), np.int64)
            hi = np.full(len(t), 0x40000000, np.int64)
            while (lo < hi).any():
                mid = (lo + hi) // 2
                val = (k * mid.astype(np.uint32).view(F)).astype(F) + F(0.1)
                ok = pred(val)
                hi = np.where(ok & (lo < hi), mid, hi)
                lo = np.where(~ok & (lo < hi), mid + 1, lo)
            return lo
        lo_all = np.maximum(lo_all, first_where(lambda v: v >= t[:, c]))
        hi_all = np.minimum(hi_all, first_where(lambda v: v > t[:, c]))
    assert (lo_all < hi_all).all(), "stand-in could not encode a pixel"
    out[hit] = lo_all.astype(np.uint32).view(F)
    out = out.reshape(shape)
    assert (tilesplit.shade_color(out) == np.asarray(rgba, F)).all()
    return out


def render_part(scene, frames, plan, part):
    """[n_frames][rows_part0][width] Lambert terms of `part`: only its rows are traced (rows beyond its own stay 0: padding)."""
    nodes, gmin, voxel = scene
    rows = tilesplit.row_map(plan, part)
    local = np.zeros((plan.n_frames, plan.rows_part0, plan.width), F)
    for i, fr in enumerate(frames):
        view = np.array(list(fr.view), F)
        pos = np.array(list(fr.cam_pos), F)
        full = np.zeros((plan.height, plan.width, 4), F)
        for y in rows:
            orc.render(nodes, gmin, voxel, view, pos, fr.aspect, fr.fov_deg, plan.width, plan.height, rows=(int(y), int(y) + 1), out=full)
        local[i, : len(rows)] = encode_shade(full[rows])
    return local


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = orc.test_sphere_grid(16)
    nodes = orc.build_flat_octree(g)
    scene = (nodes, g.min, g.voxel_size)
    bounds = tilesplit.scene_bounds_of_nodes(nodes, g.min, g.voxel_size)
    ok = True
    why = []

    def check(cond, msg):
        nonlocal ok
        if not cond:
            ok = False
            why.append(msg)

    # (W, H, band rows, cameras of the batch).  Cameras: centred, sphere partly off-screen, looking away (token window),
    # close up (window = whole width); batches of 1, 2 and 4 frames; heights that are no multiple of the band.
    away = orc.Camera(0.5, 0.7, 1.8)
    away.set_target(40.0, 0.0, 40.0)
    side = orc.Camera(0.5, 0.7, 1.8)
    side.set_target(0.55, 0.1, 0.0)
    cases = (
        (64, 48, 8, [orc.Camera(0.5, 0.7, 1.8)]),
        (50, 37, 16, [orc.Camera(0.9, 0.6, 2.4), side]),
        (40, 100, 24, [orc.Camera(0.2, 0.65, 1.75), away, orc.Camera(0.5, 0.7, 0.9), orc.Camera(1.4, 0.75, 3.0)]),
    )
    for (W, H, band, cams) in cases:
        frames = [hip.make_frame(c.get_view(), c.get_pos(), W / H, 45.0, W, H) for c in cams]
        plan = tilesplit.make_plan(bounds, frames, world, band)
        # 1. every rank derived the same plan from its own copy of the inputs
        mine = np.frombuffer(tilesplit.plan_bytes(plan), np.uint8).copy()
        allp = [torch.zeros(len(mine), dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(allp, torch.from_numpy(mine))
        check(all(bytes(p.numpy()) == bytes(mine) for p in allp), f"{W}x{H}: plans differ between ranks")
        # 2. the plan is a partition: every row comes from exactly one (part, local row), parts are contiguous from 0
        src_part, src_local = tilesplit.row_sources(plan)
        check(plan.render_parts == (world - 1 if world >= 4 else world), "render_parts")
        check(tilesplit.part_of_rank(plan, 0) == (-1 if world >= 4 else 0), "rank 0 gathers only from 4 ranks on")
        seen = set()
        for p in range(plan.render_parts):
            rows = tilesplit.row_map(plan, p)
            check(len(rows) == tilesplit.rows_of_part(plan, p) <= plan.rows_part0, "rows of a part")
            check(np.array_equal(src_local[rows], np.arange(len(rows))), "local rows of a part are 0..n-1 in global order")
            seen.update(int(y) for y in rows)
        check(seen == set(range(H)), "the parts cover every row exactly once")
        check(plan.cropped == 1 and plan.pack_floats == sum(plan.rows_part0 * plan.win_w[i] for i in range(len(cams))), "pack_floats")
        for i in range(len(cams)):
            check(0 <= plan.win_x0[i] and plan.win_x0[i] + plan.win_w[i] <= W and plan.win_w[i] > 0, "window inside the frame")
        # 3. render my part, pack the plan's windows, exchange the plan's counts, assemble with the plan's offsets
        part = tilesplit.part_of_rank(plan, rank)
        packed = None
        if part >= 0:
            local = render_part(scene, frames, plan, part)
            packed = tilesplit.pack_columns(plan, local)
            check(packed.size == plan.pack_floats, "a rank ships exactly pack_floats floats")
        gathered = tilesplit.exchange(plan, rank, packed)
        if rank == 0:
            imgs = tilesplit.assemble(plan, gathered)
            for i, c in enumerate(cams):
                want, _ = orc.render(nodes, g.min, g.voxel_size, c.get_view(), c.get_pos(), W / H, 45.0, W, H)
                check(imgs[i].tobytes() == np.ascontiguousarray(want, F).tobytes(), f"{W}x{H} band {band} frame {i}: assembled frame differs from the oracle")
        else:
            check(gathered is None, "only rank 0 receives")
    # a batch too large for the window table ships whole rows: same counts on every rank
    W, H = 24, 16
    c0 = orc.Camera(0.5, 0.7, 1.8)
    many = [hip.make_frame(c0.get_view(), c0.get_pos(), W / H, 45.0, W, H)] * (hip.SPLIT_MAX_FRAMES + 1)
    plan = tilesplit.make_plan(bounds, many, world, 8)
    check(plan.cropped == 0 and plan.pack_floats == plan.full_floats == plan.rows_part0 * W * len(many), "uncropped batch")
    if why:
        print(f"rank {rank}: " + "; ".join(why[:5]), file=sys.stderr, flush=True)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
