"""tools/lists_debug.py -- the launch lists of a few config-2 frames: counts per bucket, listed slots, time per frame."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import ray_tracing_octrees_amd as rto
from oracle import orc

og = orc.test_sphere_grid(256)
ctx = rto.Context(0)
ctx.build_octree(og.data, og.min, og.voxel_size)
cam = orc.Camera(0.5, 0.7, 1.8)
W, H = 1920, 1080
f = rto.make_frame(cam.get_view(), cam.get_pos(), W / H, 45.0, W, H)
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
for lists in (True, False, True):
    ctx.debug_set_launch_lists(lists)
    for k in range(6):
        ctx.render_device(f, out.data_ptr())
        ctx.synchronize()
        if lists:
            i = ctx.debug_launch_lists_info()
            print(k, "frames", i["frames"], "listed_slots", i["listed_slots"], "host_count", i["host_count"], "totals", i["counts"].sum(axis=1), flush=True)
    ctx.timing_begin(-1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(200):
        ctx.render_device(f, out.data_ptr())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("lists", lists, "us/frame", (t1 - t0) / 200 * 1e6, flush=True)
    if lists:
        i = ctx.debug_launch_lists_info()
        print(i["counts"])
for lists in (False, True):
    ctx.debug_set_launch_lists(lists)
    for k in range(4):
        ctx.render_device(f, out.data_ptr())
    ctx.synchronize()
    cost = ctx.debug_tile_cost()
    vals, cnts = np.unique(np.clip(cost, -2, 80), return_counts=True)
    print("lists", lists, "cost histogram:", dict(zip(vals.tolist(), cnts.tolist())), flush=True)
import ctypes as C
L = ctx._L
L.rto_debug_launch_list_tiles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]
i = ctx.debug_launch_lists_info()
cur = (i["frames"] - 1) % 3
print("cur buffer", cur, "bucket31 count", i["counts"][cur][31], "bucket 8", i["counts"][cur][8])
cost2 = cost.reshape((H + 7) // 8, (W + 7) // 8)
for b in (31, 8):
    t = np.zeros(20000, np.int32)
    assert L.rto_debug_launch_list_tiles(ctx._h, cur, b, t.ctypes.data, t.size) == 0
    n = int(i["counts"][cur][b])
    t = t[:n]
    tx, ty = t & 0xffff, (t.astype(np.uint32) >> 16).astype(np.int64)
    cc = cost2[np.clip(ty, 0, cost2.shape[0] - 1), np.clip(tx, 0, cost2.shape[1] - 1)]
    vals, cnts = np.unique(cc, return_counts=True)
    print("bucket", b, "entries", n, "unique tiles", len(np.unique(t)), "tx range", tx.min(), tx.max(), "ty range", ty.min(), ty.max(), "costs of listed tiles:", dict(zip(vals.tolist()[:12], cnts.tolist()[:12])))
    print("  first entries", [(int(a), int(b_)) for a, b_ in zip(tx[:12], ty[:12])])
