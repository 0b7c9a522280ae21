"""tools/lists_debug2.py -- multiplicity of the tiles listed by the FIRST frame of a fresh context."""
import sys, ctypes as C
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
L = ctx._L
L.rto_debug_launch_list_tiles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]
nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for k in range(nframes):
    ctx.render_device(f, out.data_ptr())
    ctx.synchronize()
i = ctx.debug_launch_lists_info()
cur = (i["frames"] - 1) % 3
cost = ctx.debug_tile_cost().reshape((H + 7) // 8, (W + 7) // 8)
steps = np.abs(ctx.render_steps(f))
capped = (steps >= 512)[: H // 8 * 8].reshape(H // 8, 8, W // 8, 8).sum(axis=(1, 3))
print("frames", i["frames"], "listed_slots", i["listed_slots"], "totals", i["counts"].sum(axis=1))
allt = []
for b in range(32):
    n = int(i["counts"][cur][b])
    if n == 0:
        continue
    t = np.zeros(max(n, 1), np.int32)
    assert L.rto_debug_launch_list_tiles(ctx._h, cur, b, t.ctypes.data, t.size) == 0
    u, m = np.unique(t[:n], return_counts=True)
    print("bucket", b, "entries", n, "unique", len(u), "max multiplicity", m.max())
    if m.max() > 1:
        order = np.argsort(-m)[:10]
        for o in order:
            tx, ty = int(u[o]) & 0xffff, int(np.uint32(u[o]) >> 16)
            print("    tile", (tx, ty), "x", int(m[o]), "cost", int(cost[ty, tx]), "capped rays", int(capped[ty, tx]), "raster index in row", (tx - 0) % 8)
