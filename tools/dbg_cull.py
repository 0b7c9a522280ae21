import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ray_tracing_octrees_amd as rto
from oracle import orc
z = np.load(os.path.join(ROOT, "tests/golden/ref_scene_cache.npz"))
dims = tuple(int(x) for x in z["dims"]); data = np.unpackbits(z["packed"])[:dims[0]*dims[1]*dims[2]].reshape(dims[2],dims[1],dims[0])
g = orc.Grid(dims, z["min"].astype(np.float32), np.float32(z["voxel"]), data)
nodes = orc.build_flat_octree(g)
cam = orc.Camera(float(np.float32(np.pi/2)), 0.0, 500.0); cam.pan(0.0, 100.0)
view, pos = cam.get_view(), cam.get_pos()
W, H = 480, 270
aspect = W / H
want_nodes, vis = orc.cull_compact(nodes, g.min, g.voxel_size, view, 45.0, aspect)
want, st = orc.render(want_nodes, g.min, g.voxel_size, view, pos, aspect, 45.0, W, H)
wsteps = orc.render_steps(want_nodes, g.min, g.voxel_size, view, pos, aspect, 45.0, W, H)
ctx = rto.Context(0)
ctx.upload_octree(nodes, g.min, g.voxel_size)
ctx.update_frustum(view, 45.0, aspect, True)
info = ctx.info()
print("info", info.visible_nodes, info.culling_active, "oracle stats", st)
f = rto.make_frame(view, pos, aspect, 45.0, W, H)
for name, k in (("generic", rto.KERNEL_GENERIC), ("packed", rto.KERNEL_PACKED)):
    ctx.set_kernel(k)
    got = ctx.render_host(f)
    gsteps = ctx.render_steps(f)
    bad = (got.view(np.uint32) != want.view(np.uint32)).any(axis=-1)
    print(name, "bad pixels", int(bad.sum()), "stats", ctx.frame_stats(f))
    ys, xs = np.nonzero(bad)
    for y, x in list(zip(ys, xs))[:6]:
        print("   ", (x, y), got[y, x], want[y, x], "steps gpu", gsteps[y, x], "oracle", wsteps[y, x])
    print("   steps equal:", int((gsteps == wsteps).sum()), "of", W * H, "; oracle steps hist", np.unique(wsteps, return_counts=True)[0][:10])
