import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ray_tracing_octrees_amd as rto
g = rto.VoxelGrid.test_sphere(256)
root = rto.createOctreeFromVoxelGrid(g)
ctx = rto.Context(0)
ctx.upload_octree(root.flatten(), g.min, g.voxelSize)
W, H = 1920, 1080
cam = rto.Camera(0.5, 0.7, 1.8)
f = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
def run(label, n=40):
    ms = []
    for i in range(n):
        ctx.render_host(f); ms.append(ctx.last_kernel_ms())
    print(f"{label:40s} median {np.median(ms[2:])*1e3:7.1f} us  min {min(ms)*1e3:7.1f} us")
ctx.set_launch_order(0); run("centre-out (no cost recording)")
ctx.set_launch_order(1); run("temporal (device sort each frame)")
ms = []
for i in range(40):
    cam2 = rto.Camera(0.5, 0.7 + 0.01 * i, 1.8)
    f2 = rto.make_frame(cam2.getView(), cam2.getPos(), W / H, 45.0, W, H)
    ctx.render_host(f2); ms.append(ctx.last_kernel_ms())
print(f"{'temporal, camera orbiting 0.01 rad/frame':40s} median {np.median(ms[2:])*1e3:7.1f} us")
ctx.set_launch_order(0)
ms = []
for i in range(40):
    cam2 = rto.Camera(0.5, 0.7 + 0.01 * i, 1.8)
    f2 = rto.make_frame(cam2.getView(), cam2.getPos(), W / H, 45.0, W, H)
    ctx.render_host(f2); ms.append(ctx.last_kernel_ms())
print(f"{'centre-out, same orbit':40s} median {np.median(ms[2:])*1e3:7.1f} us")
ctx.set_launch_order(1)
ctx.render_host(f)
cost = ctx.debug_tile_cost()
print("tiles", len(cost), "cost>0", int((cost > 0).sum()), "cost>=16", int((cost >= 16).sum()), "max", cost.max())
ident = np.arange(len(cost), dtype=np.int32)
ctx.debug_set_tile_order(ident); run("fixed: identity (row-major)")
ctx.debug_set_tile_order(np.argsort(-cost, kind="stable").astype(np.int32)); run("fixed: full sort by cost desc")
deep = np.nonzero(cost >= 16)[0]; rest = np.nonzero(cost < 16)[0]
ctx.debug_set_tile_order(np.concatenate([deep, rest]).astype(np.int32)); run("fixed: cost>=16 first, rest row-major")
deep = np.nonzero(cost >= 32)[0]; rest = np.nonzero(cost < 32)[0]
ctx.debug_set_tile_order(np.concatenate([deep, rest]).astype(np.int32)); run("fixed: cost>=32 first, rest row-major")
rng = np.random.default_rng(0)
ctx.debug_set_tile_order(rng.permutation(len(cost)).astype(np.int32)); run("fixed: random permutation")
# interleave: deep tiles spread evenly among the light ones at the front half
order = np.argsort(-cost, kind="stable").astype(np.int32)
ctx.debug_set_tile_order(order[::-1].copy()); run("fixed: cost ascending (worst case)")
