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
    print(f"{label:56s} median {np.median(ms[2:])*1e3:7.1f} us  min {min(ms)*1e3:7.1f} us")
ctx.set_launch_order(1, 1)
ctx.render_host(f)
cost = ctx.debug_tile_cost()
idx = np.arange(len(cost))
desc = np.argsort(-cost, kind="stable")
def variant(thr):
    hi = desc[cost[desc] >= thr]
    zero = idx[cost == 0]
    mid = desc[(cost[desc] < thr) & (cost[desc] > 0)]
    return np.concatenate([hi, zero, mid]).astype(np.int32)
ctx.debug_set_tile_order(desc.astype(np.int32)); run("full sort desc (zero-cost tiles last)")
for thr in (32, 24, 16, 8, 4):
    ctx.debug_set_tile_order(variant(thr)); run(f"cost>={thr} desc, then zero-cost, then the rest desc")
# zero-cost tiles interleaved: one zero tile after every k-th costly tile
nz = desc[cost[desc] > 0]; z = idx[cost == 0]
out = []
zi = 0
ratio = len(z) / max(1, len(nz))
acc = 0.0
for t in nz:
    out.append(t); acc += ratio
    while acc >= 1 and zi < len(z):
        out.append(z[zi]); zi += 1; acc -= 1
out.extend(z[zi:])
ctx.debug_set_tile_order(np.array(out, np.int32)); run("zero-cost tiles interleaved evenly with the sorted rest")
