"""Fixed per-frame costs: all rays miss the root (camera far away / looking away) vs the real frame."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ray_tracing_octrees_amd as rto

g = rto.VoxelGrid.test_sphere(256)
root = rto.createOctreeFromVoxelGrid(g)
nodes = root.flatten()
ctx = rto.Context(0)
ctx.upload_octree(nodes, g.min, g.voxelSize)
W, H = 1920, 1080
def timeit(f, reps=30):
    ms = []
    for _ in range(reps):
        ctx.render_host(f); ms.append(ctx.last_kernel_ms())
    return np.median(ms), min(ms)
for name, k in (("packed", rto.KERNEL_PACKED), ("generic", rto.KERNEL_GENERIC)):
    ctx.set_kernel(k)
    for label, (t, p, r, tgt) in (("normal", (0.5, 0.7, 1.8, (0, 0, 0))), ("all-miss (look away)", (0.5, 0.7, 1.8, (5.0, 5.0, 9.0))),
                                  ("tiny sphere r=40", (0.5, 0.7, 40.0, (0, 0, 0))), ("inside r=0.3 (all hit fast)", (0.5, 0.7, 0.3, (0, 0, 0)))):
        cam = rto.Camera(t, p, r); cam.setTarget(np.array(tgt, np.float32))
        f = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
        st = ctx.frame_stats(f)
        med, mn = timeit(f)
        print(f"{name:8s} {label:28s} hits {st['hits']:8d} pops/ray {st['pops']/st['rays']:6.2f}  kernel med {med*1e3:7.1f} us  min {mn*1e3:7.1f} us")
