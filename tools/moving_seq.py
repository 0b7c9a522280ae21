"""tools/moving_seq.py -- kernel time of every launch (the timing ring's event pairs) while the camera changes: A x 12, B x 12 (0.01 rad
away), A x 12, then a 24-frame orbit, config 2.  Dev builds take RTO_ORDER_MOVED_PERIOD."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ray_tracing_octrees_amd as rto

g = rto.VoxelGrid.test_sphere(256)
ctx = rto.Context(0)
ctx.build_octree(g.data, g.min, g.voxelSize)
W, H = 1920, 1080
buf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
def frame(th):
    cam = rto.Camera(th, 0.7, 1.8)
    return rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
A, B = frame(0.5), frame(0.51)
ctx.timing_begin(-1)
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    for _ in range(20): ctx.render_device(A, buf.data_ptr())
    ctx.synchronize()
seq = [A] * 12 + [B] * 12 + [A] * 12 + [frame(0.5 + 0.01 * i) for i in range(1, 25)] + [frame(0.74)] * 12
ctx.timing_begin(len(seq))
for f in seq: ctx.render_device(f, buf.data_ptr())
ctx.synchronize()
ms = ctx.timing_read()
for name, lo, hi in (("A", 0, 12), ("B", 12, 24), ("A", 24, 36), ("orbit", 36, 60), ("last orbit camera standing", 60, 72)):
    print(f"{name:28s}", " ".join(f"{x * 1e3:5.1f}" for x in ms[lo:hi]), flush=True)
print("violations", ctx.debug_sort_violations())
