"""tools/moving_cost.py -- why a moving camera's frame costs more than a standing one's (config 2, single-frame plain launches):
us per frame for the same camera repeated, two cameras 0.01 rad apart alternating, and an orbit of 0.01 rad per frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ray_tracing_octrees_amd as rto

g = rto.VoxelGrid.test_sphere(256)
ctx = rto.Context(0)
ctx.build_octree(g.data, g.min, g.voxelSize)
ctx.timing_begin(-1)
W, H = 1920, 1080
buf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
def frame(th):
    cam = rto.Camera(th, 0.7, 1.8)
    return rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
def run(frames, reps=3):
    best = 1e9
    for _ in range(reps):
        for f in frames[:16]: ctx.render_device(f, buf.data_ptr())
        ctx.synchronize()
        t = time.perf_counter()
        for f in frames: ctx.render_device(f, buf.data_ptr())
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t) / len(frames) * 1e6)
    return best
t_end = time.perf_counter() + 0.3
f0 = frame(0.5)
while time.perf_counter() < t_end:
    run([f0] * 20, 1)
for th in (0.5, 1.7):
    a, b = frame(th), frame(th + 0.01)
    print(f"theta {th}: standing {run([a] * 240):.2f} us; alternating with theta + 0.01: {run([a, b] * 120):.2f} us; "
          f"orbit of 240 frames from here: {run([frame(th + 0.01 * i) for i in range(240)]):.2f} us; "
          f"orbit of 0.001 rad per frame: {run([frame(th + 0.001 * i) for i in range(240)]):.2f} us; "
          f"the 240 orbit cameras, each standing for 8 frames: {run([frame(th + 0.01 * (i // 8)) for i in range(240 * 8)], 1):.2f} us", flush=True)
