import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ray_tracing_octrees_amd as rto
g = rto.VoxelGrid.test_sphere(256)
root = rto.createOctreeFromVoxelGrid(g)
ctx = rto.Context(0)
ctx.upload_octree(root.flatten(), g.min, g.voxelSize)
W, H = 1920, 1080
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
st = torch.cuda.Stream()
def run(frames, label):
    for f in frames[:8]:
        ctx.render_device(f, out.data_ptr(), None, st.cuda_stream)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for f in frames:
        ctx.render_device(f, out.data_ptr(), None, st.cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / len(frames)
    print(f"{label:44s} {dt*1e6:7.1f} us/frame  {W*H/dt/1e9:6.2f} Grays/s")
cam = rto.Camera(0.5, 0.7, 1.8)
static = [rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)] * 200
orbit = []
for i in range(200):
    c2 = rto.Camera(0.5, 0.7 + 0.01 * i, 1.8)
    orbit.append(rto.make_frame(c2.getView(), c2.getPos(), W / H, 45.0, W, H))
fast = []
for i in range(200):
    c2 = rto.Camera(0.5 + 0.3 * np.sin(i * 0.05), 0.7 + 0.05 * i, 1.8 + 0.4 * np.sin(i * 0.03))
    fast.append(rto.make_frame(c2.getView(), c2.getPos(), W / H, 45.0, W, H))
ctx.set_launch_order(0); run(static, "centre-out, static"); run(orbit, "centre-out, orbit 0.01 rad/frame"); run(fast, "centre-out, fast motion 0.05 rad/frame")
for R in (1, 2, 4, 8, 16):
    ctx.set_launch_order(1, R)
    run(static, f"temporal R={R}, static"); run(orbit, f"temporal R={R}, orbit 0.01 rad/frame"); run(fast, f"temporal R={R}, fast motion 0.05 rad/frame")
