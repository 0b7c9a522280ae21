"""Kernel time of config 2's scene against the camera's distance (the sphere from filling the frame to a few tiles), and of an empty frame:
where a frame's floor lies when the geometry's rectangle is small -- the fill duty (33 MB of black) and the deepest rays.  GPU box.
    python tools/radius_sweep.py [minfill=N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ray_tracing_octrees_amd as rto

W, H = 1920, 1080
g = rto.VoxelGrid.test_sphere(256)
ctx = rto.Context(0)
ctx.build_octree(g.data, g.min, g.voxelSize)
fb = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
def frames_of(cam, fov=45.0): return rto.make_frame(cam.getView(), cam.getPos(), W / H, fov, W, H)
warm = frames_of(rto.Camera(0.5, 0.7, 1.8))
for _ in range(3000):
    ctx.render_device(warm, fb.data_ptr(), None, s.cuda_stream)
torch.cuda.synchronize()
cases = [("r=1.0", rto.Camera(0.5, 0.7, 1.0)), ("r=1.8 (headline)", rto.Camera(0.5, 0.7, 1.8)), ("r=2.5", rto.Camera(0.5, 0.7, 2.5)), ("r=4", rto.Camera(0.5, 0.7, 4.0)),
         ("r=6", rto.Camera(0.5, 0.7, 6.0)), ("r=10", rto.Camera(0.5, 0.7, 10.0)), ("r=30", rto.Camera(0.5, 0.7, 30.0)), ("r=500 (app default)", rto.Camera(0.5, 0.7, 500.0))]
away = rto.Camera(0.5, 0.7, 1.8); away.setTarget(np.array([50.0, 0.0, 0.0], np.float32))
cases.append(("looking past the volume", away))
for name, cam in cases:
    f = frames_of(cam)
    for _ in range(30):
        ctx.render_device(f, fb.data_ptr(), None, s.cuda_stream)
    torch.cuda.synchronize()
    ctx.timing_begin(40)
    for _ in range(40):
        ctx.render_device(f, fb.data_ptr(), None, s.cuda_stream)
    torch.cuda.synchronize()
    k = np.sort(ctx.timing_read())
    ctx.timing_begin(-1)
    st = ctx.frame_stats(f)
    hit = int((fb[..., 0] > 0).sum().item())
    print(f"{name:28s} kernel median {k[len(k) // 2] * 1e3:7.1f} us  min {k[0] * 1e3:7.1f}   hit pixels {hit:8d}  capped rays {st['capped']}")
