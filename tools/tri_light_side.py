"""tools/tri_light_side.py -- config 5's frame (512^3, 4K, triangles + shadow rays) from the lit side (the bench's camera) and from the far side of the
light: us per frame (the shadow ray of a hit that faces away from the light is skipped in colour frames)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ray_tracing_octrees_amd as rto
g = rto.VoxelGrid.test_sphere(512)
ctx = rto.Context(0)
ctx.build_octree(g.data, g.min, g.voxelSize)
ctx.build_leaf_triangles(None)
ctx.timing_begin(-1)
W, H = 3840, 2160
buf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
for name, th, ph in (("lit side (bench camera)", 0.5, 0.7), ("side on", 2.1, 0.0), ("far side of the light", 0.5 + 3.14159, -0.7)):
    cam = rto.Camera(th, ph, 1.8)
    f = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        for _ in range(10): ctx.render_triangles_device(f, buf.data_ptr(), True)
        ctx.synchronize()
    t = time.perf_counter()
    for _ in range(100): ctx.render_triangles_device(f, buf.data_ptr(), True)
    ctx.synchronize()
    print(f"{name:26s}: {(time.perf_counter() - t) / 100 * 1e6:7.1f} us per frame", flush=True)
