"""Does a frame's cost depend on where the orbiting camera stands?  Static frames of config 2 at several theta: us per frame
(plain launches, settled launch order) and pops per ray."""
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
for th in (0.5, 0.8, 1.1, 1.4, 1.7, 2.0, 2.3, 2.6, 2.9):
    cam = rto.Camera(th, 0.7, 1.8)
    f = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
    for _ in range(80): ctx.render_device(f, buf.data_ptr())
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(200): ctx.render_device(f, buf.data_ptr())
    ctx.synchronize()
    us = (time.perf_counter() - t) / 200 * 1e6
    st = ctx.frame_stats(f)
    ctx.render_device(f, buf.data_ptr()); ctx.synchronize()
    cost = ctx.debug_tile_cost()
    live = cost[cost > 0]
    ctx.debug_set_tile_mask(0)
    for _ in range(40): ctx.render_device(f, buf.data_ptr())
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(200): ctx.render_device(f, buf.data_ptr())
    ctx.synchronize()
    us0 = (time.perf_counter() - t) / 200 * 1e6
    ctx.render_device(f, buf.data_ptr()); ctx.synchronize()
    cost0 = ctx.debug_tile_cost()
    ctx.debug_set_tile_mask(1)
    print(f"theta {th:.1f}: {us:6.1f} us per frame ({us0:5.1f} without the mask), {st['pops'] / (W * H):6.2f} pops per ray, {st['capped']} capped; "
          f"tiles with trips {len(live)} (no mask: {(cost0 > 0).sum()}), trips per live tile mean {live.mean():.1f} max {live.max()}, sum {int(live.sum())}")
