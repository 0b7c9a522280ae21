"""Cycles per loop trip per SIMD of the traversal kernel under a perfectly uniform load: a frame with a tiny field of view
(every ray of the image is almost the same ray), so every tile does the same work and the machine is full until the end.
frame time x 1024 SIMDs x 2.4 GHz / (tiles x trips) = SIMD-cycles per trip at saturation.  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ray_tracing_octrees_amd as rto

dim, W, H = 256, 1920, 1080
g = rto.VoxelGrid.test_sphere(dim)
ctx = rto.Context(0)
ctx.build_octree(g.data, g.min, g.voxelSize)
for name, kern in (("lean", rto.KERNEL_PACKED), ("packed_v3", rto.KERNEL_PACKED_V3)):
    ctx.set_kernel(kern)
    for theta, phi, label in ((0.5, 0.7, "through the shell"), (0.02, 0.3, "grazing")):
        for tgt in ((0.0, 0.0, 0.0), (0.31, 0.0, 0.0), (0.0, 0.395, 0.0)):
            cam = rto.Camera(theta, phi, 1.8)
            cam.setTarget(np.array(tgt, np.float32))
            f = rto.make_frame(cam.getView(), cam.getPos(), W / H, 0.002, W, H)
            for _ in range(4):
                ctx.render_host(f)
            ts = []
            for _ in range(10):
                ctx.render_host(f)
                ts.append(ctx.last_kernel_ms())
            ms = float(np.median(ts))
            cost = ctx.debug_tile_cost()
            st = ctx.frame_stats(f)
            tiles = (cost > 0).sum()
            trips = cost[cost > 0].mean() if tiles else 0
            if tiles:
                cyc = ms * 1e-3 * 2.4e9 * 1024 / (tiles * trips)
                print(f"{name:10s} target {tgt} {label:18s}: {ms*1e3:7.1f} us, tiles {tiles}, trips/tile {trips:.1f} (min {cost[cost>0].min()} max {cost.max()}), "
                      f"pops/ray {st['pops']/st['rays']:.1f} -> {cyc:6.0f} SIMD-cycles per trip")
