#!/usr/bin/env python3
"""Where the ~6.6 us between bench.py's ms_per_step and the kernel's own duration go (config 2, one GPU):
frames back to back with / without the per-launch event ring, with / without the launch-order rebuild."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ray_tracing_octrees_amd as rto

W, H = 1920, 1080
grid = rto.VoxelGrid.test_sphere(256)
root = rto.createOctreeFromVoxelGrid(grid)
nodes = root.flatten()
cam = rto.Camera(0.5, 0.7, 1.8)
frame = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
ctx = rto.Context(0)
ctx.upload_octree(nodes, grid.min, grid.voxelSize)
stream = torch.cuda.Stream()
buf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")


def run(n, ring, period):
    ctx.set_launch_order(1, period)
    for _ in range(40):
        ctx.render_device(frame, buf.data_ptr(), None, stream.cuda_stream)
    torch.cuda.synchronize()
    ctx.timing_begin(n if ring else 0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        ctx.render_device(frame, buf.data_ptr(), None, stream.cuda_stream)
    t_issue = time.perf_counter() - t
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    k = ctx.timing_read() if ring else []
    return dt / n * 1e6, t_issue / n * 1e6, (sum(k) / len(k) * 1e3 if len(k) else float("nan"))


for ring in (True, False):
    for period in (4, 1000000):
        us, issue, k = run(400, ring, period)
        print(f"event ring {'on ' if ring else 'off'}, order rebuilt every {period if period < 1000 else 'never (after the first)'}: "
              f"{us:.2f} us/frame, host issue {issue:.2f} us/frame, kernel {k:.2f} us", flush=True)
