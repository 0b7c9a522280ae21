#!/usr/bin/env python3
"""Does replaying the frame sequence from a captured HIP graph shrink the gap between dependent launches?
Config 2, static camera: 16 frames (two launch-order periods) captured once, replayed; compared with plain launches."""
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
s = torch.cuda.Stream()
buf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
for _ in range(64):
    ctx.render_device(frame, buf.data_ptr(), None, s.cuda_stream)
torch.cuda.synchronize()

n = 1600
t = time.perf_counter()
for _ in range(n):
    ctx.render_device(frame, buf.data_ptr(), None, s.cuda_stream)
torch.cuda.synchronize()
print(f"plain launches : {(time.perf_counter() - t) / n * 1e6:.2f} us/frame")

g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    for _ in range(16):
        ctx.render_device(frame, buf.data_ptr(), None, s.cuda_stream)
torch.cuda.synchronize()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(n // 16):
    g.replay()
torch.cuda.synchronize()
print(f"graph replay   : {(time.perf_counter() - t) / n * 1e6:.2f} us/frame (16 frames per graph)")

# the whole timed region as ONE graph, with the per-launch event ring inside it
K = 2000
ctx.timing_begin(K)
torch.cuda.synchronize()
t = time.perf_counter()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=s):
    for _ in range(K):
        ctx.render_device(frame, buf.data_ptr(), None, s.cuda_stream)
torch.cuda.synchronize()
t_cap = time.perf_counter() - t
t = time.perf_counter()
g2.replay()
torch.cuda.synchronize()
dt = time.perf_counter() - t
k = ctx.timing_read()
print(f"one graph of {K} frames: capture+instantiate {t_cap * 1e3:.1f} ms, replay {dt / K * 1e6:.2f} us/frame, "
      f"ring: {len(k)} kernels, avg {sum(k) / len(k) * 1e3:.2f} us, min {min(k) * 1e3:.2f}")
import numpy as np
from oracle import orc
og = orc.test_sphere_grid(256); on = orc.build_flat_octree(og); oc = orc.Camera(0.5, 0.7, 1.8)
want, _ = orc.render(on, og.min, og.voxel_size, oc.get_view(), oc.get_pos(), W / H, 45.0, W, H, nthreads=16)
print("frame equals oracle:", buf.cpu().numpy().tobytes() == want.tobytes())
