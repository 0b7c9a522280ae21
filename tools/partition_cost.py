#!/usr/bin/env python3
"""Per-rank GPU work of the N-way screen split, measured on ONE GPU (the xGMI gather itself cannot be measured
here): traversal kernel of part p of N (4-byte shade payload), and rank 0's assemble kernel.  Config 2."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ray_tracing_octrees_amd as rto
from ray_tracing_octrees_amd import hip

W, H, band = 1920, 1080, 16
grid = rto.VoxelGrid.test_sphere(256)
root = rto.createOctreeFromVoxelGrid(grid)
nodes = root.flatten()
cam = rto.Camera(0.5, 0.7, 1.8)
frame = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
ctx = rto.Context(0)
ctx.upload_octree(nodes, grid.min, grid.voxelSize)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
full = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
for n in (1, 2, 4, 8):
    rows0 = ctx.partition_rows(frame, hip.Partition(n, 0, band)) if n > 1 else H
    gathered = torch.zeros((n, rows0, W), dtype=torch.float32, device="cuda")
    line = [f"N={n}: part buffer {rows0 * W * 4 / 1e6:.2f} MB;"]
    worst = 0.0
    for p in range(n):
        part = hip.Partition(n, p, band) if n > 1 else None
        for _ in range(12):
            ctx.render_shade_device(frame, gathered[p].data_ptr(), part, stream.cuda_stream)
        torch.cuda.synchronize()
        ctx.timing_begin(60)
        for _ in range(60):
            ctx.render_shade_device(frame, gathered[p].data_ptr(), part, stream.cuda_stream)
        torch.cuda.synchronize()
        k = np.asarray(ctx.timing_read())
        worst = max(worst, float(k.mean()))
        if p in (0, n - 1):
            line.append(f"part {p} kernel {k.mean() * 1e3:.1f} us")
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
    for a, b in ev:
        a.record(stream)
        ctx.assemble_shade_device(frame, hip.Partition(n, 0, band), gathered.data_ptr(), full.data_ptr(), stream.cuda_stream)
        b.record(stream)
    torch.cuda.synchronize()
    asm = sorted(a.elapsed_time(b) for a, b in ev)[25]
    line.append(f"slowest part {worst * 1e3:.1f} us; assemble_shade {asm * 1e3:.1f} us")
    print(" ".join(line), flush=True)
