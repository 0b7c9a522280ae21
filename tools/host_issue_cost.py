#!/usr/bin/env python3
"""Host time per frame of the N>1 path (kernel launch + torch.distributed.gather + assemble launch, three pipelines),
measured with a ONE-rank RCCL group on one GPU: what the Python side costs before any traffic between GPUs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29621")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import ray_tracing_octrees_amd as rto
from ray_tracing_octrees_amd import tilesplit

W, H = 1920, 1080
grid = rto.VoxelGrid.test_sphere(256)
root = rto.createOctreeFromVoxelGrid(grid)
ctx = rto.Context(0)
ctx.upload_octree(root.flatten(), grid.min, grid.voxelSize)
cam = rto.Camera(0.5, 0.7, 1.8)
frame = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
for npipe, fpg in ((3, 1), (3, 2), (3, 4), (3, 8), (1, 4)):
    R = [tilesplit.TileSplitRenderer(tilesplit.HipBackend(ctx), 0, 1, payload="shade", force_collective=True) for _ in range(npipe)]
    S = [torch.cuda.Stream() for _ in range(npipe)]
    torch.cuda.set_stream(S[0])

    def run(n):
        for j in range(n // fpg):
            with torch.cuda.stream(S[j % npipe]):
                R[j % npipe].submit_batch([frame] * fpg)
        for i in range(npipe):
            with torch.cuda.stream(S[i]):
                R[i].flush_batch()

    run(120)
    torch.cuda.synchronize()
    n = 1920
    t = time.perf_counter()
    run(n)
    t_issue = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    print(f"{npipe} pipeline(s), {fpg} frame(s) per gather: host issue {t_issue / n * 1e6:6.1f} us/frame, complete {t_all / n * 1e6:6.1f} us/frame", flush=True)
dist.destroy_process_group()
