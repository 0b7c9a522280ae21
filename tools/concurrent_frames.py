"""tools/concurrent_frames.py [branches] -- config 2 with `branches` independent chains of frames inside ONE HIP graph (each chain
on its own stream and frame buffer): what the GPU delivers when frames of a sequence overlap instead of following one another.
One frame's kernel is bounded by its deepest tile's serial latency; overlapping frames fill the wave slots that leaves idle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ray_tracing_octrees_amd as rto

branches = int(sys.argv[1]) if len(sys.argv) > 1 else 2
per_branch = 25
W, H = 1920, 1080
g = rto.VoxelGrid.test_sphere(256)
cam = rto.Camera(0.5, 0.7, 1.8)
frame = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
ctx = rto.Context(0)
ctx.build_octree(g.data, g.min, g.voxelSize)
main = torch.cuda.Stream()
torch.cuda.set_stream(main)
side = [torch.cuda.Stream() for _ in range(branches - 1)]
streams = [main] + side
bufs = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(branches)]
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:                       # clock ramp + launch-order tables of every stream
    for s, b in zip(streams, bufs):
        for _ in range(10):
            ctx.render_device(frame, b.data_ptr(), None, s.cuda_stream)
    torch.cuda.synchronize()
ctx.timing_begin(0)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=main):
    for s in side:
        s.wait_stream(main)
    for _ in range(per_branch):
        for s, b in zip(streams, bufs):
            ctx.render_device(frame, b.data_ptr(), None, s.cuda_stream)
    for s in side:
        main.wait_stream(s)
graph.replay(); torch.cuda.synchronize()
reps = 40
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(main)
for _ in range(reps):
    graph.replay()
b.record(main)
torch.cuda.synchronize()
frames = reps * per_branch * branches
us = a.elapsed_time(b) * 1e3 / frames
print(f"{branches} chains: {us:.2f} us per frame, {W * H / us / 1e3:.2f} Grays/s")
