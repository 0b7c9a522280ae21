"""tools/frames_per_launch.py [config-2 dim] -- us per frame of rto_render_batch_device with 1, 2, 3, 4, 6, 8 frames per kernel launch
(graph of 48 frames, replayed), and of the same with a camera that moves between the frames of a batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ray_tracing_octrees_amd as rto
from ray_tracing_octrees_amd import hip

W, H = 1920, 1080
g = rto.VoxelGrid.test_sphere(256)
ctx = rto.Context(0)
ctx.build_octree(g.data, g.min, g.voxelSize)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
bufs = torch.empty((8, H, W, 4), dtype=torch.float32, device="cuda")
def frames_for(n, moving):
    out = []
    for i in range(n):
        cam = rto.Camera(0.5 + (0.01 * i if moving else 0.0), 0.7, 1.8)
        out.append(rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H))
    return hip.Context.frame_array(out)
t_end = time.perf_counter() + 0.3
one = frames_for(1, False)
while time.perf_counter() < t_end:
    for _ in range(20):
        ctx.render_batch_device(one, bufs.data_ptr(), bufs.stride(0) * 4, None, False, stream.cuda_stream)
    torch.cuda.synchronize()
ctx.timing_begin(0)
for moving in (False, True):
    for F in (1, 2, 3, 4, 6, 8):
        arr = frames_for(F, moving)
        for _ in range(10):
            ctx.render_batch_device(arr, bufs.data_ptr(), bufs.stride(0) * 4, None, False, stream.cuda_stream)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        launches = 48 // F
        with torch.cuda.graph(graph, stream=stream):
            for _ in range(launches):
                ctx.render_batch_device(arr, bufs.data_ptr(), bufs.stride(0) * 4, None, False, stream.cuda_stream)
        graph.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(40):
            graph.replay()
        b.record(stream)
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / (40 * launches * F)
        print(f"{'moving' if moving else 'static'} camera, {F} frames per launch: {us:6.2f} us per frame, {W * H / us / 1e3:6.2f} Grays/s", flush=True)
