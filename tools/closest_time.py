"""tools/closest_time.py -- the closest-hit mode (rto_render_closest_*) at config 2's size: time per frame of the near-first
descriptor-tree form (k_closest_near_first, default), of the pop-order form (k_closest_lean, RTO_KERNEL_PACKED_V1) and of the
node-by-node form (k_trace_closest, RTO_KERNEL_GENERIC), as wall time over 200 back-to-back launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ray_tracing_octrees_amd as rto
from ray_tracing_octrees_amd import hip

ctx = rto.Context(0)
g = rto.VoxelGrid.test_sphere(256)
ctx.build_octree(g.data, g.min, g.voxelSize)
cam = rto.Camera(0.5, 0.7, 1.8)
W, H = 1920, 1080
fr = hip.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
ref = None
for name, mode in (("descriptor tree, near children first", hip.KERNEL_AUTO), ("descriptor tree, the reference's pop order", hip.KERNEL_PACKED_V1),
                   ("node by node", hip.KERNEL_GENERIC)):
    ctx.set_kernel(mode)
    ts = []
    for i in range(8):
        ctx.synchronize()
        t = time.perf_counter()
        for _ in range(200):
            ctx.render_closest_device(fr, out.data_ptr())
        ctx.synchronize()
        ts.append((time.perf_counter() - t) / 200 * 1e3)
    ts.sort()
    img = out.cpu().numpy()
    if ref is None:
        ref = img
    same = np.array_equal(ref.view(np.uint32), img.view(np.uint32))
    print(f"closest-hit, {name}: per frame median {ts[len(ts) // 2] * 1e3:.1f} us, min {ts[0] * 1e3:.1f} us; frames identical: {same}", flush=True)
