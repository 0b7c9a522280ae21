"""Looking for frames that cost more than their work: us per frame against the frame's pops (instrumented frame) over scenes,
cameras and frame sizes.  A frame whose time per pop stands out is worth a timeline (tools/timeline_simd.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ray_tracing_octrees_amd as rto

def calgary():
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_scene_cache.npz"))
    dims = tuple(int(x) for x in z["dims"])
    data = np.unpackbits(z["packed"])[: dims[0] * dims[1] * dims[2]].reshape(dims[2], dims[1], dims[0])
    return rto.VoxelGrid.from_array(data, z["min"].astype(np.float32), np.float32(z["voxel"]))

ctx = rto.Context(0)
ctx.timing_begin(-1)
cases = [("sphere256", [(0.5, 0.7, 1.8), (1.2, 2.0, 1.2), (0.3, 4.0, 0.9), (0.5, 0.7, 4.0)], [(1920, 1080), (3840, 2160), (640, 360)]),
         ("sphere512", [(0.5, 0.7, 1.8), (2.4, 1.0, 1.3)], [(1920, 1080), (3840, 2160)]),
         ("calgary", [(0.6, 0.5, 3500.0), (1.2, 0.1, 900.0), (0.2, 2.5, 2500.0), (2.5, 1.0, 3000.0)], [(1920, 1080), (3840, 2160)])]
for name, cams, sizes in cases:
    g = calgary() if name == "calgary" else rto.VoxelGrid.test_sphere(int(name[6:]))
    ctx.build_octree(g.data, g.min, g.voxelSize)
    for (W, H) in sizes:
        buf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
        for (t, p, r) in cams:
            cam = rto.Camera(t, p, r)
            f = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
            for _ in range(60): ctx.render_device(f, buf.data_ptr())
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(100): ctx.render_device(f, buf.data_ptr())
            ctx.synchronize()
            us = (time.perf_counter() - t0) / 100 * 1e6
            st = ctx.frame_stats(f)
            print(f"{name:9s} {W}x{H} cam {(t, p, r)}: {us:7.1f} us, {st['pops'] / (W * H):6.2f} pops per ray, {st['hits']:8d} hits, {st['capped']:6d} capped, "
                  f"{us * 1e3 / max(1, st['pops']) * 1e3:6.2f} ps per pop, {W * H / us / 1e3:6.1f} Grays/s")
