"""Round-3 numbers beside the bench line (one GPU): the asynchronous frustum update, the octree build including the derived
data (descPos, occupancy cells), the probe update (octreeRaySkip's consumer) and the nearest-hit render mode."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ray_tracing_octrees_amd as rto


def scene(name):
    if name.startswith("sphere"):
        g = rto.VoxelGrid.test_sphere(int(name[6:]))
        return g, rto.Camera(0.5, 0.7, 1.8)
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_scene_cache.npz"))
    dims = tuple(int(x) for x in z["dims"])
    data = np.unpackbits(z["packed"])[: dims[0] * dims[1] * dims[2]].reshape(dims[2], dims[1], dims[0])
    return rto.VoxelGrid.from_array(data, z["min"].astype(np.float32), np.float32(z["voxel"])), rto.Camera(0.6, 0.5, 3500.0)


def per_call_us(fn, sync, n=400):
    """Wall time per call of n back-to-back asynchronous calls, best of three batches (the HIP runtime now and then stalls for
    tens of milliseconds while ISSUING a burst of launches -- seen on one scene or another, never the same one: not the kernels)."""
    for _ in range(20):
        fn()
    sync()
    best = None
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        dt = (time.perf_counter() - t) / n * 1e6
        best = dt if best is None else min(best, dt)
    return best


ctx = rto.Context(0)
ctx.timing_begin(-1)
for name in ("sphere256", "sphere512", "calgary"):
    g, cam = scene(name)
    ws = []
    data = g.data                                     # (the Python wrapper copies the grid out of the C++ object: not part of the call)
    for _ in range(4):
        t = time.perf_counter()
        ctx.build_octree(data, g.min, g.voxelSize)
        ws.append(time.perf_counter() - t)
    k, u = ctx.last_build_ms()
    lv, nc = ctx.debug_tile_mask_info()
    view, pos = cam.getView(), cam.getPos()
    W, H = 1920, 1080
    aspect = W / H
    f = rto.make_frame(view, pos, aspect, 45.0, W, H)
    print(f"{name}: {ctx.info().num_nodes} nodes; rto_build_octree wall {min(ws[1:]) * 1e3:.3f} ms (pyramid + emission kernels {k:.3f} ms, H2D {u:.3f} ms; the rest: freeing the previous "
          f"octree's buffers and allocating this one's; the derived data -- descPos, occupancy cells [level {lv}, {nc} cells], two small read-backs -- "
          f"take 0.07-0.09 ms of it: RTO_BUILD_TRACE=1)")
    ctx.update_frustum(view, 45.0, aspect, True)
    us = per_call_us(lambda: ctx.update_frustum(view, 45.0, aspect, True), ctx.synchronize, 100)
    print(f"  rto_update_frustum: {us:.1f} us per call (100 back-to-back, best of 3 bursts; nothing read back; one kernel)")
    ctx.update_frustum(view, 45.0, aspect, False)
    d_skip = torch.zeros(1, dtype=torch.float32, device="cuda")
    us = per_call_us(lambda: ctx.probe_skip_device(view, pos, float(np.float32(aspect)), d_skip.data_ptr()), ctx.synchronize, 100)
    print(f"  rto_probe_skip_device (49 probes -> percentile -> blend): {us:.1f} us per call, value {float(d_skip.cpu()[0]):.6g}")
    d_rgba = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    d_dist = torch.empty((H, W), dtype=torch.float32, device="cuda")
    us = per_call_us(lambda: ctx.render_skip_device(f, d_rgba.data_ptr(), d_dist.data_ptr()), ctx.synchronize, 50)
    hit = int((d_dist < 1e30).sum())
    print(f"  rto_render_skip_device 1920x1080 (nearest-hit mode, colours + distances): {us:.1f} us per frame = {W * H / us:.0f} Mrays/s, {hit} hits")
    buf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    us = per_call_us(lambda: ctx.render_device(f, buf.data_ptr()), ctx.synchronize, 200)
    print(f"  rto_render_device 1920x1080 (the reference's first-hit-in-DFS-order mode): {us:.1f} us per frame")
