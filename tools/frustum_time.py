"""tools/frustum_time.py -- cost of rto_update_frustum (row A7: the GPU form of the reference's per-node frustum loop + compaction)
at config 2 / config 5 sizes: wall time per call and, under rocprofv3 --kernel-trace --stats, its kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracing_octrees_amd as rto

ctx = rto.Context(0)
for dim in (256, 512):
    g = rto.VoxelGrid.test_sphere(dim)
    ctx.build_octree(g.data, g.min, g.voxelSize)
    cam = rto.Camera(0.5, 0.7, 1.8)
    view = cam.getView()
    for _ in range(5):
        ctx.update_frustum(view, 45.0, 16 / 9, True)
    ts = []
    for _ in range(50):
        t = time.perf_counter()
        ctx.update_frustum(view, 45.0, 16 / 9, True)
        ts.append(time.perf_counter() - t)
    ts.sort()
    print(f"{dim}^3 ({ctx.info().num_nodes} nodes): rto_update_frustum median {ts[len(ts) // 2] * 1e3:.4f} ms, min {ts[0] * 1e3:.4f} ms", flush=True)
    ctx.update_frustum(view, 45.0, 16 / 9, False)
