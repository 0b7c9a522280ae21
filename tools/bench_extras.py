#!/usr/bin/env python3
"""Timings of the rows beside the headline path (not the bench line): config 5 (leaf triangles + shadow ray) at
full size on one GPU, config 4/5 primary rays, the GPU octree build (N4) and octreeRaySkip (N1).
Also the command profiled for profiles/r01_extras_*: rocprofv3 --kernel-trace --stats -- python3 tools/bench_extras.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import ray_tracing_octrees_amd as rto

out = {}
ctx = rto.Context(0)
reps = int(os.environ.get("RTO_EXTRAS_REPS", "20"))


def kernel_ms(fn):
    fn()
    ts = []
    for _ in range(reps):
        fn()
        ts.append(ctx.last_kernel_ms())
    return float(np.median(ts))


# ---- config 5: 512^3 sphere, 3840x2160
g = rto.VoxelGrid.test_sphere(512)
t = time.perf_counter()
ctx.build_octree(g.data, g.min, g.voxelSize)
nodes = ctx.download_nodes()
info = ctx.info()
ctx.build_leaf_triangles(None)                      # GPU builder, voxels still resident from build_octree
ctx.build_leaf_triangles(None)
t_tris_gpu_ms = ctx.last_build_ms()[0]
tris, off = ctx.download_leaf_triangles()
t_tris = None
if os.environ.get("RTO_EXTRAS_HOST_TRIS"):           # the C++ host builder (localMC per leaf), for comparison: ~4.5 s
    t = time.perf_counter()
    rto.buildLeafTriangles(g, nodes)
    t_tris = time.perf_counter() - t
cam = rto.Camera(0.5, 0.7, 1.8)
W, H = 3840, 2160
f = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
import torch

buf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
c5 = {"nodes": int(info.num_nodes), "triangles": int(len(tris)), "leaf_triangle_build_gpu_ms": round(t_tris_gpu_ms, 3)}
if t_tris is not None:
    c5["leaf_triangle_build_host_s"] = round(t_tris, 3)
for kname, kernel in (("packed", rto.KERNEL_AUTO), ("generic", rto.KERNEL_GENERIC)):
    ctx.set_kernel(kernel)
    for shadow in (True, False):
        ms = kernel_ms(lambda: (ctx.render_triangles_device(f, buf.data_ptr(), shadow), ctx.synchronize()))
        c5[f"{kname}_{'shadow' if shadow else 'noshadow'}_ms"] = round(ms, 4)
ctx.set_kernel(rto.KERNEL_AUTO)
ms = kernel_ms(lambda: (ctx.render_device(f, buf.data_ptr()), ctx.synchronize()))
c5["primary_octree_rays_ms"] = round(ms, 4)
c5["primary_Mrays_per_s"] = round(W * H / ms / 1e3, 1)
out["config5_512_4k"] = c5

# ---- N4: GPU octree build
n4 = {}
for dim in (256, 512):
    gg = rto.VoxelGrid.test_sphere(dim) if dim != 512 else g
    ks = []
    for _ in range(5):
        ctx.build_octree(gg.data, gg.min, gg.voxelSize)
        ks.append(ctx.last_build_ms()[0])
    n4[f"sphere{dim}_kernels_ms"] = round(float(np.median(ks)), 3)
out["gpu_octree_build"] = n4

# ---- N1: octreeRaySkip, 49 probe rays (what the reference issues per frame) and 1M rays
g256 = rto.VoxelGrid.test_sphere(256)
ctx.build_octree(g256.data, g256.min, g256.voxelSize)
rng = np.random.default_rng(1)
pos = np.asarray(rto.Camera(0.5, 0.7, 1.8).getPos(), np.float32)
for n in (49, 1 << 20):
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    ctx.octree_ray_skip(pos, rd)
    t = time.perf_counter()
    for _ in range(5):
        ctx.octree_ray_skip(pos, rd)
    out[f"octree_ray_skip_{n}_rays_ms_incl_copies"] = round((time.perf_counter() - t) / 5 * 1e3, 3)

# ---- frustum update (A7): GPU cull + compaction + visibility masks vs the reference's CPU loop + SSBO re-upload
for dim, gg in ((256, g256), (512, g)):
    ctx.build_octree(gg.data, gg.min, gg.voxelSize)
    cam2 = rto.Camera(0.5, 0.7, 0.9)            # close enough that part of the sphere leaves the frustum
    view = cam2.getView()
    ctx.update_frustum(view, 45.0, 16 / 9, enable=True)
    ts = []
    for _ in range(10):
        t = time.perf_counter()
        ctx.update_frustum(view, 45.0, 16 / 9, enable=True)
        ts.append(time.perf_counter() - t)
    out[f"frustum_update_{dim}_ms_wall"] = round(float(np.median(ts)) * 1e3, 3)
    out[f"frustum_update_{dim}_visible_nodes"] = int(len(ctx.download_visible_nodes()))
    ctx.update_frustum(view, 45.0, 16 / 9, enable=False)

# ---- config 4: the shipped scene cache is not on the GPU box; calgary fixture lives in tests/golden
print(json.dumps(out))
