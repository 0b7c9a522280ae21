"""N4 measurement: GPU octree build time vs the host builders."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ray_tracing_octrees_amd as rto

ctx = rto.Context(0)
for dim in (64, 256, 512):
    g = rto.VoxelGrid.test_sphere(dim)
    data = g.data
    t = time.perf_counter(); root = rto.createOctreeFromVoxelGrid(g); t_host_build = time.perf_counter() - t
    t = time.perf_counter(); flat = root.flatten(); t_host_flat = time.perf_counter() - t
    rto.freeOctree(root)
    ks, us = [], []
    for _ in range(5):
        ctx.build_octree(data, g.min, g.voxelSize)
        k, u = ctx.last_build_ms(); ks.append(k); us.append(u)
    n = ctx.info().num_nodes
    vox = dim ** 3
    k = float(np.median(ks)); u = float(np.median(us))
    # algorithmic bytes: every voxel read once + the pyramid written once and read once (1/7 of the voxels each) + 60 B/node
    alg = vox + 2 * vox / 7 + n * 60
    print(f"sphere {dim}^3: nodes {n}; GPU build kernels {k:.3f} ms (H2D of {vox/1e6:.1f} MB voxels {u:.3f} ms) -> {vox/k/1e6:.1f} Gvoxel/s, "
          f"{alg/k/1e6:.1f} GB/s algorithmic; host C++ pyramid build {t_host_build*1e3:.1f} ms + flatten {t_host_flat*1e3:.1f} ms")
