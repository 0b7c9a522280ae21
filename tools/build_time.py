"""rto_build_octree timing at 256^3 / 512^3 (both forms).  Under rocprofv3 --kernel-trace --stats this is the command behind
profiles/r02_build_kernel_stats.csv."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ray_tracing_octrees_amd as rto

ctx = rto.Context(0)
for dim in (256, 512):
    g = rto.VoxelGrid.test_sphere(dim)
    data = g.data
    for legacy in (False, True):
        ctx.debug_set_build_path(legacy)
        ks, ws = [], []
        for _ in range(6):
            t = time.perf_counter()
            ctx.build_octree(data, g.min, g.voxelSize)
            ws.append(time.perf_counter() - t)
            ks.append(ctx.last_build_ms())
        k = sorted(x[0] for x in ks[1:])[len(ks) // 2 - 1]
        u = sorted(x[1] for x in ks[1:])[len(ks) // 2 - 1]
        print(f"{dim}^3 {'level-by-level' if legacy else 'morton':15s}: device span after the upload {k:.3f} ms, H2D {u:.3f} ms, wall {min(ws[1:])*1e3:.3f} ms, nodes {ctx.info().num_nodes}")
ctx.debug_set_build_path(False)
