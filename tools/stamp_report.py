"""Diagnostic build (-DRTO_STAMP): cycle shares of the packed loop's segments for waves running alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ray_tracing_octrees_amd import _build
_build.LIB_HIP = os.path.join(ROOT, "tools", "ubench", "librto_stamp.so")
import ray_tracing_octrees_amd as rto
g = rto.VoxelGrid.test_sphere(256)
root = rto.createOctreeFromVoxelGrid(g)
ctx = rto.Context(0)
ctx.upload_octree(root.flatten(), g.min, g.voxelSize)
cam = rto.Camera(0.5, 0.7, 1.8)
for (W, H, fov, tgt) in ((8, 8, 1.0, (0.0, 0.395, 0.0)), (64, 64, 3.0, (0.0, 0.39, 0.0))):
    cam.setTarget(np.array(tgt, np.float32))
    f = rto.make_frame(cam.getView(), cam.getPos(), W / H, fov, W, H)
    ctx.render_host(f)
    rec = ctx.debug_timeline(f); rec = ctx.debug_timeline(f)
    it = rec[:, 4].astype(np.float64)
    names = ["slab math", "wait descriptor + decode", "[B] merge (LDS wait)", "[C] pop/stack/next", "back edge"]
    cols = [0, 1, 5, 6, 7]
    tot = 0
    print(f"{W}x{H}: waves {len(rec)}, iterations mean {it.mean():.1f}")
    for n, c in zip(names, cols):
        per = (rec[:, c].astype(np.uint32).astype(np.float64) / np.maximum(it, 1)).mean()
        tot += per
        print(f"   {n:28s} {per:8.1f} memtime ticks / iteration")
    print(f"   total {tot:.1f} ticks / iteration (s_memtime ticks at the shader clock; includes ~40 per stamp)")
