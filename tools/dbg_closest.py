"""tools/dbg_closest.py -- the closest-hit mode's near-first kernel against the pop-order kernel and the oracle on the golden cameras (which pixels differ)."""
import os, sys, importlib.util
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ray_tracing_octrees_amd as rto
from ray_tracing_octrees_amd import hip
from oracle import orc
spec = importlib.util.spec_from_file_location("m", os.path.join(ROOT, "tests", "golden", "make_golden_glsl.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
ctx = rto.Context(0)
for scene in m.CASES:
    g, nodes = m.scene(scene)
    ctx.upload_octree(nodes, g.min, g.voxel_size)
    W, H, cams = m.CASES[scene]
    for i, cs in enumerate(cams):
        view, pos, fov = m.camera(cs)
        f = rto.make_frame(view, pos, W / H, fov, W, H)
        want, _ = orc.render_closest(nodes, g.min, g.voxel_size, view, pos, W / H, fov, W, H)
        ctx.set_kernel(hip.KERNEL_AUTO); a = ctx.render_closest_host(f)
        ctx.set_kernel(hip.KERNEL_PACKED_V1); b = ctx.render_closest_host(f)
        da = np.argwhere((a.view(np.uint32) != want.view(np.uint32)).any(axis=2)); db = np.argwhere((b.view(np.uint32) != want.view(np.uint32)).any(axis=2))
        print(scene, "camera", i, "near-first differs at", [tuple(int(v) for v in p) for p in da[:4]], "pop-order differs at", [tuple(int(v) for v in p) for p in db[:4]])
    ctx.set_kernel(hip.KERNEL_AUTO)
