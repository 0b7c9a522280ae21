"""Per-iteration latency of the packed kernel when waves run (almost) alone: tiny zoomed frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ray_tracing_octrees_amd as rto
from ray_tracing_octrees_amd import _build
if os.environ.get("RTO_LIB"):
    _build.LIB_HIP = os.environ["RTO_LIB"]

g = rto.VoxelGrid.test_sphere(256)
root = rto.createOctreeFromVoxelGrid(g)
nodes = root.flatten()
ctx = rto.Context(0)
ctx.upload_octree(nodes, g.min, g.voxelSize)
if len(sys.argv) > 1 and sys.argv[1] == "v1":
    ctx.set_kernel(rto.KERNEL_PACKED_V1)
cam = rto.Camera(0.5, 0.7, 1.8)
for (W, H, fov, tgt) in ((8, 8, 1.0, (0.0, 0.395, 0.0)), (64, 64, 3.0, (0.0, 0.39, 0.0)), (64, 64, 6.0, (0.0, 0.3, 0.0)), (256, 256, 10.0, (0.0, 0.35, 0.0))):
    cam.setTarget(np.array(tgt, np.float32))
    f = rto.make_frame(cam.getView(), cam.getPos(), W / H, fov, W, H)
    for _ in range(2):
        ctx.render_host(f)
    rec = ctx.debug_timeline(f)
    rec = ctx.debug_timeline(f)
    t0 = (rec[:, 0].astype(np.uint32).astype(np.uint64) | (rec[:, 1].astype(np.uint32).astype(np.uint64) << 32)).astype(np.int64)
    t1 = (rec[:, 2].astype(np.uint32).astype(np.uint64) | (rec[:, 3].astype(np.uint32).astype(np.uint64) << 32)).astype(np.int64)
    dur = (t1 - t0) / 100.0
    it = rec[:, 4]
    st = ctx.frame_stats(f)
    m = it >= 8
    print(f"{W}x{H} fov {fov}: waves {len(rec)} hits {st['hits']} pops/ray {st['pops']/st['rays']:.1f} iters max {it.max()} mean {it.mean():.1f} | "
          f"dur max {dur.max():.2f} us | us/iter (it>=8): " + (f"mean {(dur[m]/it[m]).mean():.3f} min {(dur[m]/it[m]).min():.3f}" if m.any() else "n/a")
          + f" | lanes active mean {rec[:,7].mean():.1f}")
