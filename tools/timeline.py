"""Per-wave timeline analysis of one frame (packed kernel). Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ray_tracing_octrees_amd as rto

dim, W, H = 256, 1920, 1080
g = rto.VoxelGrid.test_sphere(dim)
root = rto.createOctreeFromVoxelGrid(g)
nodes = root.flatten()
cam = rto.Camera(0.5, 0.7, 1.8)
if "miss" in sys.argv:
    cam.setTarget(np.array([5.0, 5.0, 9.0], np.float32))
ctx = rto.Context(0)
ctx.upload_octree(nodes, g.min, g.voxelSize)
f = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
if len(sys.argv) > 1 and sys.argv[1] == "v1":
    ctx.set_kernel(rto.KERNEL_PACKED_V1)
for _ in range(3):
    ctx.render_host(f)
rec = ctx.debug_timeline(f)
rec = ctx.debug_timeline(f)
all_tiles = len(rec)
rec = rec[(rec[:, 0] != 0) | (rec[:, 1] != 0)]          # tiles outside the root rectangle's box get no wave
print(f"tiles {all_tiles}, waves with a tile {len(rec)}")
t0 = (rec[:, 0].astype(np.uint32).astype(np.uint64) | (rec[:, 1].astype(np.uint32).astype(np.uint64) << 32)).astype(np.int64)
t1 = (rec[:, 2].astype(np.uint32).astype(np.uint64) | (rec[:, 3].astype(np.uint32).astype(np.uint64) << 32)).astype(np.int64)
base = t0.min()
s = (t0 - base) / 100.0   # us (100 MHz clock)
e = (t1 - base) / 100.0
it = rec[:, 4]; act = rec[:, 7] & 0xff; xcc = rec[:, 6]; hwid = rec[:, 5].astype(np.uint32)
dur = e - s
print(f"waves {len(rec)}  kernel span {e.max():.1f} us  (last start {s.max():.1f} us)")
print("duration us: mean %.2f  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
print("iterations : mean %.1f  p50 %d  p90 %d  p99 %d  max %d" % (it.mean(), *np.percentile(it, [50, 90, 99]).astype(int), it.max()))
print("sum wave-us %.0f -> avg concurrent waves %.0f (of %d slots)" % (dur.sum(), dur.sum() / e.max(), 256 * 32))
nz = it > 0
print("us per iteration (waves with >=10 iterations): mean %.3f" % (dur[it >= 10] / it[it >= 10]).mean())
# concurrency over time
edges = np.linspace(0, e.max(), 17)
for a, b in zip(edges[:-1], edges[1:]):
    live = ((s < b) & (e > a)).sum()
    started = ((s >= a) & (s < b)).sum()
    work = (np.minimum(e, b) - np.maximum(s, a)).clip(0).sum() / (b - a)
    print(f"  [{a:6.1f},{b:6.1f}) us: started {started:6d}  avg-resident {work:7.0f}  heavy(it>=20) resident {(((s < b) & (e > a)) & (it >= 20)).sum():5d}")
# the 10 longest waves
idx = np.argsort(-dur)[:10]
tx = 240
for i in idx:
    print(f"  tile {i} (tx {i % tx}, ty {i // tx}) start {s[i]:.1f} dur {dur[i]:.1f} iters {it[i]} act {act[i]} xcc {xcc[i]}")
# per-XCD finish time
for x in range(8):
    m = xcc == x
    print(f"  xcc {x}: waves {m.sum()} sum-iters {it[m].sum()} last end {e[m].max():.1f}")
