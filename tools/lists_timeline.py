"""tools/lists_timeline.py -- per-tile timeline of a config-2 frame rendered through the launch lists: when listed / raster slots start and end."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ray_tracing_octrees_amd as rto

dim, W, H = 256, 1920, 1080
g = rto.VoxelGrid.test_sphere(dim)
ctx = rto.Context(0)
ctx.build_octree(g.data, g.min, g.voxelSize)
cam = rto.Camera(0.5, 0.7, 1.8)
f = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
if "nolists" in sys.argv:
    ctx.debug_set_launch_lists(False)
for _ in range(12):
    ctx.render_host(f)
    ctx.synchronize()
rec = ctx.debug_timeline(f)
ctx.synchronize()
rec = ctx.debug_timeline(f)
info = ctx.debug_launch_lists_info()
listed = info["listed_slots"]
print("listed slots", listed, "host count", info["host_count"])
keep = (rec[:, 0] != 0) | (rec[:, 1] != 0)
rec = rec[keep]
t0 = (rec[:, 0].astype(np.uint32).astype(np.uint64) | (rec[:, 1].astype(np.uint32).astype(np.uint64) << 32)).astype(np.int64)
t1 = (rec[:, 2].astype(np.uint32).astype(np.uint64) | (rec[:, 3].astype(np.uint32).astype(np.uint64) << 32)).astype(np.int64)
base = t0.min()
s, e = (t0 - base) / 100.0, (t1 - base) / 100.0
it, slot = rec[:, 4], rec[:, 7] >> 8
print(f"tiles recorded {len(rec)}, span {e.max():.1f} us")
for name, m in (("listed", slot < listed), ("raster", slot >= listed)) if listed else (("all", slot >= 0),):
    if not m.any():
        continue
    d = (e - s)[m]
    print(f"{name:7s}: tiles {m.sum():6d} start p10/p50/p90/max {np.percentile(s[m], 10):6.1f} {np.percentile(s[m], 50):6.1f} {np.percentile(s[m], 90):6.1f} {s[m].max():6.1f} | end p50/p90/max {np.percentile(e[m], 50):6.1f} {np.percentile(e[m], 90):6.1f} {e[m].max():6.1f} | dur mean {d.mean():5.2f} p90 {np.percentile(d, 90):5.2f} max {d.max():5.2f} | trips>0 {int((it[m] > 0).sum())}")
    z = m & (it <= 0)
    if z.any():
        print(f"         tiles without trips: {z.sum()} dur mean {(e - s)[z].mean():5.2f} p90 {np.percentile((e - s)[z], 90):5.2f}")
for lo in range(0, int(e.max()) + 5, 5):
    a = ((s < lo + 5) & (e > lo)).sum()
    print(f"  t={lo:3d}..{lo+5:3d} us: tiles in flight {a:5d}, started {int(((s >= lo) & (s < lo + 5)).sum()):5d}, live started {int(((s >= lo) & (s < lo + 5) & (it > 0)).sum()):5d}")
