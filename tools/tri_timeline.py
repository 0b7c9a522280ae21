"""Per-wave timeline of one config-5 frame (triangle kernel), from an A/B build with -DRTO_TRI_TIMELINE:
    tools/build_variants.sh tritl "-DRTO_TRI_TIMELINE"
    RTO_HIP_LIB=build/variants/librto_hip_tritl.so python tools/tri_timeline.py [dim W H]     (on the GPU box)
Prints how the frame's time is spent: when waves start and end, the longest waves, the load per SIMD."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ray_tracing_octrees_amd as rto

dim, W, H = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 3840, 2160)
g = rto.VoxelGrid.test_sphere(dim)
ctx = rto.Context(0)
ctx.build_octree(g.data, g.min, g.voxelSize)
ctx.build_leaf_triangles(None)
cam = rto.Camera(0.5, 0.7, 1.8)
f = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
import torch
buf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
for _ in range(12):                                   # the launch order settles
    ctx.render_triangles_device(f, buf.data_ptr(), True)
ctx.synchronize()
L = rto.hip.load()
tiles = ((W + 7) // 8) * ((H + 7) // 8)
L.rto_debug_steps_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
rec = np.zeros((tiles, 8), np.int32)
# the buffer is not cleared between frames: zero it through a first read, then render one frame
ctx.render_triangles_device(f, buf.data_ptr(), True)
ctx.synchronize()
rc = L.rto_debug_steps_buffer(ctx._h, rec.ctypes.data, rec.size)
assert rc == 0, rc
keep = (rec[:, 0] != 0) | (rec[:, 1] != 0)
rec = rec[keep]
u = lambda a, b: (a.astype(np.uint32).astype(np.uint64) | (b.astype(np.uint32).astype(np.uint64) << 32)).astype(np.int64)
t0, t1 = u(rec[:, 0], rec[:, 1]), u(rec[:, 2], rec[:, 3])
base = t0.min()
s, e = (t0 - base) / 100.0, (t1 - base) / 100.0
trips, rounds, chunks, hw, slot, xcc = rec[:, 4], rec[:, 5] & 0xfff, rec[:, 5] >> 12, rec[:, 6].astype(np.uint32), rec[:, 7] >> 4, rec[:, 7] & 15
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
skey = ((((xcc.astype(np.int64) * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
dur = e - s
print(f"waves recorded {len(rec)}, span {e.max():.1f} us; live waves (trips > 0) {(trips > 0).sum()}")
print("wave duration percentiles (live): ", np.percentile(dur[trips > 0], [50, 90, 99, 100]).round(1))
print("trips percentiles (live): ", np.percentile(trips[trips > 0], [50, 90, 99, 100]), " rounds: ", np.percentile(rounds[trips > 0], [50, 90, 99, 100]))
o = np.argsort(-dur)[:8]
for i in o:
    print(f"  longest: slot {slot[i]:6d} start {s[i]:7.1f} end {e[i]:7.1f} dur {dur[i]:6.1f} trips {trips[i]:4d} rounds {rounds[i]:3d}")
# how many waves are running over time
T = np.linspace(0, e.max(), 24)
running = [(int(((s <= t) & (e > t)).sum()), int(((s <= t) & (e > t) & (trips > 0)).sum())) for t in T]
print("time us -> waves running (live):", " ".join(f"{t:.0f}:{r[0]}({r[1]})" for t, r in zip(T, running)))
us, inv = np.unique(skey, return_inverse=True)
work = np.bincount(inv, weights=trips * 140.0 + rounds * 260.0)
last = np.zeros(len(us)); np.maximum.at(last, inv, e)
print(f"SIMDs {len(us)}: est. instructions per SIMD mean {work.mean():.0f} max {work.max():.0f} (max/mean {work.max() / work.mean():.2f}); last end mean {last.mean():.1f} p10 {np.percentile(last, 10):.1f} max {last.max():.1f}")
# start time of the costliest waves: are they launched first?
big = np.argsort(-(trips * 140 + rounds * 260))[:200]
print("the 200 costliest waves: start time percentiles", np.percentile(s[big], [50, 90, 100]).round(1), " slots percentiles", np.percentile(slot[big], [50, 90, 100]))
print("cost vs slot rank correlation:", np.corrcoef(np.argsort(np.argsort(slot)), np.argsort(np.argsort(-(trips * 140 + rounds * 260))))[0, 1].round(3))
# the tail: the waves that end last
o = np.argsort(-e)[:24]
print("last waves to end:")
for i in o:
    print(f"  slot {slot[i]:6d} start {s[i]:7.1f} end {e[i]:7.1f} dur {dur[i]:6.1f} trips {trips[i]:4d} rounds {rounds[i]:3d} chunks {chunks[i]:4d} simd {skey[i]}")
late = s > 350
print(f"waves that start after 350 us: {late.sum()}, live {(late & (trips > 0)).sum()}; their trips percentiles", np.percentile(trips[late & (trips > 0)], [50, 90, 100]) if (late & (trips > 0)).any() else "-")
# start time by slot decile
for q in range(0, 100, 10):
    m = (slot >= np.percentile(slot, q)) & (slot < np.percentile(slot, q + 10))
    print(f"  slot decile {q:2d}: start {s[m].mean():6.1f} us, live {int((trips[m] > 0).sum()):5d}, mean trips {trips[m].mean():5.1f} rounds {rounds[m].mean():5.1f} chunks {chunks[m].mean():5.1f} dur {dur[m].mean():6.1f}")
# which of the three counts explains a wave's duration?  least squares over the live waves that ran on a full machine
m = (trips > 0) & (s > 30) & (e < 380)
A = np.stack([trips[m], rounds[m], chunks[m], np.ones(m.sum())], axis=1).astype(np.float64)
coef, *_ = np.linalg.lstsq(A, dur[m], rcond=None)
print("duration ~ %.2f us x trips + %.2f x rounds + %.2f x chunks + %.1f   (live waves of the full-machine phase)" % tuple(coef))
