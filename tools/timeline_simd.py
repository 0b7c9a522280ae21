"""Per-SIMD view of one frame's wave timeline (packed kernel): which waves shared a SIMD with the wave that ended last, and how the
work was spread over the SIMDs.  Run on the GPU box.  HW_ID (gfx9): wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ray_tracing_octrees_amd as rto

dim, W, H = 256, 1920, 1080
config = next((a.split("=")[1] for a in sys.argv if a.startswith("config=")), "2")      # config=4: Calgary as shipped, oblique camera
ctx = rto.Context(0)
theta = float(next((a.split("=")[1] for a in sys.argv if a.startswith("theta=")), 0.5))
if config == "4":
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_scene_cache.npz"))
    dims = tuple(int(x) for x in z["dims"])
    data = np.unpackbits(z["packed"])[: dims[0] * dims[1] * dims[2]].reshape(dims[2], dims[1], dims[0])
    g = rto.VoxelGrid.from_array(data, z["min"].astype(np.float32), np.float32(z["voxel"]))
    cam = rto.Camera(0.6, 0.5, 3500.0)
else:
    g = rto.VoxelGrid.test_sphere(dim)
    cam = rto.Camera(theta, 0.7, 1.8)
ctx.build_octree(g.data, g.min, g.voxelSize)
f = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
if "nomask" in sys.argv:
    ctx.debug_set_tile_mask(False)
learn = next((float(a.split("=")[1]) for a in sys.argv if a.startswith("learn=")), None)    # learn=<theta>: the launch-order table comes from THAT camera's costs
if learn is not None:
    camL = rto.Camera(learn, 0.7, 1.8)
    fL = rto.make_frame(camL.getView(), camL.getPos(), W / H, 45.0, W, H)
    for _ in range(12):
        ctx.render_host(fL)
else:
    for _ in range(12):
        ctx.render_host(f)
rec = ctx.debug_timeline(f)
rec = ctx.debug_timeline(f)
tiles = np.arange(len(rec))
keep = (rec[:, 0] != 0) | (rec[:, 1] != 0)
rec, tiles = rec[keep], tiles[keep]
t0 = (rec[:, 0].astype(np.uint32).astype(np.uint64) | (rec[:, 1].astype(np.uint32).astype(np.uint64) << 32)).astype(np.int64)
t1 = (rec[:, 2].astype(np.uint32).astype(np.uint64) | (rec[:, 3].astype(np.uint32).astype(np.uint64) << 32)).astype(np.int64)
base = t0.min()
s, e = (t0 - base) / 100.0, (t1 - base) / 100.0
it, act, slot, xcc, hw = rec[:, 4], rec[:, 7] & 0xff, rec[:, 7] >> 8, rec[:, 6], rec[:, 5].astype(np.uint32)
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
key = ((xcc.astype(np.int64) * 8 + se) * 2 + sh) * 16 + cu
skey = key * 4 + simd
print(f"waves {len(rec)}, kernel span {e.max():.1f} us, distinct CUs {len(np.unique(key))}, distinct SIMDs {len(np.unique(skey))}")
tx = (W + 7) // 8
# per-SIMD totals
us, inv = np.unique(skey, return_inverse=True)
tot_it = np.bincount(inv, weights=it); cnt = np.bincount(inv); last = np.zeros(len(us)); np.maximum.at(last, inv, e)
print("per SIMD: waves mean %.1f max %d | iterations mean %.0f max %d | last end mean %.1f p90 %.1f max %.1f" % (cnt.mean(), cnt.max(), tot_it.mean(), tot_it.max(), last.mean(), np.percentile(last, 90), last.max()))
jout = next((a.split("=")[1] for a in sys.argv if a.startswith("json=")), None)
if jout:
    import json
    livew = it > 0
    busy = np.zeros(len(us)); np.add.at(busy, inv, (e - s) * (it > 0))
    json.dump({"config": config, "waves": int(len(rec)), "live_waves": int(livew.sum()), "kernel_span_us": float(e.max()), "simds": int(len(us)),
               "per_simd": {"waves_mean": float(cnt.mean()), "waves_max": int(cnt.max()), "trips_mean": float(tot_it.mean()), "trips_max": int(tot_it.max()),
                            "trips_p10": float(np.percentile(tot_it, 10)), "trips_p90": float(np.percentile(tot_it, 90)),
                            "last_end_us_mean": float(last.mean()), "last_end_us_p10": float(np.percentile(last, 10)), "last_end_us_max": float(last.max())},
               "longest_wave": {"trips": int(it.max()), "duration_us": float((e - s)[np.argmax(it)]), "end_us": float(e[np.argmax(it)])},
               "trips_total": int(it.sum()),
               "issue_floor_us": {"loop": float(tot_it.mean() * 120 * 3.35 / 2.4e3), "loop_of_the_busiest_simd": float(tot_it.max() * 120 * 3.35 / 2.4e3),
                                  "how": "trips per SIMD x 120 VALU instructions x 3.35 cycles (measured issue cost of the loop's mix) / 2.4 GHz; prologues and epilogues (~330 instructions per live wave) come on top"},
               "live_waves_ended_by_us": {str(q): float(np.percentile(e[livew], q)) for q in (50, 90, 99, 100)}}, open(jout, "w"), indent=1)
order = np.argsort(-last)[:3]
for k in order:
    m = np.nonzero(inv == k)[0]
    m = m[np.argsort(s[m])]
    print(f"SIMD {us[k]}: {len(m)} waves, {int(tot_it[k])} iterations, last end {last[k]:.1f}")
    for i in m:
        print(f"    tile ({tiles[i] % tx:3d},{tiles[i] // tx:3d}) start {s[i]:6.2f} end {e[i]:6.2f} dur {e[i]-s[i]:6.2f} iters {it[i]:3d} act {act[i]:2d} wave_id {hw[i] & 15}")
# correlation: duration vs iterations for waves started in the first microsecond
first = s < 1.0
print("first-batch waves: %d; dur/iter percentiles:" % first.sum(), np.percentile((e - s)[first] / np.maximum(it[first], 1), [10, 50, 90]).round(2))
for lo, hi in ((0, 8), (8, 16), (16, 32), (32, 48), (48, 80)):
    m = first & (it >= lo) & (it < hi)
    if m.any():
        print(f"  iters [{lo},{hi}): {m.sum():5d} waves, duration mean {(e - s)[m].mean():6.2f} p90 {np.percentile((e - s)[m], 90):6.2f} max {(e - s)[m].max():6.2f}")
if "brief" in sys.argv:
    late = np.argsort(-e)[:12]
    print("the 12 waves that end last:")
    for i in late:
        print(f"    slot {slot[i]:5d} tile ({tiles[i] % tx:3d},{tiles[i] // tx:3d}) start {s[i]:6.2f} end {e[i]:6.2f} iters {it[i]:3d}")
    for lo, hi in ((0, 1024), (1024, 2048), (2048, 4096), (4096, 6144), (6144, 20000)):
        m = (slot >= lo) & (slot < hi)
        if m.any(): print(f"  slots [{lo},{hi}): start mean {s[m].mean():6.2f} max {s[m].max():6.2f}, iters mean {it[m].mean():5.1f} max {it[m].max()}, end max {e[m].max():6.2f}")
    sys.exit(0)
# where the dispatcher put the first workgroups: slot -> (xcc, se, sh, cu, simd, wave_id)
o = np.argsort(slot)
print("slot -> xcc se sh cu simd wave | start us  (first 48 slots, then every 256th)")
for i in list(o[:48]) + list(o[256:6144:256]):
    print(f"  slot {slot[i]:5d} blk {slot[i] // 4:4d}: xcc {xcc[i]} se {se[i]} sh {sh[i]} cu {cu[i]:2d} simd {simd[i]} wave {hw[i] & 15} start {s[i]:5.2f} iters {it[i]}")
# is the SIMD of the first 6144 slots a function of slot % 1024?
first = slot < 6144
import collections
by = collections.defaultdict(set)
for i in np.nonzero(first)[0]:
    by[int(slot[i]) % 1024].add(int(skey[i]))
print("slots < 6144: residues (mod 1024) whose waves all landed on ONE simd:", sum(1 for v in by.values() if len(v) == 1), "of", len(by))
byb = collections.defaultdict(set)
for i in np.nonzero(first)[0]:
    byb[(int(slot[i]) // 4) % 256].add(int(key[i]))
print("blocks: residues (mod 256) whose blocks all landed on ONE cu:", sum(1 for v in byb.values() if len(v) == 1), "of", len(byb))
