"""Config 4 (Calgary as shipped, oblique camera, 1080p), tile by tile, from the oracle's per-pixel trip counts (orc.render_visits):
how many loop bodies the frame's waves issue, the lanes still walking at every trip depth, and what re-packing the survivors of the
long tiles into full waves after trip N would save at most.  Study aid; CPU box.   python tools/model/config4_tiles.py [out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import orc

cfg = sys.argv[2] if len(sys.argv) > 2 else "4"
W, H = 1920, 1080
if cfg == "4":
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_scene_cache.npz"))
    dims = tuple(int(x) for x in z["dims"])
    data = np.unpackbits(z["packed"])[: dims[0] * dims[1] * dims[2]].reshape(dims[2], dims[1], dims[0])
    g = orc.Grid(dims, z["min"].astype(np.float32), np.float32(z["voxel"]), data)
    cam = orc.Camera(0.6, 0.5, 3500.0)
else:
    g = orc.test_sphere_grid(256)
    cam = orc.Camera(0.5, 0.7, 1.8)
nodes = orc.build_flat_octree(g)
import ctypes as C
L = orc.lib()
v = np.zeros((H, W), np.int32)
f32 = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(C.c_void_p)
nodes = np.ascontiguousarray(nodes)
L.orc_render_visits.restype = None
L.orc_render_visits(C.c_void_p(nodes.ctypes.data), C.c_int64(len(nodes)), f32(g.min), C.c_float(float(g.voxel_size)), f32(cam.get_view()), f32(cam.get_pos()),
                    C.c_float(W / H), C.c_float(45.0), C.c_int(W), C.c_int(H), C.c_void_p(v.ctypes.data))
t = v.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)          # (tiles, lanes)
mx = t.max(axis=1)
live = mx > 1                                                                    # more than the root's own trip
tl = t[live]
mxl = mx[live]
bodies = int(mxl.sum())
lane_trips = int(tl.sum())
res = {"config": cfg, "tiles": int(len(t)), "live_tiles": int(live.sum()), "loop_bodies": bodies, "lane_trips": lane_trips,
       "lane_utilisation_of_the_loop": lane_trips / (64.0 * bodies), "trips_of_the_longest_tile": int(mxl.max()),
       "trip_percentiles_of_live_tiles": {str(p): float(np.percentile(mxl, p)) for p in (50, 90, 99, 100)}}
print(json.dumps(res))
# lanes still walking at trip depth k, over the tiles still running at k
depths = [1, 4, 8, 12, 16, 24, 32, 48, 64, 96]
prof = []
for k in depths:
    run = mxl > k
    if not run.any():
        break
    alive = (tl[run] > k).sum()
    prof.append({"trip": k, "tiles_running": int(run.sum()), "lanes_walking_per_tile": float(alive / run.sum()),
                 "share_of_all_bodies_beyond": float(np.maximum(mxl - k, 0).sum() / bodies)})
    print(f"trip {k:3d}: tiles running {run.sum():6d}  lanes walking {alive / run.sum():5.1f} of 64   bodies beyond this depth {np.maximum(mxl - k, 0).sum() / bodies:.3f} of all")
res["by_depth"] = prof
# re-packing: after trip N the survivors of all tiles are dealt into full waves (free of charge, perfect balance): bodies then
for N in (8, 12, 16, 24, 32):
    head = np.minimum(mxl, N).sum()
    surv = np.maximum(tl - N, 0)                       # remaining trips per surviving lane
    tail_lane_trips = surv.sum()
    # survivors packed 64 to a wave, sorted by remaining length (best case): bodies = sum over packed waves of their longest lane
    rem = np.sort(surv[surv > 0])[::-1]
    packed = rem[::64].sum()
    print(f"re-pack after trip {N:2d}: bodies {head + packed} = {100.0 * (head + packed) / bodies:.1f} % of today's ({len(rem)} surviving rays in {int(np.ceil(len(rem) / 64))} waves)")
    res.setdefault("repack", []).append({"after_trip": N, "bodies": int(head + packed), "of_today": float((head + packed) / bodies), "surviving_rays": int(len(rem))})
# the two floors of a frame
per_body_us = 120 * 3.35 / 2.4e3        # 120 VALU instructions at 3.35 cycles on one SIMD, microseconds
print(f"issue floor: {bodies} bodies x {per_body_us:.3f} us / 1024 SIMDs = {bodies * per_body_us / 1024:.1f} us (+ prologues / epilogues of {live.sum()} live and the work-less waves)")
print(f"longest tile: {mxl.max()} dependent trips x ~0.4 us = {mxl.max() * 0.4:.1f} us")
res["issue_floor_us_loop_only"] = bodies * per_body_us / 1024
res["longest_tile_floor_us"] = float(mxl.max() * 0.4)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
