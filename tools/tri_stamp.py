"""Where config 5's frame spends its TIME, phase by phase, with the lanes that had work in each phase: an A/B build with
-DRTO_TRI_STAMP puts s_memtime stamps at the phase boundaries of trace_tile_lean_triangles and leaves 24 words per tile.
    tools/build_variants.sh tristamp "-DRTO_TRI_STAMP"
    RTO_HIP_LIB=build/variants/librto_hip_tristamp.so python tools/tri_stamp.py [out.json [dim W H [theta phi r]]]     (GPU box)
Ticks are s_memtime's (100 MHz on gfx950: 10 ns) summed over the waves of one frame: a wave's ticks include the time the SIMD
spent issuing for its neighbours, so the SHARES are what the figures mean, not the absolute sums."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ray_tracing_octrees_amd as rto

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tri_stamp.json"
dim, W, H = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (512, 3840, 2160)
th, ph, rr = (float(sys.argv[5]), float(sys.argv[6]), float(sys.argv[7])) if len(sys.argv) > 7 else (0.5, 0.7, 1.8)
g = rto.VoxelGrid.test_sphere(dim)
ctx = rto.Context(0)
ctx.build_octree(g.data, g.min, g.voxelSize)
ctx.build_leaf_triangles(None)
cam = rto.Camera(th, ph, rr)
f = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
import torch
buf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
for _ in range(12):                                   # the launch order settles
    ctx.render_triangles_device(f, buf.data_ptr(), True)
ctx.synchronize()
ms = []
for _ in range(5):
    ctx.render_triangles_device(f, buf.data_ptr(), True)
    ctx.synchronize()
    ms.append(ctx.last_kernel_ms())
L = rto.hip.load()
tiles = ((W + 7) // 8) * ((H + 7) // 8)
L.rto_debug_steps_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
rec = np.zeros((tiles, 24), np.uint32)
rc = L.rto_debug_steps_buffer(ctx._h, rec.ctypes.data, rec.size)
assert rc == 0, rc
rec = rec[(rec[:, 23] >> 31) == 1].astype(np.int64)
rounds = rec[:, 23] & 0x7fffffff
lane_trips_max, wave_trips = rec[:, 21] & 0xffff, rec[:, 21] >> 16
chunks = rec[:, 22]
live = wave_trips > 0
names = ["prologue", "node loop", "round set-up (records, prefix sum, keys)", "pair chunks (dealing + Moeller-Trumbore)",
         "key read-back + pop-next", "ray end (accounting, shadow start)", "epilogue (cost, store, fill)"]
units = [1, None, 1, None, 1, 1, 1]                    # work units per entry other than "entered": trips / chunks below
ticks = rec[:, 0:21:3].sum(axis=0).astype(float)
enter = rec[:, 1:21:3].sum(axis=0).astype(float)
lanes = rec[:, 2:21:3].sum(axis=0).astype(float)
tot = ticks.sum()
res = {"scene": f"sphere {dim}^3, {W}x{H}, Camera({th},{ph},{rr}), shadow rays on", "kernel_ms_stamped_build": float(np.median(ms)),
       "waves": int(len(rec)), "live_waves": int(live.sum()), "rounds": int(rounds.sum()),
       "node_loop_bodies_issued": int(wave_trips.sum()), "trips_of_each_waves_busiest_lane": int(lane_trips_max.sum()),
       "chunks": int(chunks.sum()), "pairs": int(lanes[3]), "phases": []}
print(f"stamped frame {np.median(ms):.3f} ms; waves {len(rec)} (live {live.sum()}); rounds {rounds.sum()}; node loop bodies issued {wave_trips.sum()} "
      f"(busiest lanes' trips {lane_trips_max.sum()}); chunks {chunks.sum()}")
for i, n in enumerate(names):
    execs = {1: float(wave_trips.sum()), 3: float(chunks.sum())}.get(i, enter[i])
    util = lanes[i] / (64.0 * execs) if execs else 0.0
    e = {"phase": n, "share_of_wave_time": ticks[i] / tot, "ticks": ticks[i], "entered": enter[i], "executions": execs,
         "lanes_with_work_per_execution": lanes[i] / execs if execs else 0.0, "lane_utilisation": util,
         "ticks_per_execution": ticks[i] / execs if execs else 0.0}
    res["phases"].append(e)
    print(f"  {n:48s} share {e['share_of_wave_time']:.3f}  executions {execs:10.0f}  ticks/exec {e['ticks_per_execution']:7.2f}  lanes with work {e['lanes_with_work_per_execution']:5.1f} ({util:.2f})")
# live waves only (the work-less ones only run prologue + epilogue)
lt = rec[live][:, 0:21:3].sum(axis=0).astype(float)
dead = rec[~live][:, 0:21:3].sum()
res["share_of_wave_time_in_waves_without_work"] = float(dead / tot)
res["time_weighted_lane_utilisation"] = float(sum(p["share_of_wave_time"] * p["lane_utilisation"] for p in res["phases"]))
print(f"waves without work: {dead / tot:.3f} of all wave time; time-weighted lane utilisation {res['time_weighted_lane_utilisation']:.3f}")
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
