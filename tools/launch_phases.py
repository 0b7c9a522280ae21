"""tools/launch_phases.py [TREE] [--once] -- where the wall time of a short timed region goes: 20 plain single-frame launches of config 2
between two synchronisations, host clock after every call (median over 30 repetitions).  TREE = another checkout to load the
package from (A/B between rounds)."""
import os, sys, time
ONCE = "--once" in sys.argv          # as bench.py does it: ramp and warm-up with the per-launch events on, then ONE timed region
argv = [a for a in sys.argv[1:] if a != "--once"]
ROOT = os.path.abspath(argv[0]) if argv else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ray_tracing_octrees_amd as rto
from ray_tracing_octrees_amd import hip

ctx = rto.Context(0)
g = rto.VoxelGrid.test_sphere(256)
ctx.build_octree(g.data, g.min, g.voxelSize)
cam = rto.Camera(0.5, 0.7, 1.8)
W, H = 1920, 1080
fr = hip.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
if not ONCE:
    ctx.timing_begin(-1)
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    for _ in range(20):
        ctx.render_device(fr, out.data_ptr(), None, stream.cuda_stream)
    torch.cuda.synchronize()
K = 20
rows = []
if ONCE:
    for _ in range(5):
        ctx.render_device(fr, out.data_ptr(), None, stream.cuda_stream)
    torch.cuda.synchronize()
    ctx.timing_begin(-1)
for rep in range(1 if ONCE else 30):
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    ev_a.record(stream); t.append(time.perf_counter())
    for _ in range(K):
        ctx.render_device(fr, out.data_ptr(), None, stream.cuda_stream)
    t.append(time.perf_counter())
    ev_b.record(stream); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    rows.append([(t[i + 1] - t[i]) * 1e6 for i in range(4)] + [(t[-1] - t[0]) * 1e6, ev_a.elapsed_time(ev_b) * 1e3])
m = np.median(np.array(rows), axis=0)
print(f"{ROOT}: record {m[0]:.1f} us, {K} launches {m[1]:.1f} us (host), record {m[2]:.1f} us, synchronize {m[3]:.1f} us; wall {m[4]:.1f} us, "
      f"events {m[5]:.1f} us, wall - events {m[4] - m[5]:.1f} us", flush=True)
