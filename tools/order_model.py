"""tools/order_model.py -- a list-scheduling model of one config-2 frame: workgroups of 4 consecutive launch slots go, in launch
order, to whichever of the machine's 1024 workgroup places (256 CUs x 4) is free first and hold it for the LONGEST of their four
tiles (the workgroup's LDS is released when its last wave ends).  Costs of camera B (0.01 rad from A) under B's own table and under
A's table: makespan in trips, and the place-time wasted beside a workgroup's longest wave."""
import heapq, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ray_tracing_octrees_amd as rto

g = rto.VoxelGrid.test_sphere(256)
ctx = rto.Context(0)
ctx.build_octree(g.data, g.min, g.voxelSize)
ctx.timing_begin(-1)
W, H = 1920, 1080
buf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
def costs(th):
    cam = rto.Camera(th, 0.7, 1.8)
    f = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
    for _ in range(3): ctx.render_device(f, buf.data_ptr())
    ctx.synchronize()
    return ctx.debug_tile_cost().astype(np.int64).ravel()
def table(c, G=16):
    b = np.where(c > 0, np.minimum(c, 62) + 1, np.where(c < 0, 1, 0))
    n = len(c); out = np.zeros(n, np.int64)
    for w in range(G):
        idx = np.arange(w, n, G)
        o = idx[np.argsort(-b[idx], kind="stable")]
        out[np.arange(len(o)) * G + w] = o
    return out
def model(order, c, places=1024, fixed=3.0, wave_places=False):
    c = np.maximum(c, 0).astype(float) + fixed
    heap = [0.0] * places
    busy = 0.0; waste = 0.0; end = 0.0
    step = 1 if wave_places else 4
    for i in range(0, len(order), step):
        t = c[order[i:i + step]]
        start = heapq.heappop(heap)
        d = t.max()
        heapq.heappush(heap, start + d)
        end = max(end, start + d)
        waste += (d * len(t) - t.sum())
    return end, waste
A, B = costs(0.5), costs(0.51)
for name, order in (("B under its own table", table(B)), ("B under A's table", table(A)), ("B in exact descending order", np.argsort(-B, kind="stable"))):
    e4, w4 = model(order, B)
    e1, _ = model(order, B, places=4096, wave_places=True)
    print(f"{name:30s}: workgroup places: makespan {e4:6.1f} trips, wasted beside the longest wave {w4:9.0f} wave-trips of {np.maximum(B, 0).sum():.0f}; "
          f"wave-granular places: makespan {e1:6.1f}", flush=True)
# where do B's longest tiles come from?  their cost one frame earlier, and the highest cost within 1 / 2 / 3 tiles of them then
tX = (W + 7) // 8
A2, B2 = A.reshape(-1, tX), B.reshape(-1, tX)
ys, xs = np.nonzero(B2 >= 40)
def nbmax(M, y, x, r): return int(M[max(y - r, 0):y + r + 1, max(x - r, 0):x + r + 1].max())
pos = np.empty(len(A), np.int64); pos[table(A)] = np.arange(len(A))
print(f"B has {len(ys)} tiles of >= 40 trips; of them {int(sum(A2[y, x] < 40 for y, x in zip(ys, xs)))} had < 40 in A, {int(sum(A2[y, x] < 20 for y, x in zip(ys, xs)))} had < 20")
for y, x in list(zip(ys, xs)):
    if A2[y, x] < 30:
        print(f"  tile ({x},{y}): B {B2[y, x]}, A {A2[y, x]} (position {pos[y * tX + x]} in A's table), A's max within 1 / 2 / 3 tiles: {nbmax(A2, y, x, 1)} / {nbmax(A2, y, x, 2)} / {nbmax(A2, y, x, 3)}")
