"""tools/orbit_stale.py -- what a one-frame-old launch order gets wrong when the camera moves (config 2, 0.01 rad per frame): the
order is the previous frame's tiles by descending cost; the first ~4096 positions start at once (the machine holds 4096 waves),
the rest as slots free.  Prints, per frame, the costs (trips) of THIS frame's tiles that the stale order starts late."""
import os, sys
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
prev = None
for i in range(12):
    cam = rto.Camera(0.5 + 0.01 * i, 0.7, 1.8)
    f = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
    ctx.render_device(f, buf.data_ptr()); ctx.synchronize()
    cost = ctx.debug_tile_cost().astype(np.int64).ravel()
    if prev is not None:
        order = np.argsort(-prev, kind="stable")                    # the table k_order_build makes of the previous frame's costs
        late = cost[order[4096:]]
        early_cheap = cost[order[:4096]]
        own = np.argsort(-cost, kind="stable")
        print(f"frame {i}: live tiles {int((cost > 0).sum())}, trips {int(cost[cost > 0].sum())}; started late by the stale order: "
              f"{int((late >= 10).sum())} tiles of >= 10 trips, {int((late >= 20).sum())} of >= 20, {int((late >= 40).sum())} of >= 40, max {int(late.max())}; "
              f"by its own order: {int((cost[own[4096:]] >= 10).sum())} of >= 10, max {int(cost[own[4096:]].max())}; "
              f"changed by > 5 trips: {int((np.abs(cost - prev) > 5).sum())} tiles, by > 20: {int((np.abs(cost - prev) > 20).sum())}", flush=True)
    prev = cost
