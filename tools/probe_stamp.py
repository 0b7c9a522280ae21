"""tools/probe_stamp.py -- where k_probe_skip spends a call: build `tools/build_variants.sh pstamp "-DRTO_PROBE_STAMP"`, then on the GPU box
    RTO_HIP_LIB=build/variants/librto_hip_pstamp.so python tools/probe_stamp.py
(the stamped kernel leaves the three durations behind the skip value)."""
import os, sys
ROOT="/root/repo" if os.path.isdir("/root/repo/tools") else os.getcwd()
sys.path.insert(0, ROOT)
import numpy as np, torch
import ray_tracing_octrees_amd as rto
ctx = rto.Context(0)
for dim in (256, 512):
    g = rto.VoxelGrid.test_sphere(dim)
    ctx.build_octree(g.data, g.min, g.voxelSize)
    cam = rto.Camera(0.5, 0.7, 1.8)
    d = torch.zeros(8, dtype=torch.float32, device="cuda")
    for _ in range(20):
        ctx.probe_skip_device(cam.getView(), cam.getPos(), 16 / 9, d.data_ptr())
    ctx.synchronize()
    v = d.cpu().numpy()
    print(dim, "set-up (table, rays, barrier) %.2f us, traversal %.2f us, rank + blend %.2f us" % (v[1], v[2], v[3]))
