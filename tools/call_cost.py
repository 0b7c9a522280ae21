import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import ray_tracing_octrees_amd as rto
from ray_tracing_octrees_amd import hip
W,H=1920,1080
grid=rto.VoxelGrid.test_sphere(64); root=rto.createOctreeFromVoxelGrid(grid)
ctx=rto.Context(0); ctx.upload_octree(root.flatten(), grid.min, grid.voxelSize)
cam=rto.Camera(0.5,0.7,1.8); frame=rto.make_frame(cam.getView(), cam.getPos(), W/H, 45.0, W, H)
s=torch.cuda.Stream(); part=hip.Partition(8,0,16)
rows=ctx.partition_rows(frame, part)
buf=torch.empty((rows,W),device="cuda"); full=torch.empty((H,W,4),device="cuda"); g=torch.empty((8,1,rows,W),device="cuda")
sp=s.cuda_stream; bp=buf.data_ptr(); fp=full.data_ptr(); gp=g.data_ptr()
for name,fn in (("render_shade_device(part)", lambda: ctx.render_shade_device(frame, bp, part, sp)),
                ("assemble_batch_device", lambda: ctx.assemble_batch_device(frame, part, gp, 1, 0, True, fp, sp)),
                ("buf.data_ptr()", lambda: buf.data_ptr()),
                ("current_stream lookup", lambda: torch.cuda.current_stream().cuda_stream)):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    n=3000; t=time.perf_counter()
    for _ in range(n): fn()
    dt=time.perf_counter()-t; torch.cuda.synchronize()
    print(f"{name:32s} {dt/n*1e6:6.2f} us per call (host)")
