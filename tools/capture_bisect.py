"""Which stream-capture shapes of the Python tilesplit path survive hipStreamEndCapture (one-rank RCCL group, run on the GPU box).
Default: the four cases that pass.  `render_3streams_sync` -- kernel -> torch.distributed.gather -> assemble issued from THREE
forked streams of one capture -- aborted inside hipStreamEndCapture on ROCm 7.2 / torch 2.10 (round 1); it is not run unless
named explicitly (python tools/capture_bisect.py render_3streams_sync), each run of it costs a crashed process on the GPU box.
What the passing cases rule out: kernels alone on three forked streams (kernels_3streams), collectives alone or with
async_op + wait on one stream (nccl_1stream_*), the full render -> gather -> assemble chain on one stream (render_1stream).
What is left: ProcessGroupNCCL's single internal NCCL stream being joined into ONE capture from three forked branches.
The product no longer goes there: the N>1 path issues its collectives below the C boundary (rto_comm_*, own streams) and is
not captured into graphs; see DESIGN.md section 7."""
import os, sys, subprocess
ROOT = "/root/repo" if os.path.isdir("/root/repo") else os.getcwd()
CASE = sys.argv[1] if len(sys.argv) > 1 else None
if CASE is None:
    for c in ("kernels_3streams", "nccl_1stream_sync", "nccl_1stream_async", "render_1stream"):
        p = subprocess.run([sys.executable, "-X", "faulthandler", __file__, c], capture_output=True, text=True)
        tail = [l for l in (p.stdout + p.stderr).splitlines() if "case" in l or "Error" in l or "Segmentation" in l or "capture_end" in l]
        print(c, "rc", p.returncode, "|", " ; ".join(tail[-3:]), flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29611"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
import ray_tracing_octrees_amd as rto
from ray_tracing_octrees_amd import tilesplit, hip
from oracle import orc
W,H=640,360
og=orc.test_sphere_grid(64); nodes=orc.build_flat_octree(og); oc=orc.Camera(0.5,0.7,1.8)
ctx=rto.Context(0); ctx.upload_octree(nodes, og.min, og.voxel_size)
frame=rto.make_frame(oc.get_view(), oc.get_pos(), W/H, 45.0, W, H)
S=[torch.cuda.Stream() for _ in range(3)]
bufs=[torch.empty((H,W,4),device="cuda") for _ in range(3)]
src=torch.ones(1<<18, device="cuda"); dst=[torch.zeros_like(src)]
R=[tilesplit.TileSplitRenderer(tilesplit.HipBackend(ctx),0,1,payload="shade",force_collective=True) for _ in range(3)]
torch.cuda.set_stream(S[0])
for i in range(3):
    with torch.cuda.stream(S[i]):
        for _ in range(3):
            ctx.render_device(frame, bufs[i].data_ptr(), None, S[i].cuda_stream); R[i].render(frame); dist.gather(src,dst,dst=0)
torch.cuda.synchronize()
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=S[0]):
    if CASE=="kernels_3streams":
        for i in (1,2): S[i].wait_stream(S[0])
        for k in range(6):
            with torch.cuda.stream(S[k%3]): ctx.render_device(frame, bufs[k%3].data_ptr(), None, S[k%3].cuda_stream)
        for i in (1,2): S[0].wait_stream(S[i])
    elif CASE=="nccl_1stream_sync":
        for k in range(4): dist.gather(src,dst,dst=0)
    elif CASE=="nccl_1stream_async":
        for k in range(4):
            w=dist.gather(src,dst,dst=0,async_op=True); w.wait()
    elif CASE=="render_1stream":
        for k in range(4): R[0].render(frame)
    elif CASE=="render_3streams_sync":
        for i in (1,2): S[i].wait_stream(S[0])
        for k in range(6):
            with torch.cuda.stream(S[k%3]): R[k%3].render(frame)
        for i in (1,2): S[0].wait_stream(S[i])
g.replay(); torch.cuda.synchronize()
print("case", CASE, "ok")
