import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29577"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
src=torch.ones(1<<20, device="cuda"); dst=[torch.zeros_like(src)]
s=torch.cuda.Stream()
with torch.cuda.stream(s):
    dist.gather(src,dst,dst=0); torch.cuda.synchronize()
    try:
        g=torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(4):
                dist.all_reduce(src)
                dist.gather(src,dst,dst=0)
        g.replay(); torch.cuda.synchronize()
        print("capture of NCCL ops (world 1): ok", float(dst[0][0]))
    except Exception as e:
        print("capture of NCCL ops failed:", type(e).__name__, str(e)[:200])
dist.destroy_process_group()
