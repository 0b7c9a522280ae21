#!/usr/bin/env python3
"""Host-side cost of one torch.distributed.gather call on the nccl (=RCCL) backend, world size 1.
A lower bound of what every frame of the N>1 path pays on the host besides the kernel launches."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
for nbytes in (1 << 20, 4 << 20, 33 << 20):
    src = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda")
    dst = [torch.empty_like(src)]
    for mode in ("sync", "async"):
        for _ in range(20):
            dist.gather(src, dst, dst=0)
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 300
        prev = None
        for _ in range(n):
            if mode == "sync":
                dist.gather(src, dst, dst=0)
            else:
                w = dist.gather(src, dst, dst=0, async_op=True)
                if prev is not None:
                    prev.wait()
                prev = w
        t_issue = time.perf_counter() - t
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t
        print(f"{nbytes >> 20:3d} MiB {mode:5s}: host issue {t_issue / n * 1e6:7.1f} us/call, complete {t_all / n * 1e6:7.1f} us/call", flush=True)
dist.destroy_process_group()
