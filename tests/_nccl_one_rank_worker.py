"""Worker for test_tile_split_over_rccl_with_one_rank: the product's device-to-device path (HipBackend + the nccl
backend, async gathers, batches, three pipelines on three streams) with a ONE-rank RCCL group on GPU 0, compared with
the oracle.  Everything except traffic between GPUs is exercised; run as its own process (it owns a process group)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ray_tracing_octrees_amd as rto  # noqa: E402
from oracle import orc  # noqa: E402
from ray_tracing_octrees_amd import tilesplit  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29597")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    W, H = 640, 360
    og = orc.test_sphere_grid(64)
    nodes = orc.build_flat_octree(og)
    ctx = rto.Context(0)
    ctx.upload_octree(nodes, og.min, og.voxel_size)
    cams = [orc.Camera(0.4 + 0.2 * i, 0.7 + 0.03 * i, 1.8) for i in range(12)]
    frames = [rto.make_frame(c.get_view(), c.get_pos(), W / H, 45.0, W, H) for c in cams]
    wants = [orc.render(nodes, og.min, og.voxel_size, c.get_view(), c.get_pos(), W / H, 45.0, W, H)[0] for c in cams]
    ok = True
    for payload in ("shade", "rgba"):
        npipe, fpg = 3, 2
        R = [tilesplit.TileSplitRenderer(tilesplit.HipBackend(ctx), 0, 1, band_rows=16, payload=payload, force_collective=True)
             for _ in range(npipe)]
        S = [torch.cuda.Stream() for _ in range(npipe)]
        torch.cuda.set_stream(S[0])
        got = {}
        for j in range(6):                                      # 6 batches of 2 frames over 3 pipelines, twice around
            with torch.cuda.stream(S[j % npipe]):
                out = R[j % npipe].submit_batch(frames[2 * j: 2 * j + 2])
                if out is not None:
                    torch.cuda.current_stream().synchronize()
                    for f, t in enumerate(out):
                        got[2 * (j - npipe) + f] = t.cpu().numpy().copy()
        for i in range(npipe):
            with torch.cuda.stream(S[i]):
                out = R[i].flush_batch()
                torch.cuda.current_stream().synchronize()
                j = 3 + i
                for f, t in enumerate(out):
                    got[2 * j + f] = t.cpu().numpy().copy()
        for k in range(12):
            ok = ok and k in got and got[k].tobytes() == wants[k].tobytes()
        one = R[0].render(frames[5])                             # one frame per collective on the same object
        torch.cuda.synchronize()
        ok = ok and one.cpu().numpy().tobytes() == wants[5].tobytes()
    dist.destroy_process_group()
    print("nccl one-rank worker:", "ok" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
