"""Worker for test_tilesplit_gloo.py: run under torch.distributed.run with the gloo backend (CPU).
The TileSplitRenderer logic (partition, single gather, rank-0 reassembly) is the product's; the pixel
producer is a CPU stand-in backed by the oracle, because there is no GPU here."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import orc  # noqa: E402
from ray_tracing_octrees_amd import hip, tilesplit  # noqa: E402


class OracleBackend:
    def __init__(self, nodes, gmin, voxel):
        self.nodes, self.gmin, self.voxel = nodes, gmin, voxel

    def empty(self, shape):
        return torch.full(shape, -7.0, dtype=torch.float32)

    def render_part(self, frame, part, out):
        view = np.array(list(frame.view), np.float32)
        pos = np.array(list(frame.cam_pos), np.float32)
        W, H = frame.width, frame.height
        rows = tilesplit.partition_row_map(H, part.num_parts, part.part, part.band_rows) if part else np.arange(H)
        full = np.zeros((H, W, 4), np.float32)
        for y in rows:      # only the rows this part owns are traced
            orc.render(self.nodes, self.gmin, self.voxel, view, pos, frame.aspect, frame.fov_deg, W, H, rows=(int(y), int(y) + 1), out=full)
        out[: len(rows)] = torch.from_numpy(full[rows])

    def assemble(self, frame, part0, gathered, out):
        for p in range(part0.num_parts):
            rows = tilesplit.partition_row_map(frame.height, part0.num_parts, p, part0.band_rows)
            out[torch.from_numpy(rows)] = gathered[p][: len(rows)]


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = orc.test_sphere_grid(16)
    nodes = orc.build_flat_octree(g)
    cam = orc.Camera(0.5, 0.7, 1.8)
    ok = True
    for (W, H, band) in ((64, 48, 8), (50, 37, 16), (40, 100, 24)):
        frame = hip.make_frame(cam.get_view(), cam.get_pos(), W / H, 45.0, W, H)
        r = tilesplit.TileSplitRenderer(OracleBackend(nodes, g.min, g.voxel_size), rank, world, band_rows=band)
        img = r.render(frame)
        rows_all = sum(tilesplit.partition_rows(H, world, p, band) for p in range(world))
        ok &= rows_all == H
        ok &= tilesplit.partition_rows(H, world, rank, band) == len(tilesplit.partition_row_map(H, world, rank, band))
        if rank == 0:
            want, _ = orc.render(nodes, g.min, g.voxel_size, cam.get_view(), cam.get_pos(), W / H, 45.0, W, H)
            ok &= img is not None and img.numpy().tobytes() == want.tobytes()
        else:
            ok &= img is None
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
