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


F = np.float32


def decode_shade(s: np.ndarray) -> np.ndarray:
    """numpy statement of the kernels' shade_color(): float32 operations, one rounding each."""
    s = s.astype(np.float32)
    px = np.empty(s.shape + (4,), np.float32)
    px[..., 0] = F(1.0) * s + F(0.1)
    px[..., 1] = F(0.8) * s + F(0.1)
    px[..., 2] = F(0.6) * s + F(0.1)
    px[..., 3] = F(1.0)
    px[s < 0] = (0.0, 0.0, 0.0, 1.0)
    return px


def encode_shade(rgba: np.ndarray) -> np.ndarray:
    """The oracle only produces pixels; recover a Lambert term that decodes to exactly those pixels
    (search a few ulps around r - 0.1).  Test scaffolding for the CPU stand-in only."""
    r = rgba[..., 0]
    out = np.full(r.shape, -1.0, np.float32)
    hit = ~((rgba[..., 0] == 0) & (rgba[..., 1] == 0) & (rgba[..., 2] == 0))
    todo = hit.copy()
    guess = np.maximum(r - F(0.1), F(0.0)).astype(np.float32)
    for k in range(-8, 9):
        cand = (guess.view(np.int32) + k).view(np.float32)
        cand = np.where(cand >= 0, cand, F(0.0)).astype(np.float32)
        ok = todo & (decode_shade(cand) == rgba).all(axis=-1)
        out[ok] = cand[ok]
        todo &= ~ok
    assert not todo.any(), "stand-in could not encode a pixel"
    return out


class OracleBackend:
    def __init__(self, nodes, gmin, voxel):
        self.nodes, self.gmin, self.voxel = nodes, gmin, voxel

    def empty(self, shape):
        return torch.full(shape, -7.0, dtype=torch.float32)

    def render_part(self, frame, part, out, payload="rgba"):
        view = np.array(list(frame.view), np.float32)
        pos = np.array(list(frame.cam_pos), np.float32)
        W, H = frame.width, frame.height
        rows = tilesplit.partition_row_map(H, part.num_parts, part.part, part.band_rows) if part else np.arange(H)
        full = np.zeros((H, W, 4), np.float32)
        for y in rows:      # only the rows this part owns are traced
            orc.render(self.nodes, self.gmin, self.voxel, view, pos, frame.aspect, frame.fov_deg, W, H, rows=(int(y), int(y) + 1), out=full)
        mine = full[rows]
        out[: len(rows)] = torch.from_numpy(encode_shade(mine) if payload == "shade" else mine)

    def assemble(self, frame, part0, gathered, out, payload="rgba", batch=1, index=0):
        for p in range(part0.num_parts):
            rows = tilesplit.partition_row_map(frame.height, part0.num_parts, p, part0.band_rows)
            g = gathered[p][index][: len(rows)]
            out[torch.from_numpy(rows)] = torch.from_numpy(decode_shade(g.numpy())) if payload == "shade" else g


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = orc.test_sphere_grid(16)
    nodes = orc.build_flat_octree(g)
    ok = True
    for (W, H, band, payload) in ((64, 48, 8, "shade"), (50, 37, 16, "rgba"), (40, 100, 24, "shade")):
        cam = orc.Camera(0.5, 0.7, 1.8)
        frame = hip.make_frame(cam.get_view(), cam.get_pos(), W / H, 45.0, W, H)
        r = tilesplit.TileSplitRenderer(OracleBackend(nodes, g.min, g.voxel_size), rank, world, band_rows=band, payload=payload)
        img = r.render(frame)
        rows_all = sum(tilesplit.partition_rows(H, world, p, band) for p in range(world))
        ok &= rows_all == H
        ok &= tilesplit.partition_rows(H, world, rank, band) == len(tilesplit.partition_row_map(H, world, rank, band))
        if rank == 0:
            want, _ = orc.render(nodes, g.min, g.voxel_size, cam.get_view(), cam.get_pos(), W / H, 45.0, W, H)
            ok &= img is not None and img.numpy().tobytes() == want.tobytes()
        else:
            ok &= img is None
        # pipelined form: 4 different cameras; submit(k) hands back frame k-1, flush() the last one
        cams = [orc.Camera(0.5 + 0.4 * k, 0.7 + 0.1 * k, 1.8 + 0.2 * k) for k in range(4)]
        got = []
        for c in cams:
            out = r.submit(hip.make_frame(c.get_view(), c.get_pos(), W / H, 45.0, W, H))
            got.append(None if out is None else out.numpy().copy())
        last = r.flush()
        got.append(None if last is None else last.numpy().copy())
        ok &= got[0] is None and r.flush() is None
        for k, c in enumerate(cams):
            if rank == 0:
                want, _ = orc.render(nodes, g.min, g.voxel_size, c.get_view(), c.get_pos(), W / H, 45.0, W, H)
                ok &= got[k + 1] is not None and got[k + 1].tobytes() == want.tobytes()
            else:
                ok &= got[k + 1] is None
        # two pipelines taking the frames in turn (what bench.py does for N > 1): their collectives interleave in the
        # same order on every rank and never touch each other's buffers
        pair = [tilesplit.TileSplitRenderer(OracleBackend(nodes, g.min, g.voxel_size), rank, world, band_rows=band, payload=payload)
                for _ in range(2)]
        cams2 = [orc.Camera(0.3 + 0.5 * k, 0.6, 1.7 + 0.1 * k) for k in range(5)]
        got2 = {}
        for k, c in enumerate(cams2):
            out = pair[k % 2].submit(hip.make_frame(c.get_view(), c.get_pos(), W / H, 45.0, W, H))
            if k >= 2:
                got2[k - 2] = None if out is None else out.numpy().copy()
        for i in range(2):
            k_last = max(k for k in range(len(cams2)) if k % 2 == i)
            out = pair[i].flush()
            got2[k_last] = None if out is None else out.numpy().copy()
        for k, c in enumerate(cams2):
            if rank == 0:
                want, _ = orc.render(nodes, g.min, g.voxel_size, c.get_view(), c.get_pos(), W / H, 45.0, W, H)
                ok &= got2[k] is not None and got2[k].tobytes() == want.tobytes()
            else:
                ok &= got2[k] is None
        # several frames per collective: batches of 3 cameras, pipelined (submit_batch hands back the previous batch)
        rb = tilesplit.TileSplitRenderer(OracleBackend(nodes, g.min, g.voxel_size), rank, world, band_rows=band, payload=payload)
        cams3 = [orc.Camera(0.2 + 0.3 * k, 0.65 + 0.02 * k, 1.75) for k in range(9)]
        frames3 = [hip.make_frame(c.get_view(), c.get_pos(), W / H, 45.0, W, H) for c in cams3]
        outs = []
        for bi in range(3):
            o = rb.submit_batch(frames3[3 * bi: 3 * bi + 3])
            outs.append(None if o is None else [t.numpy().copy() for t in o])
        o = rb.flush_batch()
        outs.append(None if o is None else [t.numpy().copy() for t in o])
        ok &= outs[0] is None and rb.flush_batch() is None
        for bi in range(3):
            for f in range(3):
                c = cams3[3 * bi + f]
                if rank == 0:
                    want, _ = orc.render(nodes, g.min, g.voxel_size, c.get_view(), c.get_pos(), W / H, 45.0, W, H)
                    ok &= outs[bi + 1] is not None and outs[bi + 1][f].tobytes() == want.tobytes()
                else:
                    ok &= outs[bi + 1] is None
        one = rb.render_batch(frames3[:3])                       # unpipelined batch
        if rank == 0:
            want, _ = orc.render(nodes, g.min, g.voxel_size, cams3[2].get_view(), cams3[2].get_pos(), W / H, 45.0, W, H)
            ok &= one is not None and len(one) == 3 and one[2].numpy().tobytes() == want.tobytes()
        try:                    # mixing the two forms with a frame in flight is refused
            r.submit(frame)
            r.render(frame)
            ok = False
        except RuntimeError:
            r.flush()
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
