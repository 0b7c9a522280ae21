"""Developer check on a GPU box: parity of both kernels vs the oracle + kernel timings.
Usage: python tools/gpu_check.py [dim W H]...   (default: 64 512 512 and 256 1920 1080)"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ray_tracing_octrees_amd as rto  # noqa: E402
from oracle import orc  # noqa: E402


def check(dim, W, H, ctx, reps=20):
    g = orc.test_sphere_grid(dim)
    nodes = orc.build_flat_octree(g)
    cam = orc.Camera(0.5, 0.7, 1.8)
    view, pos = cam.get_view(), cam.get_pos()
    t = time.time()
    want = np.zeros((H, W, 4), np.float32)
    want, st = orc.render(nodes, g.min, g.voxel_size, view, pos, W / H, 45.0, W, H, nthreads=orc.max_threads(), out=want)
    t_cpu = time.time() - t
    wsteps = orc.render_steps(nodes, g.min, g.voxel_size, view, pos, W / H, 45.0, W, H)
    ctx.upload_octree(nodes, g.min, g.voxel_size)
    info = ctx.info()
    print(f"--- sphere {dim}^3 {W}x{H}: nodes {info.num_nodes} internal {info.num_internal} canonical {info.canonical} "
          f"depth {info.depth}; oracle {st} ({t_cpu * 1e3:.0f} ms, {orc.max_threads()} threads)")
    f = rto.make_frame(view, pos, W / H, 45.0, W, H)
    for name, k in (("packed", rto.KERNEL_PACKED), ("packed_v1", rto.KERNEL_PACKED_V1), ("generic", rto.KERNEL_GENERIC)):
        ctx.set_kernel(k)
        got = ctx.render_host(f)
        bad = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=-1).sum())
        gsteps = ctx.render_steps(f)
        sbad = int((gsteps != wsteps).sum())
        gs = ctx.frame_stats(f)
        ms = []
        for _ in range(reps):
            ctx.render_host(f)
            ms.append(ctx.last_kernel_ms())
        ms = np.array(ms)
        print(f"  {name:8s} pixel mismatches {bad}/{W * H} maxabs {np.abs(got - want).max():.3g} | steps mismatches {sbad} | "
              f"stats {gs} vs oracle pops {st['pops']} hits {st['hits']} capped {st['capped']} | "
              f"kernel ms min {ms.min():.4f} med {np.median(ms):.4f} -> {W * H / np.median(ms) / 1e3:.1f} Mrays/s")
        if bad:
            ys, xs = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=-1))
            for y, x in list(zip(ys, xs))[:5]:
                print("    diff at", (x, y), got[y, x], want[y, x], "steps", gsteps[y, x], wsteps[y, x])


def main():
    ctx = rto.Context(0)
    print("device:", ctx.device_name)
    args = [int(a) for a in sys.argv[1:]]
    cases = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(64, 512, 512), (256, 1920, 1080)]
    for dim, W, H in cases:
        check(dim, W, H, ctx)


if __name__ == "__main__":
    main()
