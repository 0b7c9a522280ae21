#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on this repo's HIP path.

Metric   : Mrays/s of primary rays (and ms/frame) at 1920x1080.
Workload : BASELINE config 2 -- 256^3 shell-sphere voxel grid (main.cpp:337-372 rules), octree to
           min-leaf 1 (374,921 nodes), Camera(theta 0.5, phi 0.7, r 1.8), fov 45, aspect W/H.
Step     : one frame = one pass of the hot path over 2,073,600 rays, octree and framebuffer resident in HBM.
N = 1    : one kernel launch per frame into a device framebuffer, one frame strictly after the other; the timed frames
           are replayed from a HIP graph of 50 consecutive frames (--graph-frames; 0 = plain stream launches with a
           HIP event pair around every kernel): the runtime needs ~9 us between dependent plain launches, ~1 us
           between graph nodes.
N > 1    : launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`; every rank
           holds the octree, renders its round-robin bands and ONE torch.distributed.gather (RCCL over
           xGMI) per frame lands the image on rank 0, which re-interleaves it (strong scaling).  The payload
           is 4 bytes per pixel (the Lambert term; rank 0 finishes the colour, bit-identical) and the gather
           of batch k overlaps the renders of batch k+1 (8 consecutive frames travel in one gather: fewer, larger
           collectives); three such pipelines on three HIP streams take the batches in turn;
           all K frames are complete inside the timed region.

Prints ONE JSON line on rank 0.  `cpu_baseline` is this repo's own C restatement of the reference's GLSL
kernel (the reference has no CPU path and publishes no numbers), timed here on the box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
NODE_BYTES = 60                # struct GPUNodes, the reference's node record (constant even if repacked)
PIXEL_BYTES = 16               # RGBA32F
VALU_MIX_CYCLES = 2.85          # measured issue cost of the traversal loop's instruction mix
SIMDS, CLOCK_GHZ, VALU_CYCLES_PER_WAVE_INST = 1024, 2.4, 2   # 256 CUs x 4 SIMD-32: a wave64 VALU instruction issues over 2 cycles (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000, help="timed frames (2000 frames = 0.15 s at N=1: past the clock ramp of a cold GPU)")
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--dim", type=int, default=256, help="test-sphere grid edge (config 2: 256)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--band-rows", type=int, default=16)
    ap.add_argument("--kernel", choices=["auto", "packed", "persistent", "generic"], default="auto")
    ap.add_argument("--cpu-frames", type=int, default=40,
                    help="frames of the CPU baseline sample (0 = skip); 40 frames on 16 cores = about 12 core-seconds")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo (+ --share-gpu) rehearses the multi-rank path on one GPU; the gather is staged through host memory")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use GPU 0 (rehearsal only)")
    ap.add_argument("--verify", action="store_true", help="(kept for old command lines: the comparison below is always made)")
    ap.add_argument("--no-verify", action="store_true",
                    help="N>1: skip rank 0's comparison of the assembled frame with the CPU oracle (one 20 ms oracle frame, outside the timed region)")
    ap.add_argument("--order", choices=["temporal", "centre-out"], default="temporal",
                    help="tile launch order of the packed kernel (scheduling only; pixels are identical)")
    ap.add_argument("--order-period", type=int, default=8, help="temporal order: rebuild the table every n-th frame")
    ap.add_argument("--payload", choices=["shade", "rgba"], default="shade",
                    help="N>1: what a part ships to rank 0 -- 4-byte Lambert term per pixel (default) or the 16-byte pixel")
    ap.add_argument("--no-pipeline", action="store_true", help="N>1: finish the gather of a frame before rendering the next")
    ap.add_argument("--pipelines", type=int, default=3,
                    help="N>1: independent submit/flush pipelines, each on its own HIP stream, that take the frames in turn "
                         "(a part's kernel is bounded by its deepest rays, not by its pixel count, so one pipeline leaves "
                         "most of each GPU idle); 1 = a single pipeline")
    ap.add_argument("--frames-per-gather", type=int, default=8,
                    help="N>1: consecutive frames whose parts travel in ONE gather (fewer, larger collectives: a "
                         "torch.distributed call costs the host tens of microseconds, as much as a whole frame); 1 = one gather per frame")
    ap.add_argument("--graph-frames", type=int, default=50,
                    help="N=1: capture this many consecutive frames (one kernel each, strictly one after the other) in a HIP "
                         "graph and replay it: the ~9 us the runtime needs between dependent plain launches shrink to ~1 us; "
                         "0 = plain stream launches with a HIP event pair around every kernel")
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="N=1 only: render consecutive frames on this many HIP streams (own framebuffers) so that the "
                         "deep-ray tail of one frame overlaps the start of the next; 1 = strictly one frame at a time")
    return ap.parse_args()


def host_cores() -> int:
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(args, grid, nodes, view, pos, gpu_pops):
    """Oracle (own restatement of the reference GLSL) on the host cores: all cores + one thread."""
    import numpy as np

    from oracle import orc   # cpu_baseline leg only

    W, H = args.width, args.height
    cores = host_cores()
    out = np.zeros((H, W, 4), np.float32)
    gmin, voxel = grid.min, grid.voxelSize
    _, st = orc.render(nodes, gmin, voxel, view, pos, W / H, 45.0, W, H, nthreads=cores, out=out)   # warm-up + stats
    assert st["pops"] == gpu_pops, "GPU and oracle disagree on the pop count"
    ts = []
    for _ in range(args.cpu_frames):
        t = time.perf_counter()
        orc.render(nodes, gmin, voxel, view, pos, W / H, 45.0, W, H, nthreads=cores, out=out)
        ts.append(time.perf_counter() - t)
    ts.sort()
    med = ts[len(ts) // 2]
    t1s = []
    for _ in range(3):
        t = time.perf_counter()
        orc.render(nodes, gmin, voxel, view, pos, W / H, 45.0, W, H, nthreads=1, out=out)
        t1s.append(time.perf_counter() - t)
    t1 = sorted(t1s)[1]
    return {
        "value": round(W * H / med / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": f"{args.cpu_frames} full {W}x{H} frames of the same scene/camera, median, OpenMP dynamic over rows; "
                  f"plus 3 frames on 1 thread (median)",
        "single_thread_value": round(W * H / t1 / 1e6, 3),
        "note": "own restatement of reference GLSL; reference has no CPU path and publishes no numbers",
    }, out


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}`")
        args.gpus = world

    import numpy as np
    import torch

    import ray_tracing_octrees_amd as rto
    from ray_tracing_octrees_amd import tilesplit

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime

        # a rank that dies must not leave the others waiting for ten minutes in a collective
        tmo = datetime.timedelta(seconds=180)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
        else:
            dist.init_process_group("gloo", timeout=tmo)

    # ---- scene: the product's own host layer (C++), not the oracle --------------------------------
    W, H = args.width, args.height
    grid = rto.VoxelGrid.test_sphere(args.dim)
    root = rto.createOctreeFromVoxelGrid(grid)
    nodes = root.flatten()
    rto.freeOctree(root)
    cam = rto.Camera(0.5, 0.7, 1.8)
    view, pos = cam.getView(), cam.getPos()
    frame = rto.make_frame(view, pos, W / H, 45.0, W, H)

    ctx = rto.Context(local_rank)
    ctx.upload_octree(nodes, grid.min, grid.voxelSize)
    ctx.set_kernel({"auto": rto.KERNEL_AUTO, "packed": rto.KERNEL_PACKED, "persistent": rto.KERNEL_PACKED_PERSISTENT,
                    "generic": rto.KERNEL_GENERIC}[args.kernel])
    ctx.set_launch_order(1 if args.order == "temporal" else 0, args.order_period)
    info = ctx.info()
    pipelined = world > 1 and not args.no_pipeline
    npipe = max(1, args.pipelines) if pipelined else 1
    renderers = [tilesplit.TileSplitRenderer(tilesplit.HipBackend(ctx), rank, world, band_rows=args.band_rows,
                                             stage_through_host=(args.dist_backend == "gloo"), payload=args.payload)
                 for _ in range(npipe)]
    renderer = renderers[0]
    rest_renderer = tilesplit.TileSplitRenderer(tilesplit.HipBackend(ctx), rank, world, band_rows=args.band_rows,
                                                stage_through_host=(args.dist_backend == "gloo"), payload=args.payload)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x: float) -> float:
        dev = "cuda" if args.dist_backend == "nccl" else "cpu"
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # side streams: kernels, events and (for N > 1) the RCCL gathers order themselves on them; pipeline i owns stream i
    pstreams = [torch.cuda.Stream() for _ in range(npipe)]
    stream = pstreams[0]
    torch.cuda.set_stream(stream)

    fpg = max(1, args.frames_per_gather) if pipelined else 1

    def run_frames(n):
        """n frames.  With pipelines, batch j (fpg consecutive frames, one gather) goes to pipeline j % npipe -- the same
        order on every rank, so the collectives match; every frame is assembled on rank 0 before this returns."""
        img_ = None
        if not pipelined:
            for _ in range(n):
                img_ = renderer.render(frame)
            return img_
        full, rest = divmod(n, fpg)
        for j in range(full):
            with torch.cuda.stream(pstreams[j % npipe]):
                renderers[j % npipe].submit_batch([frame] * fpg)   # render, complete this pipeline's previous batch, start the gather
        for i in range(npipe):
            with torch.cuda.stream(pstreams[i]):
                out_ = renderers[i].flush_batch()
                img_ = out_[-1] if out_ is not None else img_
        if rest:                                                   # exactly n frames: the remainder as one smaller batch
            with torch.cuda.stream(pstreams[0]):                   # (own renderer: the pipelines keep their full-size buffers)
                out_ = rest_renderer.render_batch([frame] * rest)
                img_ = out_[-1] if out_ is not None else img_
        return img_

    run_frames(args.warmup)
    sync_all()

    # ---- timed region: exactly K frames ------------------------------------------------------------
    fif = max(1, args.frames_in_flight) if world == 1 else 1
    if fif > 1:
        streams = [torch.cuda.Stream() for _ in range(fif)]
        bufs = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(fif)]
        for i in range(fif):
            ctx.render_device(frame, bufs[i].data_ptr(), None, streams[i].cuda_stream)
        sync_all()
    use_graph = world == 1 and fif == 1 and args.graph_frames > 0
    graph = None
    gframes = 0
    if use_graph:
        # after the warm-up frames rto_render_device allocates nothing and never synchronises: it can be stream-captured
        gframes = min(args.graph_frames, args.steps)
        buf0 = renderer.render(frame)
        ctx.timing_begin(0)
        sync_all()
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                for _ in range(gframes):
                    ctx.render_device(frame, buf0.data_ptr(), None, stream.cuda_stream)
            graph.replay()                              # untimed: first replay of a fresh graph
            sync_all()
            ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        except Exception as e:                          # a runtime that cannot capture: plain launches, said so in the line
            print(f"bench: HIP graph capture failed ({type(e).__name__}: {e}); falling back to plain launches", file=sys.stderr, flush=True)
            use_graph, graph, gframes = False, None, 0
            torch.cuda.synchronize()
            stream = torch.cuda.Stream()
            torch.cuda.set_stream(stream)
            pstreams[0] = stream
            for _ in range(3):
                renderer.render(frame)
            sync_all()
    if not use_graph:
        ctx.timing_begin(args.steps)  # HIP event pair around every traversal kernel, on its launch stream, no syncs
    sync_all()
    t0 = time.perf_counter()
    if use_graph:
        ev_a.record(stream)
        for _ in range(args.steps // gframes):
            graph.replay()
        for _ in range(args.steps % gframes):           # exactly K frames: the remainder as plain launches
            ctx.render_device(frame, buf0.data_ptr(), None, stream.cuda_stream)
        ev_b.record(stream)
        img = buf0
    elif fif > 1:
        for k in range(args.steps):
            s_ = streams[k % fif]
            ctx.render_device(frame, bufs[k % fif].data_ptr(), None, s_.cuda_stream)
            img = bufs[k % fif]
    else:
        img = run_frames(args.steps)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed = max_over_ranks(elapsed)

    # ---- everything below is outside the timed region ---------------------------------------------
    rays = W * H
    result = None
    if rank == 0:
        stats = ctx.frame_stats(frame)                       # exact pop count of this frame (instrumented kernel)
        pops_per_ray = stats["pops"] / stats["rays"]
        bytes_per_ray = pops_per_ray * NODE_BYTES + PIXEL_BYTES
        roofline = None
        if world == 1:
            if use_graph:
                # one HIP event pair around the whole timed region, on the launch stream: GPU time per frame = the
                # traversal kernel + the ~1 us between graph nodes (an upper bound of the kernel's own duration)
                k_avg = ev_a.elapsed_time(ev_b) / args.steps
                kms = [k_avg]
            else:
                kms = sorted(float(x) for x in ctx.timing_read())
                assert len(kms) == args.steps
                k_avg = sum(kms) / len(kms)
            # cost of an event pair with nothing between (reported, not subtracted): rocprofv3's kernel
            # duration is ~ k_avg minus this
            cal = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
            for a, b in cal:
                a.record(stream); b.record(stream)
            torch.cuda.synchronize()
            pair_overhead = sorted(a.elapsed_time(b) for a, b in cal)[len(cal) // 2]
            achieved = rays * bytes_per_ray / (k_avg * 1e-3) / 1e9
            traffic = None
            valu = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("dim") == args.dim and tj.get("width") == W and tj.get("height") == H:
                    traffic = tj.get("hbm_bytes_per_launch")
                    if tj.get("SQ_INSTS_VALU"):
                        # the bound that actually holds (informational): wave-level VALU instructions (PMC) against the
                        # chip's issue rate -- 1024 SIMDs x 2.4 GHz, 2 cycles per wave64 instruction (v_pk_* count once here but do two lanes' worth)
                        insts = float(tj["SQ_INSTS_VALU"])
                        floor_ms = insts * VALU_CYCLES_PER_WAVE_INST / (SIMDS * CLOCK_GHZ * 1e9) * 1e3
                        mix_ms = insts * VALU_MIX_CYCLES / (SIMDS * CLOCK_GHZ * 1e9) * 1e3
                        valu = {"wave_insts_per_launch": int(insts), "lane_utilisation": round(tj["SQ_THREAD_CYCLES_VALU"] / (64.0 * insts), 3),
                                "issue_floor_ms": round(floor_ms, 5), "frac_of_issue_peak": round(floor_ms / k_avg, 3),
                                "mix_weighted_floor_ms": round(mix_ms, 5), "frac_of_mix_weighted_peak": round(mix_ms / k_avg, 3),
                                "model": f"{SIMDS} SIMDs x {CLOCK_GHZ} GHz; issue floor at {VALU_CYCLES_PER_WAVE_INST} cycles per wave64 VALU instruction; "
                                         f"mix-weighted at {VALU_MIX_CYCLES} (loop body: 27 v_pk_*, 16 v_max3/min3, 4 v_bcnt at ~4.3-4.5 cycles, "
                                         f"~158 single-rate ops at ~2.4; tools/ubench/valu_rate.hip)"}
            roofline = {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": {"auto": "k_trace_packed3", "packed": "k_trace_packed3", "persistent": "k_trace_packed3_persistent"}.get(args.kernel, "k_trace_generic") if (info.canonical and args.kernel != "generic") else "k_trace_generic",
                "launch_order": ("centre-out" if args.order != "temporal" else
                                 "temporal (tiles sorted by an earlier frame's trip counts; table built during the warm-up, frozen while the frames are replayed from the graph)" if use_graph else
                                 f"temporal (tiles sorted by an earlier frame's trip counts; k_sort_scatter after every {args.order_period}-th frame)"),
                "kernel_ms_avg": round(k_avg, 5), "kernel_ms_median": round(kms[len(kms) // 2], 5),
                "kernel_ms_how": (f"one HIP event pair around the {args.steps} timed frames / {args.steps} (graph replay: events inside a captured graph "
                                  f"cannot be timed): an upper bound of the kernel's duration, it includes the ~1 us between graph nodes") if use_graph
                                 else "HIP event pair around every traversal kernel launch of the timed region",
                "event_pair_overhead_ms": round(pair_overhead, 5),
                "algorithmic_bytes_per_ray": round(bytes_per_ray, 2), "pops_per_ray": round(pops_per_ray, 4),
                "algorithmic_bytes_per_launch": int(round(rays * bytes_per_ray)),
                "note": "algorithmic bytes = pops x 60 B reference node + 16 B pixel (SURVEY 8d); the packed kernel "
                        "reads 8-byte descriptors of internal nodes only, so frac may exceed 1: the real bound is VALU",
            }
            if valu is not None:
                roofline["valu"] = valu
        pcie = None
        if world == 1:
            # the C ABI's host-buffer entry point (kernel + 33 MB D2H over PCIe): informational, never `value`
            ts = []
            for _ in range(5):
                t = time.perf_counter()
                ctx.render_host(frame)
                ts.append(time.perf_counter() - t)
            ts.sort()
            pcie = {"ms_per_frame": round(ts[2] * 1e3, 4), "Mrays_per_s": round(rays / ts[2] / 1e6, 1),
                    "what": "rto_render_host: kernel + device-to-host copy of the RGBA32F frame into pageable memory"}
        cpu = None
        if world == 1 and args.cpu_frames > 0:
            cpu, want = cpu_baseline(args, grid, nodes, view, pos, stats["pops"])
            got = img.cpu().numpy()
            if got.tobytes() != want.tobytes():
                sys.exit("bench: the timed frame differs from the oracle's -- result void")
        elif not args.no_verify:
            from oracle import orc   # the checker, outside the timed region

            want = np.zeros((H, W, 4), np.float32)
            orc.render(nodes, grid.min, grid.voxelSize, view, pos, W / H, 45.0, W, H, nthreads=host_cores(), out=want)
            if img.cpu().numpy().tobytes() != want.tobytes():
                sys.exit("bench: the assembled frame differs from the oracle's -- result void")
        result = {
            "metric": "Mrays/s (primary rays), 1920x1080" if (W, H) == (1920, 1080) else f"Mrays/s (primary rays), {W}x{H}",
            "value": round(rays * args.steps / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.dim}^3 test-sphere voxel grid, octree to min-leaf 1 ({info.num_nodes} nodes), "
                            f"{W}x{H} primary rays, Camera(0.5,0.7,1.8), fov 45",
                "parallelism": (("1 GPU" + (f", frames replayed from a HIP graph of {gframes} consecutive frames" if use_graph else "")) if fif == 1 else f"1 GPU, {fif} frames in flight on {fif} HIP streams") if world == 1 else f"screen split over {world} GPUs, {args.band_rows}-row bands "
                                                          f"round-robin, 1 RCCL gather per {'frame' if fpg == 1 else f'{fpg} frames'} ({'4-byte Lambert term' if args.payload == 'shade' else 'RGBA32F'} per pixel"
                                                          f"{', gather k overlaps render k+1' if pipelined else ''}"
                                                          f"{f', {fpg} consecutive frames per gather' if fpg > 1 else ''}"
                                                          f"{f', {npipe} such pipelines on {npipe} HIP streams take the batches in turn' if npipe > 1 else ''})",
                "kernel": args.kernel,
            },
            "hit_rays": stats["hits"], "capped_rays": stats["capped"],
            "verified_against_oracle": bool((world == 1 and args.cpu_frames > 0) or (not (world == 1 and args.cpu_frames > 0) and not args.no_verify)),
            "device": ctx.device_name,
        }
        if roofline is not None:
            result["roofline"] = roofline
        if pcie is not None:
            result["pcie_inclusive"] = pcie
        if cpu is not None:
            result["cpu_baseline"] = cpu
            result["speedup_vs_cpu_all_cores"] = round(result["value"] / cpu["value"], 1)
        print(json.dumps(result), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:           # one JSON line or a loud failure, never a hang: take the whole job down
        import traceback

        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
