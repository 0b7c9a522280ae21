#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on this repo's HIP path.

Metric   : Mrays/s of primary rays (and ms/frame).
Workload : --config 2 (default) BASELINE config 2 -- 256^3 shell-sphere voxel grid (main.cpp:337-372 rules), octree to
                      min-leaf 1 (374,921 nodes), Camera(theta 0.5, phi 0.7, r 1.8), fov 45, 1920x1080.
           --config 4           the shipped sceneCache.bin grid (425x243x29, root 512; tests/golden/ref_scene_cache.npz),
                      oblique Camera(0.6, 0.5, 3500): the irregular / divergent case, 1920x1080.
           --config 4synthetic  that grid resampled nearest-neighbour to the 512x512x128 BASELINE.json names (SYNTHETIC: the
                      real one cannot be regenerated), Camera(0.6, 0.5, 4500) on the grid centre, 1920x1080.
           --config 5           512^3 sphere, 3840x2160, Marching-Cubes leaf triangles + 1 shadow ray per hit.
Step     : one frame = one pass of the hot path over W x H primary rays, octree and framebuffer resident in HBM.
N = 1    : one kernel launch per frame into a device framebuffer, one frame strictly after the other; the timed frames
           are replayed from a HIP graph of --graph-frames consecutive frames (0 = plain stream launches with a HIP
           event pair around every kernel).
N > 1    : `python3 bench.py --gpus N` starts N fresh ranks itself (python -m torch.distributed.run, one per GPU; the
           parent process never touches a GPU) and relays rank 0's line; launched under torch.distributed.run it is a
           rank.  Every rank holds the octree and renders its round-robin bands; ONE grouped RCCL send/recv per batch of
           frames lands the parts on rank 0, which re-interleaves them -- all of it below the C boundary (rto_comm_* in
           include/rto_hip.h; torch.distributed only carries the 128-byte communicator id, the barriers and the max over
           ranks).  Strong scaling: one frame split N ways.  --dist-backend gloo (+ --share-gpu) is the rehearsal path
           through ray_tracing_octrees_amd.tilesplit (gather staged through host memory).

Prints ONE JSON line on rank 0.  `roofline` prices the traversal kernel against the bound that holds -- VALU issue
(wave-level VALU instructions from the committed PMC pass x 2 cycles on 1,024 SIMD-32 at 2.4 GHz) -- and keeps SURVEY
8d's HBM-algorithmic figure beside it.  `cpu_baseline` is this repo's own C restatement of the reference's GLSL
kernel (the reference has no CPU path and publishes no numbers), timed here on the box's host cores.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
NODE_BYTES = 60                # struct GPUNodes, the reference's node record (constant even if repacked)
PIXEL_BYTES = 16               # RGBA32F
SIMDS, CLOCK_GHZ, VALU_CYCLES_PER_WAVE_INST = 1024, 2.4, 2   # 256 CUs x 4 SIMD-32: a wave64 VALU instruction issues over 2 cycles (MI355X_MICROARCH.md)
VALU_PEAK_GINST = SIMDS * CLOCK_GHZ / VALU_CYCLES_PER_WAVE_INST   # 1228.8 G wave-instructions/s
LEAN_LOOP_CYCLES_PER_INST = 401.5 / 120.0   # k_trace_lean's loop trip (rays that start outside the root box): 120 VALU instructions,
                                            # 401.5 issue cycles at the measured per-opcode costs (profiles/r02_valu_issue_rates.txt)
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_counters.json")

CONFIGS = {
    "2": dict(width=1920, height=1080, mode="octree"),
    "4": dict(width=1920, height=1080, mode="octree"),
    "4synthetic": dict(width=1920, height=1080, mode="octree"),
    "5": dict(width=3840, height=2160, mode="triangles"),
}
KERNEL_NAMES = {"auto": "k_trace_lean", "packed": "k_trace_lean", "persistent": "k_trace_lean_persistent", "packed_v3": "k_trace_packed3",
                "packed_v1": "k_trace_packed", "generic": "k_trace_generic"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000, help="timed frames (2000 frames = 0.1 s at N=1, config 2)")
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="2", help="BASELINE.json configuration (see the module docstring)")
    ap.add_argument("--dim", type=int, default=None, help="config 2 / 5: test-sphere grid edge (default 256 / 512)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--band-rows", type=int, default=16)
    ap.add_argument("--kernel", choices=sorted(KERNEL_NAMES), default="auto")
    ap.add_argument("--cpu-frames", type=int, default=None,
                    help="frames of the CPU baseline sample (0 = skip); default: about 10-30 core-seconds for the configuration")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo (+ --share-gpu) rehearses the multi-rank path on one GPU; the gather is staged through host memory")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use GPU 0 (rehearsal only)")
    ap.add_argument("--verify", action="store_true", help="(kept for old command lines: the comparison below is always made)")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip rank 0's comparison of the timed frame with the CPU oracle (one oracle frame, outside the timed region)")
    ap.add_argument("--order", choices=["temporal", "centre-out"], default="temporal",
                    help="tile launch order of the packed kernel (scheduling only; pixels are identical)")
    ap.add_argument("--order-period", type=int, default=8, help="temporal order: rebuild the table every n-th frame")
    ap.add_argument("--payload", choices=["shade", "rgba"], default="shade",
                    help="N>1: what a part ships to rank 0 -- 4-byte Lambert term per pixel (default) or the 16-byte pixel")
    ap.add_argument("--no-pipeline", action="store_true", help="N>1: finish the gather of a frame before rendering the next")
    ap.add_argument("--pipelines", type=int, default=3,
                    help="N>1: independent submit/flush pipelines, each on its own HIP stream, that take the batches in turn; 1 = a single pipeline")
    ap.add_argument("--frames-per-gather", type=int, default=8,
                    help="N>1: consecutive frames whose parts travel in ONE gather (fewer, larger collectives); 1 = one gather per frame")
    ap.add_argument("--graph-frames", type=int, default=None,
                    help="N=1: capture this many consecutive frames in a HIP graph and replay it; 0 = plain stream launches with a "
                         "HIP event pair around every kernel.  Default: 48 when --steps >= 200, else 0 (with several frames per "
                         "launch a short run gains nothing from a graph and pays its first launch)")
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="N=1 only: render consecutive frames on this many HIP streams (own framebuffers); 1 = strictly one frame at a time")
    ap.add_argument("--ramp-ms", type=float, default=300.0,
                    help="N=1: untimed frames of the same workload rendered for this long before the W warm-up frames, so that a cold "
                         "GPU has reached its working clock (disclosed in the line as `clock_ramp`); 0 = none")
    ap.add_argument("--orbit-frames", type=int, default=None,
                    help="N=1, octree configs: extra figure with the camera orbiting 0.01 rad per frame, plain launches, launch-order "
                         "table rebuilt every --order-period-th frame (default 240; 0 = skip)")
    ap.add_argument("--frames-per-launch", type=int, default=None,
                    help="N=1, octree configs: frames rendered by ONE kernel launch (rto_render_batch_device, at most 8; default 4). A single "
                         "frame's kernel lasts as long as its deepest tile's chain of node visits with most of the GPU idle; frames "
                         "launched together fill it.  1 = one launch per frame (also reported in the line as `one_frame_per_launch`)")
    ap.add_argument("--rehearse-world", type=int, default=0,
                    help="with --gpus 1 --force-comm: render / ship / assemble as rank 0 of that many GPUs (rto_comm_debug_rehearse); "
                         "the line then reports the per-rank cost of the split, NOT a frame rate (only 1/N of every frame is rendered)")
    ap.add_argument("--rehearse-rank", type=int, default=0, help="with --rehearse-world: the rank to play (from 4 GPUs on rank 0 only gathers "
                                                                  "and assembles, ranks 1..N-1 render)")
    ap.add_argument("--force-comm", action="store_true",
                    help="N=1: drive the frames through rto_comm_* with a one-rank RCCL communicator (rehearses the N>1 code path on one GPU)")
    ap.add_argument("--launcher-dry-run", action="store_true", help="--gpus N without WORLD_SIZE: print the child command line and exit")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ N > 1: self-launch
def child_command(args, argv, port: int) -> list[str]:
    """The N fresh ranks `python3 bench.py --gpus N ...` starts: one process per GPU over RCCL, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv) -> int:
    """Parent of an N>1 run: touches no GPU, starts the ranks as CHILD processes (never an exec of this process),
    relays rank 0's single JSON line and returns the children's status -- non-zero, loudly, on any failure."""
    cmd = child_command(args, argv, free_port())
    if args.launcher_dry_run:
        print(json.dumps({"launcher": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    if p.returncode != 0 or len(lines) != 1:
        sys.stderr.write(p.stderr[-8000:])
        sys.stderr.write(p.stdout[-4000:])
        sys.stderr.write(f"\nbench: the {args.gpus}-rank job failed (exit status {p.returncode}, {len(lines)} result lines)\n")
        return p.returncode if p.returncode != 0 else 1
    print(lines[0], flush=True)
    return 0


# ------------------------------------------------------------------------------------------------ helpers
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


def device_source_hash() -> str:
    """Identifies the device code a PMC entry was measured on (the counters are per binary)."""
    h = hashlib.sha256()
    for rel in ("ray_tracing_octrees_amd/csrc/rto_device.hip.h", "ray_tracing_octrees_amd/csrc/rto_api.hip"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_scene(args):
    """The product's own host layer (C++) builds the grid; returns everything both the GPU path and the checker need."""
    import numpy as np

    import ray_tracing_octrees_amd as rto

    cfg = args.config
    if cfg in ("2", "5"):
        dim = args.dim or (256 if cfg == "2" else 512)
        grid = rto.VoxelGrid.test_sphere(dim)                       # main.cpp:337-372, 1052-1070 + recenterFilledVoxels
        cam = rto.Camera(0.5, 0.7, 1.8)
        what = f"{dim}^3 test-sphere voxel grid"
        camtxt = "Camera(0.5,0.7,1.8)"
    else:
        z = np.load(os.path.join(ROOT, "tests", "golden", "ref_scene_cache.npz"))     # sceneCache.bin via the reference's loadVoxelGrid
        dims = tuple(int(x) for x in z["dims"])
        data = np.unpackbits(z["packed"])[: dims[0] * dims[1] * dims[2]].reshape(dims[2], dims[1], dims[0])
        gmin, voxel = z["min"].astype(np.float32), np.float32(z["voxel"])
        if cfg == "4synthetic":
            tz, ty, tx = 128, 512, 512
            iz = ((np.arange(tz) + 0.5) * data.shape[0] / tz).astype(np.int64)
            iy = ((np.arange(ty) + 0.5) * data.shape[1] / ty).astype(np.int64)
            ix = ((np.arange(tx) + 0.5) * data.shape[2] / tx).astype(np.int64)
            data = np.ascontiguousarray(data[iz][:, iy][:, :, ix])
            grid = rto.VoxelGrid.from_array(data, gmin, voxel)
            cam = rto.Camera(0.6, 0.5, 4500.0)
            centre = gmin + 0.5 * np.array([tx, ty, tz], np.float32) * voxel
            cam.setTarget(centre)
            what = "sceneCache.bin resampled nearest-neighbour to 512x512x128 (SYNTHETIC stand-in for the grid BASELINE.json names)"
            camtxt = "Camera(0.6,0.5,4500) on the grid centre"
        else:
            grid = rto.VoxelGrid.from_array(data, gmin, voxel)
            cam = rto.Camera(0.6, 0.5, 3500.0)
            what = "sceneCache.bin as shipped (425x243x29 voxels, octree root 512)"
            camtxt = "oblique Camera(0.6,0.5,3500)"
    return grid, cam, what, camtxt


def cpu_baseline(args, scene, frames_default):
    """Oracle (own restatement of the reference GLSL) on the host cores: all cores + one thread; returns the oracle's frame."""
    import numpy as np

    from oracle import orc   # cpu_baseline leg only

    W, H = scene["W"], scene["H"]
    cores = host_cores()
    nodes, gmin, voxel, view, pos = scene["nodes"], scene["gmin"], scene["voxel"], scene["view"], scene["pos"]
    n = frames_default if args.cpu_frames is None else args.cpu_frames
    if scene["mode"] == "triangles":
        tris, off = scene["tris"], scene["tri_offset"]

        def run(threads):
            return orc.render_triangles(nodes, tris, off, gmin, voxel, view, pos, W / H, 45.0, W, H, shadow=True, nthreads=threads)
    else:
        out = np.zeros((H, W, 4), np.float32)

        def run(threads):
            return orc.render(nodes, gmin, voxel, view, pos, W / H, 45.0, W, H, nthreads=threads, out=out)
    want, st = run(cores)                                             # warm-up + stats + the checker's frame
    want = np.array(want, copy=True)
    if n <= 0:
        return None, want, st
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        run(cores)
        ts.append(time.perf_counter() - t)
    ts.sort()
    med = ts[len(ts) // 2]
    t = time.perf_counter()
    run(1)
    t1 = time.perf_counter() - t
    return {
        "value": round(W * H / med / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": f"{n} full {W}x{H} frames of the same scene/camera, median, OpenMP dynamic over rows; plus 1 frame on 1 thread",
        "single_thread_value": round(W * H / t1 / 1e6, 3),
        "note": "own restatement of reference GLSL; reference has no CPU path and publishes no numbers",
    }, want, st


def pmc_entry(config: str, kernel_name: str, order: str):
    """Counters of the committed rocprofv3 --pmc passes for this (config, kernel, launch order), or None."""
    if not os.path.exists(PMC_FILE):
        return None
    with open(PMC_FILE) as f:
        table = json.load(f)
    for e in table.get("entries", []):
        if e.get("config") == config and e.get("kernel") == kernel_name and e.get("order") == order:
            return e
    return None


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.graph_frames is None:
        args.graph_frames = 48 if args.steps >= 200 else 0
    if args.rehearse_world > 1:
        args.no_verify, args.cpu_frames, args.force_comm = True, 0, True     # the assembled frames hold one rank's bands only
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(self_launch(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
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
        # control plane only (communicator id, barriers, max over ranks): the frames travel through rto_comm_* (RCCL)
        dist.init_process_group("gloo", timeout=tmo)

    # ---- scene ------------------------------------------------------------------------------------
    cfg = CONFIGS[args.config]
    W, H = args.width or cfg["width"], args.height or cfg["height"]
    triangles = cfg["mode"] == "triangles"
    grid, cam, what, camtxt = build_scene(args)
    view, pos = cam.getView(), cam.getPos()
    frame = rto.make_frame(view, pos, W / H, 45.0, W, H)

    ctx = rto.Context(local_rank)
    vox = grid.data
    ctx.build_octree(vox, grid.min, grid.voxelSize)                 # createOctreeFromVoxelGrid + setOctree on the GPU (row N4)
    if triangles:
        ctx.build_leaf_triangles(None)                              # MarchingCubesRenderer per leaf, on the GPU (row N2)
    ctx.set_kernel({"auto": rto.KERNEL_AUTO, "packed": rto.KERNEL_PACKED, "persistent": rto.KERNEL_PACKED_PERSISTENT,
                    "packed_v3": rto.KERNEL_PACKED_V3, "packed_v1": rto.KERNEL_PACKED_V1, "generic": rto.KERNEL_GENERIC}[args.kernel])
    ctx.set_launch_order(1 if args.order == "temporal" else 0, args.order_period)
    info = ctx.info()
    use_comm = (world > 1 and args.dist_backend == "nccl") or (world == 1 and args.force_comm)
    pipelined = (world > 1 or args.force_comm) and not args.no_pipeline
    npipe = (max(1, args.pipelines) if pipelined else 1) if not use_comm else 1
    comm = None
    if use_comm:
        from ray_tracing_octrees_amd import hip as _hip

        ids = [_hip.comm_unique_id() if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(ids, src=0)
        comm = _hip.Comm(ctx, world, rank, ids[0], band_rows=args.band_rows)
        if args.rehearse_world > 1:
            if world != 1:
                raise SystemExit("--rehearse-world needs --gpus 1 --force-comm")
            comm.debug_rehearse(args.rehearse_world, args.rehearse_rank)   # this GPU plays that rank: per-rank cost of the split, no peer traffic
        comm_mode = _hip.RESIDENT_TRIANGLES_SHADOW if triangles else _hip.RESIDENT_OCTREE

    def backend():
        return tilesplit.HipBackend(ctx, triangles=triangles, shadow=True)

    renderers = [tilesplit.TileSplitRenderer(backend(), rank, world, band_rows=args.band_rows,
                                             stage_through_host=(args.dist_backend == "gloo"), payload=args.payload)
                 for _ in range(npipe)]
    renderer = renderers[0]
    rest_renderer = tilesplit.TileSplitRenderer(backend(), rank, world, band_rows=args.band_rows,
                                                stage_through_host=(args.dist_backend == "gloo"), payload=args.payload)

    def render_to(buf_ptr, stream_handle, fr=None):
        if triangles:
            ctx.render_triangles_device(fr or frame, buf_ptr, True, None, stream_handle)
        else:
            ctx.render_device(fr or frame, buf_ptr, None, stream_handle)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x: float) -> float:
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # side streams: kernels, events and (for N > 1) the RCCL gathers order themselves on them; pipeline i owns stream i
    pstreams = [torch.cuda.Stream() for _ in range(npipe)]
    stream = pstreams[0]
    torch.cuda.set_stream(stream)

    fpg = max(1, args.frames_per_gather) if pipelined else 1
    comm_frames = torch.empty((fpg, H, W, 4), dtype=torch.float32, device="cuda") if (use_comm and rank == 0) else None
    comm_arr = rto.Context.frame_array([frame] * fpg) if use_comm else None

    def run_frames(n):
        """n frames.  With pipelines, batch j (fpg consecutive frames, one gather) goes to pipeline j % npipe -- the same
        order on every rank, so the collectives match; every frame is assembled on rank 0 before this returns."""
        img_ = None
        if use_comm:
            # rto_comm: batch k's grouped send/recv + assembly overlap batch k+1's render (two buffer sets, two HIP streams per
            # rank, no host wait until the flush); --no-pipeline flushes after every batch
            full, rest = divmod(n, fpg)
            ptr = comm_frames.data_ptr() if rank == 0 else 0
            stride = H * W * 16
            for _ in range(full):
                comm.submit(comm_arr, ptr, stride, comm_mode)
                if not pipelined:
                    comm.flush()
            if rest:
                comm.submit(rto.Context.frame_array([frame] * rest), ptr, stride, comm_mode)
            comm.flush()
            return comm_frames[(rest or fpg) - 1] if rank == 0 else None
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

    # ---- untimed: clock ramp (disclosed), then the W warm-up frames ------------------------------------
    ramp_frames = 0
    if world == 1 and args.ramp_ms > 0 and not use_comm:
        t_end = time.perf_counter() + args.ramp_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(20):
                renderer.render(frame)
            torch.cuda.synchronize()
            ramp_frames += 20
    primed = 0
    if use_comm:
        # untimed, disclosed: full batches through the pipe before the warm-up -- RCCL opens its connections at the first
        # send/recv, the buffers take the size of a full batch, every GPU reaches its working clock (all ranks: same count)
        t_end = time.perf_counter() + max(args.ramp_ms, 100.0) * 1e-3
        ptr0 = comm_frames.data_ptr() if rank == 0 else 0
        rounds = 0
        while True:
            for _ in range(4):
                comm.submit(comm_arr, ptr0, H * W * 16, comm_mode)
            comm.flush()
            rounds += 1
            go_on = torch.tensor([1.0 if time.perf_counter() < t_end else 0.0], dtype=torch.float64)
            if dist is not None:
                dist.all_reduce(go_on, op=dist.ReduceOp.MIN)          # every rank submits the same number of collectives
            if float(go_on.item()) == 0.0:
                break
        primed = rounds * 4 * fpg
    run_frames(args.warmup)
    sync_all()

    # ---- timed region: exactly K frames ------------------------------------------------------------
    fif = max(1, args.frames_in_flight) if world == 1 else 1
    if fif > 1:
        streams = [torch.cuda.Stream() for _ in range(fif)]
        bufs = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(fif)]
        for i in range(fif):
            render_to(bufs[i].data_ptr(), streams[i].cuda_stream)
        sync_all()
    use_graph = world == 1 and fif == 1 and args.graph_frames > 0 and not use_comm
    # frames per kernel launch (rto_render_batch_device / rto_render_triangles_batch_device): the single-GPU path
    fpl = 1
    if world == 1 and fif == 1 and not use_comm:
        fpl = max(1, min(8, args.steps, 4 if args.frames_per_launch is None else args.frames_per_launch))
    batch_buf = torch.empty((fpl, H, W, 4), dtype=torch.float32, device="cuda") if fpl > 1 else None
    batch_arrs = {}

    def render_chunk(nf, out=None):
        """nf <= fpl consecutive frames in one launch, into batch_buf[0..nf-1]"""
        if nf not in batch_arrs:
            batch_arrs[nf] = rto.Context.frame_array([frame] * nf)
        if triangles:
            ctx.render_triangles_batch_device(batch_arrs[nf], (out if out is not None else batch_buf).data_ptr(), H * W * 16, True, None, False, stream.cuda_stream)
        else:
            ctx.render_batch_device(batch_arrs[nf], (out if out is not None else batch_buf).data_ptr(), H * W * 16, None, False, stream.cuda_stream)

    def render_frames_plain(nframes):
        if fpl == 1:
            return run_frames(nframes)
        full, rest = divmod(nframes, fpl)
        for _ in range(full):
            render_chunk(fpl)
        if rest:
            render_chunk(rest)
        return batch_buf[(rest or fpl) - 1]

    if fpl > 1:
        for _ in range(3):
            render_chunk(fpl)
        sync_all()
    graph = None
    gframes = 0
    if use_graph:
        # after the warm-up frames the render entry points allocate nothing and never synchronise: they can be stream-captured
        gframes = min(args.graph_frames, args.steps) // fpl * fpl
        buf0 = renderer.render(frame)
        ctx.timing_begin(0)
        sync_all()
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                if fpl > 1:
                    for _ in range(gframes // fpl):
                        render_chunk(fpl)
                else:
                    for _ in range(gframes):
                        render_to(buf0.data_ptr(), stream.cuda_stream)
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
        ctx.timing_begin(-(-args.steps // fpl) if not (triangles or use_comm) else 0)  # HIP event pair around every traversal kernel, on its launch stream, no syncs
    sync_all()
    t0 = time.perf_counter()
    if use_graph:
        ev_a.record(stream)
        for _ in range(args.steps // gframes):
            graph.replay()
        if fpl > 1:                                     # exactly K frames: the remainder as plain launches
            img = render_frames_plain(args.steps % gframes) if args.steps % gframes else batch_buf[fpl - 1]
        else:
            for _ in range(args.steps % gframes):
                render_to(buf0.data_ptr(), stream.cuda_stream)
            img = buf0
        ev_b.record(stream)
    elif fif > 1:
        for k in range(args.steps):
            s_ = streams[k % fif]
            render_to(bufs[k % fif].data_ptr(), s_.cuda_stream)
            img = bufs[k % fif]
    else:
        if world == 1 and triangles and not use_comm:
            ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev_a.record(stream)
        img = render_frames_plain(args.steps)
        if world == 1 and triangles and not use_comm:
            ev_b.record(stream)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed = max_over_ranks(elapsed)

    # ---- everything below is outside the timed region ---------------------------------------------
    if img is not None:
        img = img.clone()                                # the legs below reuse the frame buffers
    if fpl > 1 and not all(bool(torch.equal(batch_buf[i], batch_buf[0])) for i in range(1, fpl)):
        sys.exit("bench: the frames of one launch differ from each other -- result void")
    rays = W * H
    result = None
    latency = None
    if use_comm:
        # single-frame latency of the split: one frame per collective, the host waits for the assembled frame before it
        # submits the next (what an interactive viewer sees); the timed region above is the batched, pipelined throughput
        one = rto.Context.frame_array([frame])
        ptr = comm_frames.data_ptr() if rank == 0 else 0
        for _ in range(5):
            comm.submit(one, ptr, H * W * 16, comm_mode); comm.flush()
        sync_all()
        n_lat = 40
        t_l = time.perf_counter()
        for _ in range(n_lat):
            comm.submit(one, ptr, H * W * 16, comm_mode); comm.flush()
        lat = (time.perf_counter() - t_l) / n_lat
        if dist is not None:
            lat = max_over_ranks(lat)
        sent, whole = comm.debug_last_payload()
        latency = {"payload_bytes_per_frame_and_rank": sent * 4, "payload_bytes_whole_rows": whole * 4,
                   "ms_per_frame": round(lat * 1e3, 5), "frames": n_lat,
                   "what": "rto_comm_submit of ONE frame + rto_comm_flush per frame: render part -> grouped send/recv -> assemble, host waits for each frame"}
    if rank == 0:
        kernel_name = KERNEL_NAMES[args.kernel] if info.canonical else "k_trace_generic"
        if fpl > 1 and kernel_name == "k_trace_lean":
            kernel_name = "k_trace_lean_batch"
        if triangles:
            kernel_name = ("k_trace_packed_triangles" if args.kernel == "packed_v3" else "k_trace_lean_triangles") if (info.canonical and args.kernel != "generic") else "k_trace_triangles"
            if fpl > 1 and kernel_name == "k_trace_lean_triangles":
                kernel_name = "k_trace_lean_triangles_batch"
            _, tstats = ctx.render_triangles_host(frame, shadow=True, stats=True)     # primary + shadow pops (instrumented kernel)
            stats = {"rays": rays, "pops": tstats["pops"], "hits": tstats["hits"], "capped": None}
        else:
            stats = ctx.frame_stats(frame)                       # exact pop count of this frame (instrumented kernel)
        pops_per_ray = stats["pops"] / rays
        bytes_per_ray = pops_per_ray * NODE_BYTES + PIXEL_BYTES
        tri_note = ""
        if triangles:
            # SURVEY 8d leaves config 5's triangle / shadow terms to the build: every pop of either traversal prices one
            # 60 B node, every triangle tested 48 B (v0, v1, v2, normal); the latter is not counted by the instrumented
            # kernel, so the figure below is a lower bound of the algorithmic bytes.
            tri_note = " (config 5: pops of the primary AND the shadow traversal; triangle bytes not included)"
        roofline = None
        if world == 1 and not use_comm:
            if use_graph or triangles:
                # one HIP event pair around the whole timed region, on the launch stream: GPU time per frame = the
                # traversal kernel + the gap between consecutive launches (an upper bound of the kernel's own duration)
                k_avg = ev_a.elapsed_time(ev_b) / args.steps
                kms = [k_avg]
            else:
                kms = [float(x) for x in ctx.timing_read()]
                assert len(kms) == -(-args.steps // fpl)
                k_avg = sum(kms) / args.steps                       # per frame
                kms = sorted(x / fpl for x in kms[: args.steps // fpl]) or sorted(kms)
            cal = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
            for a, b in cal:
                a.record(stream); b.record(stream)
            torch.cuda.synchronize()
            pair_overhead = sorted(a.elapsed_time(b) for a, b in cal)[len(cal) // 2]
            hbm_alg = rays * bytes_per_ray / (k_avg * 1e-3) / 1e9
            order_key = "centre-out" if args.order != "temporal" else "temporal"
            pmc = pmc_entry(args.config, kernel_name, order_key)
            if pmc is not None and int(pmc.get("frames_per_launch") or 1) != fpl:
                pmc = None                                           # counted for another batch size
            roofline = {
                "bound": "valu_issue", "achieved": None, "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave-instructions/s", "frac": None,
                "traffic": None, "traffic_source": None,
                "kernel": kernel_name,
                "model": f"{SIMDS} SIMD-32 x {CLOCK_GHZ} GHz / {VALU_CYCLES_PER_WAVE_INST} cycles per wave64 VALU instruction (MI355X_MICROARCH.md); "
                         "achieved = SQ_INSTS_VALU per launch (PMC pass, traffic_source) / the kernel's average duration measured in this run",
                "launch_order": ("centre-out" if order_key == "centre-out" else
                                 "temporal (tiles sorted by an earlier frame's trip counts; table built during the warm-up, frozen while the frames are replayed from the graph)" if use_graph else
                                 f"temporal (tiles sorted by an earlier frame's trip counts; k_sort_scatter after every {args.order_period}-th frame)"),
                "frames_per_launch": fpl,
                "kernel_ms_avg": round(k_avg * fpl, 5), "kernel_ms_median": round(kms[len(kms) // 2] * fpl, 5), "kernel_ms_per_frame": round(k_avg, 5),
                "kernel_ms_how": (f"one HIP event pair on the launch stream around the {args.steps} timed frames / {args.steps} (events inside a captured "
                                  f"graph cannot be timed): an upper bound of the kernel's duration, it includes the gap between consecutive launches") if (use_graph or triangles)
                                 else "HIP event pair around every traversal kernel launch of the timed region, on its launch stream",
                "event_pair_overhead_ms": round(pair_overhead, 5),
                "hbm_algorithmic": {
                    "achieved": round(hbm_alg, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "ratio": round(hbm_alg / HBM_PEAK_GBS, 4),
                    "bytes_per_ray": round(bytes_per_ray, 2), "pops_per_ray": round(pops_per_ray, 4),
                    "bytes_per_launch": int(round(rays * bytes_per_ray)) * fpl,
                    "note": "SURVEY 8d: pops x 60 B reference node + 16 B pixel" + tri_note + ". The packed kernels read 8-byte descriptors of "
                            "internal nodes only (L1/L2 resident), so this ratio exceeds 1: it prices the reference's layout, not this kernel's traffic",
                },
            }
            if pmc is not None:
                insts = float(pmc["SQ_INSTS_VALU"])                  # per launch = fpl frames
                achieved = insts / (k_avg * fpl * 1e-3) / 1e9
                roofline["achieved"] = round(achieved, 1)
                roofline["frac"] = round(achieved / VALU_PEAK_GINST, 4)
                roofline["valu_insts_per_launch"] = int(insts)
                if pmc.get("SQ_THREAD_CYCLES_VALU"):
                    roofline["lane_utilisation"] = round(float(pmc["SQ_THREAD_CYCLES_VALU"]) / (64.0 * insts), 3)
                if kernel_name in ("k_trace_lean", "k_trace_lean_batch", "k_trace_lean_persistent"):
                    # `frac` prices every instruction at the guide's 2 cycles; two thirds of this loop's instructions are
                    # half-rate on gfx950 (min/max, cvt, cmp, packed f32, 3-operand forms): at the measured issue costs
                    busy = insts * LEAN_LOOP_CYCLES_PER_INST / (SIMDS * CLOCK_GHZ * 1e9) / (k_avg * fpl * 1e-3)
                    roofline["issue_weighted"] = {"frac": round(busy, 4), "cycles_per_instruction": round(LEAN_LOOP_CYCLES_PER_INST, 3),
                                                  "source": "profiles/r02_valu_issue_rates.txt x the loop's instruction mix (DESIGN.md section 5)"}
                roofline["traffic"] = pmc.get("hbm_bytes_per_launch")
                roofline["traffic_source"] = {"file": "profiles/pmc_counters.json", "summary": pmc.get("source"), "commit": pmc.get("commit"),
                                              "device_source_hash": pmc.get("device_source_hash"),
                                              "matches_this_build": pmc.get("device_source_hash") == device_source_hash(),
                                              "how": "rocprofv3 --pmc, separate passes (FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE, SQ_*), counters only"}
            else:
                roofline["note"] = (f"no PMC entry for config {args.config} / {kernel_name} / {order_key} in profiles/pmc_counters.json: "
                                    "achieved and frac cannot be stated for this combination")
        orbit = None
        if world == 1 and not triangles and fif == 1 and not use_comm:
            n_orbit = 240 if args.orbit_frames is None else args.orbit_frames
            if n_orbit > 0:
                # a camera that moves: theta advances 0.01 rad per frame, plain stream launches (no graph), the launch-order
                # table is rebuilt after every --order-period-th frame from that frame's costs
                th0, ph0, r0, tg0 = float(cam.theta), float(cam.phi), float(cam.radius), cam.getTarget()
                oframes = []
                for i in range(n_orbit):
                    c2 = rto.Camera(th0 + 0.01 * i, ph0, r0)
                    c2.setTarget(tg0)
                    oframes.append(rto.make_frame(c2.getView(), c2.getPos(), W / H, 45.0, W, H))
                obuf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
                ctx.timing_begin(-1)                      # no per-launch events: only the pair around the whole sequence below
                for fr in oframes[:16]:
                    render_to(obuf.data_ptr(), stream.cuda_stream, fr)
                torch.cuda.synchronize()
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t_o = time.perf_counter()
                ea.record(stream)
                for fr in oframes:
                    render_to(obuf.data_ptr(), stream.cuda_stream, fr)
                eb.record(stream)
                torch.cuda.synchronize()
                wall = time.perf_counter() - t_o
                ctx.timing_begin(0)
                orbit = {"frames": n_orbit, "rad_per_frame": 0.01, "launches": "plain stream launches, launch-order table rebuilt when the geometry's tile box moves and at "
                         f"least every {args.order_period}-th frame (k_order_build inside the timed region)",
                         "ms_per_frame": round(wall / n_orbit * 1e3, 5), "gpu_ms_per_frame": round(ea.elapsed_time(eb) / n_orbit, 5),
                         "Mrays_per_s": round(rays * n_orbit / wall / 1e6, 1)}
                if fpl > 1:
                    # the same orbit when the cameras of fpl consecutive frames are known together (a recorded path, an offline
                    # sequence): fpl frames per launch, each with its own camera, plain launches
                    groups = [rto.Context.frame_array(oframes[i:i + fpl]) for i in range(0, n_orbit - n_orbit % fpl, fpl)]
                    ctx.timing_begin(-1)
                    for ga in groups[:4]:
                        ctx.render_batch_device(ga, batch_buf.data_ptr(), H * W * 16, None, False, stream.cuda_stream)
                    torch.cuda.synchronize()
                    t_o = time.perf_counter()
                    for ga in groups:
                        ctx.render_batch_device(ga, batch_buf.data_ptr(), H * W * 16, None, False, stream.cuda_stream)
                    torch.cuda.synchronize()
                    wall = time.perf_counter() - t_o
                    ctx.timing_begin(0)
                    nb = len(groups) * fpl
                    orbit["frames_per_launch_%d" % fpl] = {"frames": nb, "ms_per_frame": round(wall / nb * 1e3, 5), "Mrays_per_s": round(rays * nb / wall / 1e6, 1)}
        pcie = None
        if world == 1 and not triangles:
            # the C ABI's host-buffer entry point (kernel + D2H over PCIe): informational, never `value`
            ts = []
            for _ in range(5):
                t = time.perf_counter()
                ctx.render_host(frame)
                ts.append(time.perf_counter() - t)
            ts.sort()
            pcie = {"ms_per_frame": round(ts[2] * 1e3, 4), "Mrays_per_s": round(rays / ts[2] / 1e6, 1),
                    "what": "rto_render_host: kernel + device-to-host copy of the RGBA32F frame into pageable memory"}
        cpu = None
        verified = False
        skip_cpu = args.cpu_frames == 0
        if not (skip_cpu and args.no_verify):
            scene = {"W": W, "H": H, "mode": cfg["mode"], "nodes": ctx.download_nodes(), "gmin": grid.min, "voxel": grid.voxelSize,
                     "view": view, "pos": pos}
            if triangles:
                from oracle import orc   # the checker builds its own triangle buffer from the voxels

                og = orc.Grid(grid.dims, grid.min, grid.voxelSize, vox)
                scene["tris"], scene["tri_offset"] = orc.build_leaf_triangles(og, scene["nodes"])
            default_frames = {"2": 40, "4": 12, "4synthetic": 10, "5": 4}[args.config]
            if world > 1 or skip_cpu:
                args.cpu_frames = 0
            cpu, want, ost = cpu_baseline(args, scene, default_frames)
            if ost["pops"] != stats["pops"] or ost["hits"] != stats["hits"]:
                sys.exit("bench: GPU and oracle disagree on the pop / hit counts -- result void")
            if img.cpu().numpy().tobytes() != np.ascontiguousarray(want, np.float32).tobytes():
                sys.exit("bench: the timed frame differs from the oracle's -- result void")
            verified = True
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
                "workload": f"BASELINE config {args.config}: {what}, octree to min-leaf 1 ({info.num_nodes} nodes), "
                            f"{W}x{H} primary rays{', Marching-Cubes leaf triangles + 1 shadow ray per hit' if triangles else ''}, {camtxt}, fov 45",
                "parallelism": (("1 GPU" + (f", {fpl} consecutive frames per kernel launch ({'rto_render_triangles_batch_device' if triangles else 'rto_render_batch_device'})" if fpl > 1 else "") + (f", launches replayed from a HIP graph of {gframes} consecutive frames" if use_graph else ", plain stream launches")) if fif == 1 else f"1 GPU, {fif} frames in flight on {fif} HIP streams") if world == 1 else
                               (f"screen split over {world} GPUs, {args.band_rows}-row bands round-robin, rto_comm_submit: ONE grouped RCCL send/recv per {'frame' if fpg == 1 else f'{fpg} frames'} "
                                f"into rank 0 (4-byte Lambert term per pixel, the columns of the geometry's rectangle only{'; rank 0 gathers and assembles, ranks 1..N-1 render' if (args.rehearse_world or world) >= 4 else ''}), batch k's gather overlaps batch k+1's render" if use_comm else "") + ("" if use_comm else f"screen split over {world} GPUs, {args.band_rows}-row bands "
                                                          f"round-robin, 1 RCCL gather per {'frame' if fpg == 1 else f'{fpg} frames'} ({'4-byte Lambert term' if args.payload == 'shade' else 'RGBA32F'} per pixel"
                                                          f"{', gather k overlaps render k+1' if pipelined else ''}"
                                                          f"{f', {fpg} consecutive frames per gather' if fpg > 1 else ''}"
                                                          f"{f', {npipe} such pipelines on {npipe} HIP streams take the batches in turn' if npipe > 1 else ''})"),
                "kernel": args.kernel,
                "clock_ramp": (f"{ramp_frames} untimed frames of the same workload ({args.ramp_ms:.0f} ms) before the {args.warmup} warm-up frames, "
                               "so that a cold GPU has reached its working clock") if ramp_frames else
                              (f"{primed} untimed frames in full batches through the pipe before the {args.warmup} warm-up frames (RCCL connections, buffers, clocks)" if primed else "none"),
            },
            "hit_rays": stats["hits"], "capped_rays": stats["capped"],
            "verified_against_oracle": verified,
            "device": ctx.device_name,
        }
        if fpl > 1:
            # the same frames one launch each, in a graph of their own (outside the timed region): what a caller gets that must
            # show frame i before it knows frame i+1's camera
            n1 = max(1, min(50, args.steps))
            one_ms = None
            try:
                g1 = torch.cuda.CUDAGraph()
                ctx.timing_begin(0)
                for _ in range(3):
                    render_to(batch_buf.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
                with torch.cuda.graph(g1, stream=stream):
                    for _ in range(n1):
                        render_to(batch_buf.data_ptr(), stream.cuda_stream)
                g1.replay(); torch.cuda.synchronize()
                reps1 = max(1, min(40, args.steps // n1))
                e1a, e1b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e1a.record(stream)
                for _ in range(reps1):
                    g1.replay()
                e1b.record(stream)
                torch.cuda.synchronize()
                one_ms = e1a.elapsed_time(e1b) / (reps1 * n1)
            except Exception as e:
                print(f"bench: one-frame-per-launch leg skipped ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
            if one_ms:
                result["one_frame_per_launch"] = {"ms_per_frame": round(one_ms, 5), "Mrays_per_s": round(rays / one_ms / 1e3, 1), "frames": reps1 * n1,
                                                  "what": "one kernel launch per frame (rto_render_device / rto_render_triangles_device), replayed from a HIP graph: a frame's kernel is as long as its "
                                                          "deepest tile's chain of dependent node visits; `value` launches several frames together instead"}
        if args.rehearse_world > 1:
            result["rehearsal"] = {"as_rank": args.rehearse_rank, "of": args.rehearse_world,
                                   "what": "ONE GPU doing what that rank does per frame (render its bands, grouped send/recv with itself) plus the assembly "
                                           "(in a rehearsal the one GPU is also the assembling rank 0); "
                                           "value / ms_per_step are the per-rank pipeline rate of the split, not a frame rate of this machine"}
        if roofline is not None:
            result["roofline"] = roofline
        if orbit is not None:
            result["orbit"] = orbit
        if latency is not None:
            result["single_frame_latency"] = latency
        if pcie is not None:
            result["pcie_inclusive"] = pcie
        if cpu is not None:
            result["cpu_baseline"] = cpu
            result["speedup_vs_cpu_all_cores"] = round(result["value"] / cpu["value"], 1)
        print(json.dumps(result), flush=True)

    if comm is not None:
        comm.close()
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
