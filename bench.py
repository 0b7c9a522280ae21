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
N = 1    : `value` is ONE KERNEL LAUNCH PER FRAME (rto_render_device: what the reference's renderSceneCompute does per
           call), one frame strictly after the other on one stream, K plain stream launches without per-launch events
           (--graph-frames n: replayed from a HIP graph of n frames captured before the timed region).  Secondary figures in the same
           line, all outside the timed region: `frames_per_launch` (several frames per kernel launch, identical and
           distinct cameras -- the throughput form for callers that know the next cameras), `dropin_call` (the C++ class
           exactly as main.cpp:1357-1363 calls it: renderSceneComputeWithCulling with a frustum update every frame),
           `orbit` (a moving camera, plain launches), `pcie_inclusive` (host-buffer entry point).  The default run (config 2) also
           carries, each measured after the headline and checked bitwise against the oracle on one frame: `configs` (BASELINE
           configs 4 and 5 as 20 plain launches each), `cold` (the first 20 frames of a fresh context after 0.5 s of idle: no
           clock ramp, no launch-order history), `random_cameras` (64 seeded camera jumps, one launch each: mean and max),
           `split_rehearsal` (this GPU playing ranks 0 and 1 of a 2 / 4 / 8-GPU split).  --no-extras skips them.
N > 1    : `python3 bench.py --gpus N` starts N fresh ranks itself (python -m torch.distributed.run, one per GPU; the
           parent process never touches a GPU) and relays rank 0's line; launched under torch.distributed.run it is a
           rank.  Every rank holds the octree and renders its round-robin bands; ONE grouped RCCL send/recv per batch of
           frames lands the parts on rank 0, which re-interleaves them -- all of it below the C boundary (rto_comm_* in
           include/rto_hip.h; torch.distributed (gloo) only carries the 128-byte communicator id, the barriers and the max
           over ranks).  Strong scaling: one frame split N ways.  If the communicator cannot be made, or its first (probe)
           batch fails, on any rank: every rank renders whole frames on its own GPU and the line says so (`split_error`,
           `split_fallback`, "scaling": "weak") -- independent replicas with their caveat instead of no line.

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
import signal
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
MEASURED_CYCLES_PER_WAVE_INST = 2.95   # the traversal loop's mix at the issue costs measured on a saturated SIMD (profiles/r02_valu_issue_rates.txt, tools/ubench/valu_rate4.hip): round 5's exact-grid loop, ~326 cycles / ~111 instructions (rounds 2-4: 401.5 / 120 = 3.35)
R02_VALU_INSTS = {"2": 24.12e6, "4": 55.13e6, "5": 328.1e6}   # SQ_INSTS_VALU per single-frame launch of round 2's kernels (profiles/r02_config*_summary.json)
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
    ap.add_argument("--no-verify", action="store_true",
                    help="skip rank 0's comparison of the timed frame with the CPU oracle (one oracle frame, outside the timed region)")
    ap.add_argument("--order", choices=["temporal", "centre-out"], default="temporal",
                    help="tile launch order of the packed kernel (scheduling only; pixels are identical)")
    ap.add_argument("--order-period", type=int, default=8, help="temporal order: rebuild the table every n-th frame")
    ap.add_argument("--no-pipeline", action="store_true", help="N>1: finish the gather of a batch before rendering the next")
    ap.add_argument("--frames-per-gather", type=int, default=8,
                    help="N>1: consecutive frames whose parts travel in ONE gather (fewer, larger collectives); 1 = one gather per frame")
    ap.add_argument("--graph-frames", type=int, default=None,
                    help="N=1: capture this many consecutive single-frame launches in a HIP graph (before the timed region) and replay it; "
                         "0 (default) = plain stream launches without per-launch events")
    ap.add_argument("--ramp-ms", type=float, default=300.0,
                    help="N=1: untimed frames of the same workload rendered for this long before the W warm-up frames, so that a cold "
                         "GPU has reached its working clock (disclosed in the line as `clock_ramp`); 0 = none")
    ap.add_argument("--orbit-frames", type=int, default=None,
                    help="N=1, octree configs: extra figure with the camera orbiting 0.01 rad per frame, plain launches (default 240; 0 = skip)")
    ap.add_argument("--frames-per-launch", type=int, default=4,
                    help="N=1: frames per kernel launch of the SECONDARY throughput figure `frames_per_launch` (rto_render_batch_device, at most 8; 1 = skip)")
    ap.add_argument("--dropin-frames", type=int, default=200,
                    help="N=1, octree configs: frames of the `dropin_call` leg -- the C++ class, renderSceneComputeWithCulling(update=true) per frame (0 = skip)")
    ap.add_argument("--rehearse-world", type=int, default=0,
                    help="with --gpus 1 --force-comm: render / ship / assemble as a rank of that many GPUs (rto_comm_debug_rehearse); "
                         "the line then reports the per-rank cost of the split, NOT a frame rate (only 1/N of every frame is rendered)")
    ap.add_argument("--rehearse-rank", type=int, default=0, help="with --rehearse-world: the rank to play (from 4 GPUs on rank 0 only gathers "
                                                                  "and assembles, ranks 1..N-1 render)")
    ap.add_argument("--inject-split-failure", action="store_true", help=argparse.SUPPRESS)     # tests: the communicator's first batch "fails" (exercises the replicas fallback)
    ap.add_argument("--force-comm", action="store_true",
                    help="N=1: drive the frames through rto_comm_* with a one-rank RCCL communicator (rehearses the N>1 code path on one GPU)")
    ap.add_argument("--launch-timeout", type=float, default=300.0,
                    help="N>1: seconds the parent waits for its ranks before it kills their process group and exits 124 (0 = no limit)")
    ap.add_argument("--launcher-test-command", default="", help=argparse.SUPPRESS)     # tests: python source run in place of the ranks
    ap.add_argument("--no-tile-mask", action="store_true", help="A/B: switch the occupancy mask of the default kernels off (rto_debug_set_tile_mask)")
    ap.add_argument("--no-extras", action="store_true",
                    help="N=1, config 2: skip the legs behind the headline (configs 4 and 5, cold context, random cameras, split rehearsal)")
    ap.add_argument("--extras-steps", type=int, default=20, help="timed launches of each `configs` leg and of the `cold` leg")
    ap.add_argument("--launcher-dry-run", action="store_true", help="--gpus N without WORLD_SIZE: print the child command line and exit")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ N > 1: self-launch
def child_command(args, argv, port: int) -> list[str]:
    """The N fresh ranks `python3 bench.py --gpus N ...` starts: one process per GPU over RCCL, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


COMM_TIMEOUT_MS = 120000     # every rto_comm flush of a bench run: a rank whose peers never arrive fails (RTO_E_TIMEOUT) instead of hanging


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
    if args.launcher_test_command:                 # tests: a stand-in for the ranks (a process that sleeps, one that prints a line)
        cmd = [sys.executable, "-c", args.launcher_test_command]
    # The ranks run in a process group of their own, under a watchdog: a rank that dies before its ncclSend leaves the others
    # waiting in a collective for ever (rto_comm_flush_timeout bounds that inside a rank; this bounds the whole job).  On expiry
    # exactly the group started here is killed -- never a pattern, never an exec -- and the exit status is non-zero.
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    timed_out = False
    try:
        out, err = p.communicate(timeout=args.launch_timeout if args.launch_timeout > 0 else None)
    except subprocess.TimeoutExpired:
        timed_out = True
        for sig, wait in ((signal.SIGTERM, 10), (signal.SIGKILL, 10)):
            try:
                os.killpg(p.pid, sig)              # the session leader's pid is the group's id: the children started above, nothing else
            except ProcessLookupError:
                break
            try:
                p.wait(timeout=wait)
                break
            except subprocess.TimeoutExpired:
                continue
        try:
            out, err = p.communicate(timeout=10)
        except subprocess.TimeoutExpired:
            out, err = "", ""
    lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    if timed_out or p.returncode != 0 or len(lines) != 1:
        sys.stderr.write(err[-8000:])
        sys.stderr.write(out[-4000:])
        if timed_out:
            sys.stderr.write(f"\nbench: the {args.gpus}-rank job did not finish within --launch-timeout {args.launch_timeout:g} s: its process group was killed\n")
            return 124
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
        short = f"{dim}^3 test sphere"
        what = f"{dim}^3 test-sphere voxel grid (main.cpp:337-372), octree to min-leaf 1"
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
            short = "sceneCache.bin resampled to 512x512x128 (SYNTHETIC)"
            what = "sceneCache.bin resampled nearest-neighbour to 512x512x128 (SYNTHETIC stand-in for the grid BASELINE.json names)"
            camtxt = "Camera(0.6,0.5,4500) on the grid centre"
        else:
            grid = rto.VoxelGrid.from_array(data, gmin, voxel)
            cam = rto.Camera(0.6, 0.5, 3500.0)
            short = "sceneCache.bin as shipped 425x243x29"
            what = "sceneCache.bin as shipped (425x243x29 voxels, octree root 512)"
            camtxt = "Camera(0.6,0.5,3500)"
    return grid, cam, short, what, camtxt


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


def pmc_entry(config: str, kernel_name: str, order: str, fpl: int):
    """Counters of the committed rocprofv3 --pmc passes for this (config, kernel, launch order, frames per launch), or None."""
    if not os.path.exists(PMC_FILE):
        return None
    with open(PMC_FILE) as f:
        table = json.load(f)
    for e in table.get("entries", []):
        if e.get("config") == config and e.get("kernel") == kernel_name and e.get("order") == order and int(e.get("frames_per_launch") or 1) == fpl:
            return e
    return None


def seeded_camera(rto, np, seed, grid):
    """The camera generator of tests/test_gpu_parity.py::test_random_cameras_at_full_size (same seeds, same draws): far, near, inside
    the volume, looking past it; fov 30 / 45 / 70."""
    dims = np.array(grid.dims, np.float32)
    ext = float(dims.max() * np.float32(grid.voxelSize))
    centre = np.asarray(grid.min, np.float32) + 0.5 * dims * np.float32(grid.voxelSize)
    rng = np.random.default_rng(77000 + seed)
    kind = ("far", "near", "inside", "past")[seed % 4]
    radius = ext * {"far": rng.uniform(1.5, 5.0), "near": rng.uniform(0.55, 1.0), "inside": rng.uniform(0.05, 0.45), "past": rng.uniform(0.8, 2.0)}[kind]
    cam = rto.Camera(float(rng.uniform(0, 6.28)), float(rng.uniform(-1.4, 1.4)), float(radius))
    aim = rng.uniform(-0.15, 0.15, 3) if kind != "past" else rng.uniform(0.6, 1.2, 3) * rng.choice([-1.0, 1.0], 3)
    cam.setTarget(centre + aim.astype(np.float32) * ext)
    fov = float(rng.choice([30.0, 45.0, 70.0]))
    return cam, fov, kind


def valu_frac(config, kernel_name, ms):
    """roofline.frac of a leg from the committed PMC entry of (config, kernel): SQ_INSTS_VALU per launch / ms / the 2-cycle VALU peak."""
    pmc = pmc_entry(config, kernel_name, "temporal", 1)
    if pmc is None or not ms:
        return None
    insts = float(pmc["SQ_INSTS_VALU"])
    return {"bound": "valu_issue", "frac": round(insts / (ms * 1e-3) / 1e9 / VALU_PEAK_GINST, 4), "valu_insts_per_launch": int(insts),
            "lane_utilisation": round(float(pmc["SQ_THREAD_CYCLES_VALU"]) / (64.0 * insts), 3) if pmc.get("SQ_THREAD_CYCLES_VALU") else None,
            "traffic": pmc.get("hbm_bytes_per_launch"), "pmc_matches_this_build": pmc.get("device_source_hash") == device_source_hash()}


def extras(args, local_rank, stream, main_ctx, main_grid, main_frame, main_img, small=False, out=None):
    """The legs behind the headline of the default run: every figure one plain launch per frame on `stream`,
    every leg checked bitwise against the oracle on one frame, outside every timed region.  small: the contract test's sizes
    (a reduced headline scene brings reduced legs: config 5 at 64^3 / 640x360, config 4 at 480x270, 8 cameras, N = 2 only)."""
    import argparse as _ap

    import numpy as np
    import torch

    import ray_tracing_octrees_amd as rto
    from oracle import orc   # the checker, outside every timed region
    from ray_tracing_octrees_amd import hip as _hip

    cores = host_cores()
    out = {} if out is None else out          # filled leg by leg: a leg that raises leaves the finished ones in the caller's hands
    K = max(1, args.extras_steps)

    def timed_block(fn, n):
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record(stream); eb.record(stream); torch.cuda.synchronize()
        t = time.perf_counter()
        ea.record(stream)
        for _ in range(n):
            fn()
        eb.record(stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3, ea.elapsed_time(eb) / n

    # ---- configs 4 and 5: scene built by rto_build_octree in a context of its own, short disclosed ramp, K plain launches
    configs = {}
    for name in ("4", "5"):
        t_leg = time.perf_counter()
        cfg = CONFIGS[name]
        W, H = cfg["width"], cfg["height"]
        if small:
            W, H = (640, 360) if name == "5" else (480, 270)
        tri = cfg["mode"] == "triangles"
        grid, cam, short, what, camtxt = build_scene(_ap.Namespace(config=name, dim=64 if (small and name == "5") else None))
        c = rto.Context(local_rank)
        t_b = time.perf_counter()
        c.build_octree(grid.data, grid.min, grid.voxelSize)
        if tri:
            c.build_leaf_triangles(None)
        c.synchronize()
        build_ms = (time.perf_counter() - t_b) * 1e3
        fr = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
        fb = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
        render = (lambda: c.render_triangles_device(fr, fb.data_ptr(), True, None, stream.cuda_stream)) if tri else \
                 (lambda: c.render_device(fr, fb.data_ptr(), None, stream.cuda_stream))
        c.timing_begin(-1)
        ramp = 0
        t_end = time.perf_counter() + 0.1
        while time.perf_counter() < t_end:
            for _ in range(5):
                render()
            torch.cuda.synchronize(); ramp += 5
        wall_ms, gpu_ms = timed_block(render, K)
        c.timing_begin(K)
        for _ in range(K):
            render()
        kms = sorted(float(x) for x in c.timing_read())
        c.timing_begin(-1)
        got = fb.cpu().numpy()
        nodes = c.download_nodes()
        if tri:
            og = orc.Grid(grid.dims, grid.min, grid.voxelSize, grid.data)
            tris, off = orc.build_leaf_triangles(og, nodes)
            want, ost = orc.render_triangles(nodes, tris, off, grid.min, grid.voxelSize, cam.getView(), cam.getPos(), W / H, 45.0, W, H, shadow=True, nthreads=cores)
            kname = "k_trace_lean_triangles"
        else:
            want, ost = orc.render(nodes, grid.min, grid.voxelSize, cam.getView(), cam.getPos(), W / H, 45.0, W, H, nthreads=cores)
            kname = "k_trace_lean"
        ok = got.tobytes() == np.ascontiguousarray(want, np.float32).tobytes()
        if not ok:
            sys.exit(f"bench: config {name}'s frame differs from the oracle's -- result void")
        configs[name] = {
            "workload": f"cfg{name}: {short}, {c.info().num_nodes} nodes, {W}x{H}{', MC triangles+shadow' if tri else ''}, {camtxt} fov45",
            "steps": K, "ms_per_frame": round(wall_ms, 5), "gpu_ms_per_frame": round(gpu_ms, 5), "Mrays_per_s": round(W * H / wall_ms / 1e3, 1),
            "kernel_ms_event_pair_median": round(kms[len(kms) // 2], 5),
            "roofline": None if small else valu_frac(name, kname, gpu_ms), "kernel": kname,
            "hit_rays": int(ost["hits"]), "verified_against_oracle": True,
            "octree_build_ms": round(build_ms, 2), "built_by": "rto_build_octree" + (" + rto_build_leaf_triangles" if tri else "") + " (GPU; includes the H2D copy of the voxels)",
            "clock_ramp": f"{ramp} untimed frames (100 ms) in a fresh context", "launches": "plain stream launches, one per frame, no per-launch events",
            "leg_seconds": None}
        c.close()
        del fb
        configs[name]["leg_seconds"] = round(time.perf_counter() - t_leg, 1)
    out["configs"] = configs

    # ---- cold: a fresh context of the headline's scene, 0.5 s of idle, then its first K frames with a per-launch event pair
    W, H = main_frame.width, main_frame.height
    c = rto.Context(local_rank)
    c.build_octree(main_grid.data, main_grid.min, main_grid.voxelSize)
    c.synchronize()
    fb = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    time.sleep(0.5)
    c.timing_begin(K)
    t = time.perf_counter()
    for _ in range(K):
        c.render_device(main_frame, fb.data_ptr(), None, stream.cuda_stream)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / K * 1e3
    kms = [float(x) for x in c.timing_read()]
    c.timing_begin(-1)
    same = bool(torch.equal(fb, main_img))
    if not same:
        sys.exit("bench: the cold context's frame differs from the timed frame -- result void")
    out["cold"] = {"frames": K, "ms_per_frame": round(wall, 5), "Mrays_per_s": round(W * H / wall / 1e3, 1),
                   "first_frame_kernel_ms": round(kms[0], 5), "kernel_ms_mean": round(sum(kms) / len(kms), 5), "kernel_ms_max": round(max(kms), 5),
                   "kernel_ms_each": [round(x, 4) for x in kms], "frame_equals_timed_frame": same,
                   "what": "a FRESH context of the headline's scene (no launch-order table, no tile costs, no mask history), the GPU idle for 0.5 s, "
                           f"then its first {K} frames as plain launches with a HIP event pair around each kernel: no clock ramp, no warm-up"}
    c.close()

    # ---- random cameras: 64 seeded jumps (tests' generator), ONE launch each on the headline's context
    n_cam = 8 if small else 64
    frames, kinds = [], []
    for seed in range(n_cam):
        cam, fov, kind = seeded_camera(rto, np, seed, main_grid)
        frames.append((rto.make_frame(cam.getView(), cam.getPos(), W / H, fov, W, H), cam, fov))
        kinds.append(kind)
    # pass 1: the GPU at its working clock (100 ms of the headline's frame), then the 64 jumps back to back: kernel time per camera
    main_ctx.timing_begin(-1)
    t_end = time.perf_counter() + (0.0 if small else 0.1)
    while True:
        for _ in range(10):
            main_ctx.render_device(main_frame, fb.data_ptr(), None, stream.cuda_stream)
        torch.cuda.synchronize()
        if time.perf_counter() >= t_end:
            break
    main_ctx.timing_begin(n_cam)
    for fr, cam, fov in frames:
        main_ctx.render_device(fr, fb.data_ptr(), None, stream.cuda_stream)
    torch.cuda.synchronize()
    kms = [float(x) for x in main_ctx.timing_read()]
    main_ctx.timing_begin(-1)
    # pass 1b: every camera once more, 5 frames each: the time of the 5th is what the jump costs once the launch order has settled
    main_ctx.timing_begin(n_cam * 5)
    for fr, cam, fov in frames:
        for _ in range(5):
            main_ctx.render_device(fr, fb.data_ptr(), None, stream.cuda_stream)
    torch.cuda.synchronize()
    settled = [float(x) for x in main_ctx.timing_read()][4::5]
    main_ctx.timing_begin(-1)
    # pass 2: the same jumps one call at a time, the host waiting for each frame (what an interactive caller sees; the GPU idles
    # between calls and drops its clock); the first four frames go to the checker
    for _ in range(10):
        main_ctx.render_device(main_frame, fb.data_ptr(), None, stream.cuda_stream)
    torch.cuda.synchronize()
    walls = []
    check = {}
    for i, (fr, cam, fov) in enumerate(frames):
        t = time.perf_counter()
        main_ctx.render_device(fr, fb.data_ptr(), None, stream.cuda_stream)
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t) * 1e3)
        if i < 4:
            check[i] = fb.cpu().numpy()
    nodes = main_ctx.download_nodes()
    for i, got in check.items():
        fr, cam, fov = frames[i]
        want, _ = orc.render(nodes, main_grid.min, main_grid.voxelSize, cam.getView(), cam.getPos(), W / H, fov, W, H, nthreads=cores)
        if got.tobytes() != np.ascontiguousarray(want, np.float32).tobytes():
            sys.exit(f"bench: random camera {i} differs from the oracle's frame -- result void")
    worst = max(range(n_cam), key=lambda i: kms[i])
    out["random_cameras"] = {"cameras": n_cam, "kernel_ms_mean": round(sum(kms) / n_cam, 5), "kernel_ms_max": round(kms[worst], 5),
                             "kernel_ms_median": round(median(kms), 5), "max_at_seed": worst, "max_kind": kinds[worst],
                             "kernel_ms_mean_settled": round(sum(settled) / n_cam, 5), "kernel_ms_max_settled": round(max(settled), 5),
                             "jump_penalty_mean": round(sum(kms) / sum(settled), 3),
                             "call_ms_mean": round(sum(walls) / n_cam, 5), "call_ms_max": round(max(walls), 5),
                             "kernel_ms_mean_by_kind": {k: round(sum(kms[i] for i in range(n_cam) if kinds[i] == k) / max(1, kinds.count(k)), 5) for k in ("far", "near", "inside", "past")},
                             "frames_verified_against_oracle": sorted(check),
                             "what": f"{n_cam} seeded cameras of the headline's scene (the generator of tests' test_random_cameras_at_full_size: far / near / inside the "
                                     "volume / looking past it, fov 30 / 45 / 70), each a JUMP from the one before: one launch per camera, its launch order and "
                                     "rim learned from another camera; *_settled: the 5th consecutive frame of the same camera (its own order and rim), so jump_penalty_mean = what the stale "
                                     "schedule costs; kernel_ms = the traversal kernel (event pair) with the jumps launched back to back on a GPU at its working clock, "
                                     "call_ms = a second pass, launch + wait on the host per camera (the GPU idles between calls)"}
    del fb

    # ---- split rehearsal: this GPU playing ranks 0 and 1 of a 2 / 4 / 8-GPU split (rto_comm_debug_rehearse), 8 frames per gather
    try:
        import ctypes
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)                                   # RCCL's banner must not reach this process's one-line stdout
        try:
            comm = _hip.Comm(main_ctx, 1, 0, _hip.comm_unique_id(), band_rows=args.band_rows)
        finally:
            ctypes.CDLL(None).fflush(None)
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        fpg = 8
        arr = rto.Context.frame_array([main_frame] * fpg)
        cf = torch.empty((fpg, H, W, 4), dtype=torch.float32, device="cuda")
        comm.debug_set_rehearsal_clear(False)
        comm.debug_set_timing(True)
        reh = {}
        for n in ((2,) if small else (2, 4, 8)):
            for r in (0, 1):
                comm.debug_rehearse(n, r)
                t_end = time.perf_counter() + (0.0 if small else 0.08)          # untimed: back to the working clock, buffers of this split
                while True:
                    for _ in range(6):
                        comm.submit(arr, cf.data_ptr(), H * W * 16, _hip.RESIDENT_OCTREE)
                    comm.flush(COMM_TIMEOUT_MS)
                    if time.perf_counter() >= t_end:
                        break
                t = time.perf_counter()
                nb = 40
                for _ in range(nb):
                    comm.submit(arr, cf.data_ptr(), H * W * 16, _hip.RESIDENT_OCTREE)
                comm.flush(COMM_TIMEOUT_MS)
                ms = (time.perf_counter() - t) / (nb * fpg) * 1e3
                rm, gm = comm.debug_last_timing()
                sent, _whole = comm.debug_last_payload()
                reh[f"N{n}_rank{r}"] = {"ms_per_frame": round(ms, 5), "render_ms_per_frame": round(rm / fpg, 5), "gather_assemble_ms_per_frame": round(gm / fpg, 5),
                                        "payload_bytes_per_frame": sent * 4 // fpg}
        comm.debug_rehearse(0)
        comm.close()
        reh["what"] = ("ONE GPU doing what rank r of N does per frame (render its bands, grouped send/recv with itself, assembly), 8 frames per gather, pipelined: "
                       "per-rank cost of the split, no peer traffic; from 4 GPUs on rank 0 only gathers and assembles")
        out["split_rehearsal"] = reh
    except Exception as e:                                # no RCCL in this process: say so, keep the line
        out["split_rehearsal"] = {"skipped": f"{type(e).__name__}: {e}"}
    return out


def _trace(msg):
    if os.environ.get('RTO_BENCH_TRACE'):
        print('[bench] ' + msg, file=sys.stderr, flush=True)


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if xs else None


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.graph_frames is None:
        args.graph_frames = 0          # plain launches without per-launch events beat the graph replay at every K (tools/steps_sweep.sh: K = 20: 0.0405 vs 0.0435 ms)
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
    from ray_tracing_octrees_amd import hip as _hip

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import datetime

        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # control plane only (communicator id, barriers, max over ranks): the frames travel through rto_comm_* (RCCL).
        # A rank that dies must not leave the others waiting for ten minutes in a collective.
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=180))

    # ---- scene ------------------------------------------------------------------------------------
    cfg = CONFIGS[args.config]
    W, H = args.width or cfg["width"], args.height or cfg["height"]
    triangles = cfg["mode"] == "triangles"
    grid, cam, short, what, camtxt = build_scene(args)
    view, pos = cam.getView(), cam.getPos()
    frame = rto.make_frame(view, pos, W / H, 45.0, W, H)
    rays = W * H

    ctx = rto.Context(local_rank)
    vox = grid.data
    ctx.build_octree(vox, grid.min, grid.voxelSize)                 # createOctreeFromVoxelGrid + setOctree on the GPU (row N4)
    if triangles:
        ctx.build_leaf_triangles(None)                              # MarchingCubesRenderer per leaf, on the GPU (row N2)
    ctx.set_kernel({"auto": rto.KERNEL_AUTO, "packed": rto.KERNEL_PACKED, "persistent": rto.KERNEL_PACKED_PERSISTENT,
                    "packed_v3": rto.KERNEL_PACKED_V3, "packed_v1": rto.KERNEL_PACKED_V1, "generic": rto.KERNEL_GENERIC}[args.kernel])
    ctx.set_launch_order(1 if args.order == "temporal" else 0, args.order_period)
    if args.no_tile_mask:
        ctx.debug_set_tile_mask(False)
    info = ctx.info()
    use_comm = world > 1 or args.force_comm
    pipelined = use_comm and not args.no_pipeline
    comm = None
    split_error = None
    if use_comm:
        ids = [_hip.comm_unique_id() if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(ids, src=0)
        # RCCL prints a version banner to STDOUT when its first communicator is made: this process's stdout carries ONE JSON line, so
        # file descriptor 1 points at stderr while the communicator is created (C stdio flushed before it is restored)
        import ctypes
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        # If the communicator cannot be made, or its first batch does not come back, on ANY rank, every rank falls back to rendering whole
        # frames on its own GPU (independent replicas: frames are the path's independent units) and the line says so (`split_error`,
        # "scaling": "weak"): a number with its caveat instead of no line.  A hang inside RCCL is ended by the flush limit / the launcher.
        try:
            comm = _hip.Comm(ctx, world, rank, ids[0], band_rows=args.band_rows)
            probe = torch.empty((H, W, 4), dtype=torch.float32, device="cuda") if rank == 0 else None
            comm.submit(rto.Context.frame_array([frame]), probe.data_ptr() if rank == 0 else 0, H * W * 16,
                        _hip.RESIDENT_TRIANGLES_SHADOW if triangles else _hip.RESIDENT_OCTREE)
            comm.flush(COMM_TIMEOUT_MS)
            if args.inject_split_failure:
                raise RuntimeError("injected by --inject-split-failure")
            del probe
        except Exception as e:
            split_error = f"rank {rank}: {type(e).__name__}: {e}"
            print(f"bench: the screen split cannot run here ({split_error})", file=sys.stderr, flush=True)
        finally:
            ctypes.CDLL(None).fflush(None)
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        if dist is not None:
            errs = [None] * world
            dist.all_gather_object(errs, split_error)
            split_error = next((e for e in errs if e), None)
        if split_error is not None:
            if comm is not None:
                try:
                    comm.close()
                except Exception:
                    pass
            comm, use_comm, pipelined = None, False, False
        if args.rehearse_world > 1 and comm is not None:
            if world != 1:
                raise SystemExit("--rehearse-world needs --gpus 1 --force-comm")
            comm.debug_rehearse(args.rehearse_world, args.rehearse_rank)   # this GPU plays that rank: per-rank cost of the split, no peer traffic
            comm.debug_set_rehearsal_clear(False)                          # the per-batch clear of the absent ranks' rows is test scaffolding, not a rank's work
        comm_mode = _hip.RESIDENT_TRIANGLES_SHADOW if triangles else _hip.RESIDENT_OCTREE

    stream = torch.cuda.Stream()          # kernels and events order themselves on this side stream
    torch.cuda.set_stream(stream)
    fbuf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")

    def render_to(buf_ptr, fr=None):
        """one frame = one kernel launch (what renderSceneCompute does per call)"""
        if triangles:
            ctx.render_triangles_device(fr or frame, buf_ptr, True, None, stream.cuda_stream)
        else:
            ctx.render_device(fr or frame, buf_ptr, None, stream.cuda_stream)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x: float) -> float:
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    fpg = max(1, args.frames_per_gather) if pipelined else 1
    comm_frames = torch.empty((fpg, H, W, 4), dtype=torch.float32, device="cuda") if (use_comm and rank == 0) else None
    comm_arr = rto.Context.frame_array([frame] * fpg) if use_comm else None

    def run_comm_frames(n):
        """n frames through rto_comm: batch k's grouped send/recv + assembly overlap batch k+1's render (two buffer sets, two HIP
        streams per rank, no host wait until the flush); --no-pipeline flushes after every batch"""
        full, rest = divmod(n, fpg)
        ptr = comm_frames.data_ptr() if rank == 0 else 0
        stride = H * W * 16
        for _ in range(full):
            comm.submit(comm_arr, ptr, stride, comm_mode)
            if not pipelined:
                comm.flush(COMM_TIMEOUT_MS)
        if rest:
            comm.submit(rto.Context.frame_array([frame] * rest), ptr, stride, comm_mode)
        comm.flush(COMM_TIMEOUT_MS)
        return comm_frames[(rest or fpg) - 1] if rank == 0 else None

    def run_plain_frames(n):
        for _ in range(n):
            render_to(fbuf.data_ptr())
        return fbuf

    run_frames = run_comm_frames if use_comm else run_plain_frames

    # ---- untimed: clock ramp (disclosed), then the W warm-up frames ------------------------------------
    ramp_frames = 0
    primed = 0
    if not use_comm and args.ramp_ms > 0:
        t_end = time.perf_counter() + args.ramp_ms * 1e-3
        while time.perf_counter() < t_end:
            run_plain_frames(20)
            torch.cuda.synchronize()
            ramp_frames += 20
    if use_comm:
        # untimed, disclosed: full batches through the pipe before the warm-up -- RCCL opens its connections at the first
        # send/recv, the buffers take the size of a full batch, every GPU reaches its working clock (all ranks: same count)
        t_end = time.perf_counter() + max(args.ramp_ms, 100.0) * 1e-3
        ptr0 = comm_frames.data_ptr() if rank == 0 else 0
        rounds = 0
        while True:
            for _ in range(4):
                comm.submit(comm_arr, ptr0, H * W * 16, comm_mode)
            comm.flush(COMM_TIMEOUT_MS)
            rounds += 1
            go_on = torch.tensor([1.0 if time.perf_counter() < t_end else 0.0], dtype=torch.float64)
            if dist is not None:
                dist.all_reduce(go_on, op=dist.ReduceOp.MIN)          # every rank submits the same number of collectives
            if float(go_on.item()) == 0.0:
                break
        primed = rounds * 4 * fpg
    run_frames(args.warmup)
    sync_all()

    # ---- the graph of single-frame launches (captured before the timed region; replaying it does all the work) ------
    graph, gframes = None, 0
    if not use_comm and args.graph_frames > 0 and args.steps >= 2:
        gframes = min(args.graph_frames, args.steps)
        ctx.timing_begin(-1)                          # no event pairs: events inside a capture cannot be timed
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                for _ in range(gframes):
                    render_to(fbuf.data_ptr())
            graph.replay()                            # untimed: first replay of a fresh graph
            sync_all()
        except Exception as e:                        # a runtime that cannot capture: plain launches, said so in the line
            print(f"bench: HIP graph capture failed ({type(e).__name__}: {e}); falling back to plain launches", file=sys.stderr, flush=True)
            graph, gframes = None, 0
            torch.cuda.synchronize()
            stream = torch.cuda.Stream()
            torch.cuda.set_stream(stream)
            run_plain_frames(3)
            sync_all()
    if not use_comm and graph is None:
        ctx.timing_begin(-1)                          # the timed launches carry no per-launch events either way (2 us each); see the kernel pass below

    # ---- timed region: exactly K frames ------------------------------------------------------------
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # torch creates a HIP event at its FIRST record (hipEventCreate: 9-17 us, tools/launch_phases.py --once): record both once here
    # so that the timed region pays for K frames and two event records, not for creating the bench's own instruments
    ev_a.record(stream); ev_b.record(stream)
    sync_all()
    t0 = time.perf_counter()
    ev_a.record(stream)
    if graph is not None:
        for _ in range(args.steps // gframes):
            graph.replay()
        img = run_plain_frames(args.steps % gframes) if args.steps % gframes else fbuf      # exactly K frames: the remainder as plain launches
    else:
        img = run_frames(args.steps)
    ev_b.record(stream)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed = max_over_ranks(elapsed)

    # ---- everything below is outside the timed region ---------------------------------------------
    _trace(f'timed region done: {elapsed:.6f} s')
    if img is not None:
        img = img.clone()                                # the legs below reuse the frame buffers
    latency = None
    if use_comm:
        # single-frame latency of the split: one frame per collective, the host waits for the assembled frame before it
        # submits the next (what an interactive viewer sees); the timed region above is the batched, pipelined throughput
        one = rto.Context.frame_array([frame])
        ptr = comm_frames.data_ptr() if rank == 0 else 0
        for _ in range(5):
            comm.submit(one, ptr, H * W * 16, comm_mode); comm.flush(COMM_TIMEOUT_MS)
        sync_all()
        n_lat = 40
        t_l = time.perf_counter()
        for _ in range(n_lat):
            comm.submit(one, ptr, H * W * 16, comm_mode); comm.flush(COMM_TIMEOUT_MS)
        lat = (time.perf_counter() - t_l) / n_lat
        if dist is not None:
            lat = max_over_ranks(lat)
        sent, whole = comm.debug_last_payload()
        latency = {"payload_bytes_per_frame_and_rank": sent * 4, "payload_bytes_whole_rows": whole * 4,
                   "ms_per_frame": round(lat * 1e3, 5), "frames": n_lat,
                   "what": "rto_comm_submit of ONE frame + rto_comm_flush per frame: render part -> grouped send/recv -> assemble, host waits for each frame"}
        # per-rank GPU time of a full batch: timed events around every rank's render and around its gather
        # (+ rank 0's assembly), 5 batches each waited for; the line carries every rank's figure and the slowest
        comm.debug_set_timing(True)
        ptr = comm_frames.data_ptr() if rank == 0 else 0
        rts, gts = [], []
        for _ in range(5):
            comm.submit(comm_arr, ptr, H * W * 16, comm_mode); comm.flush(COMM_TIMEOUT_MS)
            rm, gm = comm.debug_last_timing()
            rts.append(rm / fpg); gts.append(gm / fpg)
        comm.debug_set_timing(False)
        mine = {"rank": rank, "render_ms_per_frame": round(median(rts), 5), "gather_ms_per_frame": round(median(gts), 5)}
        per_rank = [mine]
        if dist is not None:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
        ranks_seen = comm.ranks_seen()
        if dist is not None:
            seen = torch.tensor([float(ranks_seen)], dtype=torch.float64)
            dist.all_reduce(seen, op=dist.ReduceOp.MIN)
            ranks_seen = int(seen.item())
        rank_timing = {"ranks_seen": ranks_seen, "ranks_seen_how": "ncclCommCount of every rank's communicator (minimum over ranks)",
                       "per_rank": per_rank,
                       "render_ms_per_frame_max": max(p["render_ms_per_frame"] for p in per_rank),
                       "gather_ms_per_frame_max": max(p["gather_ms_per_frame"] for p in per_rank),
                       "what": f"GPU ms per frame of a batch of {fpg}, one batch at a time (no overlap between batches): render = this rank's bands (+ pack), "
                               "gather = its grouped send / recv from the moment its parts were rendered (+ rank 0's assembly)"}
        # like for like at N = 1: ONE GPU rendering the same batches whole (fpg frames per launch, the form every rank of the split uses)
        n1 = None
        if rank == 0 and args.rehearse_world <= 1:
            nb = min(4 if triangles else 8, fpg)          # config 5: 4 x 133 MB of frames
            bb = torch.empty((nb, H, W, 4), dtype=torch.float32, device="cuda")
            ba = rto.Context.frame_array([frame] * nb)

            def whole_batch():
                if triangles:
                    ctx.render_triangles_batch_device(ba, bb.data_ptr(), H * W * 16, True, None, False, stream.cuda_stream)
                else:
                    ctx.render_batch_device(ba, bb.data_ptr(), H * W * 16, None, False, stream.cuda_stream)
            for _ in range(4):
                whole_batch()
            torch.cuda.synchronize()
            t_n = time.perf_counter()
            reps = 10
            for _ in range(reps):
                whole_batch()
            torch.cuda.synchronize()
            ms1 = (time.perf_counter() - t_n) / (reps * nb) * 1e3
            n1 = {"frames_per_launch": nb, "ms_per_frame": round(ms1, 5), "Mrays_per_s": round(rays / ms1 / 1e3, 1),
                  "what": "rank 0's GPU alone, whole frames, several frames per kernel launch as every rank of the split renders them (rto_render_[triangles_]batch_device), "
                          "measured behind the timed region while the other ranks wait: the 1-GPU figure `value` should be compared with"}
            del bb
        sync_all()
    _trace(f'latency leg done, rank {rank}')
    if rank == 0:
        kernel_name = KERNEL_NAMES[args.kernel] if info.canonical else "k_trace_generic"
        if triangles:
            kernel_name = ("k_trace_packed_triangles" if args.kernel == "packed_v3" else "k_trace_lean_triangles") if (info.canonical and args.kernel != "generic") else "k_trace_triangles"
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
        fpl_leg = None
        if not use_comm:
            # ---- kernel pass: the same frames as plain launches, a HIP event pair around every traversal kernel on its launch stream
            n_k = max(1, min(args.steps, 200))
            ctx.timing_begin(n_k)
            run_plain_frames(n_k)
            kms = [float(x) for x in ctx.timing_read()]
            ctx.timing_begin(-1)
            assert len(kms) == n_k
            p_avg, k_med, k_min = sum(kms) / len(kms), median(kms), min(kms)
            region_ms = ev_a.elapsed_time(ev_b) / args.steps         # GPU time per frame of the timed region: kernel + the gap to the next launch
            # The kernel's average launch duration, live: HIP events at both ends of the timed region on the launch stream, divided by
            # its K launches (one traversal kernel per frame) -- the gaps between launches and the occasional launch-order rebuild
            # included, so an upper bound; the per-launch event pairs of the separate pass below read 2-3 us MORE than that for a
            # 36 us kernel (two event records around every kernel), rocprofv3's kernel trace slightly less.  Graph replays hold
            # other nodes too: then the pairs are the figure.
            k_avg = region_ms if graph is None else p_avg
            cal = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
            for a, b in cal:
                a.record(stream); b.record(stream)
            torch.cuda.synchronize()
            pair_overhead = median([a.elapsed_time(b) for a, b in cal])
            hbm_alg = rays * bytes_per_ray / (k_avg * 1e-3) / 1e9
            order_key = "centre-out" if args.order != "temporal" else "temporal"
            pmc = pmc_entry(args.config, kernel_name, order_key, 1)
            roofline = {
                "bound": "valu_issue", "achieved": None, "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave-instructions/s", "frac": None,
                "traffic": None, "traffic_source": None,
                "kernel": kernel_name,
                "model": f"{SIMDS} SIMD-32 x {CLOCK_GHZ} GHz / {VALU_CYCLES_PER_WAVE_INST} cycles per wave64 VALU instruction; achieved = SQ_INSTS_VALU per launch (PMC) / kernel_ms_avg",
                "launch_order": order_key, "frames_per_launch": 1,
                "kernel_ms_avg": round(k_avg, 5), "kernel_ms_event_pair_median": round(k_med, 5), "kernel_ms_event_pair_min": round(k_min, 5), "kernel_launches_timed": n_k,
                "kernel_ms_how": (f"avg: HIP events at both ends of the timed region on the launch stream / its {args.steps} launches (gaps between launches included: an upper bound); kernel_ms_event_pair_*: median / min of a SEPARATE pass with an event pair around every launch (another clock: reads 2-3 us more per launch, not comparable with kernel_ms_avg)"
                                  if graph is None else f"HIP event pairs around each of {n_k} plain launches"),
                "kernel_ms_event_pairs": {"avg": round(p_avg, 5), "median": round(k_med, 5), "min": round(k_min, 5), "launches": n_k,
                                          "how": f"a HIP event pair around each of {n_k} launches of the timed workload (plain launches, right after the timed region), on the launch stream"},
                "timed_region_gpu_ms_per_frame": round(region_ms, 5),
                "median_frame": {"ms": round(k_med, 5), "Mrays_per_s": round(rays / k_med / 1e3, 1),
                                 "what": "median of the per-launch kernel times (SURVEY 8d asks for a median; `value` is the mean over the timed region)"},
                "event_pair_overhead_ms": round(pair_overhead, 5),
                "hbm_algorithmic": {
                    "achieved": round(hbm_alg, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "ratio": round(hbm_alg / HBM_PEAK_GBS, 4),
                    "bytes_per_ray": round(bytes_per_ray, 2), "pops_per_ray": round(pops_per_ray, 4),
                    "bytes_per_launch": int(round(rays * bytes_per_ray)),
                    "note": "SURVEY 8d: pops x 60 B reference node + 16 B pixel" + tri_note + ". The packed kernels read 8-byte descriptors of "
                            "internal nodes only (L1/L2 resident), so this ratio exceeds 1: it prices the reference's layout, not this kernel's traffic",
                },
            }
            if pmc is not None:
                insts = float(pmc["SQ_INSTS_VALU"])
                achieved = insts / (k_avg * 1e-3) / 1e9
                roofline["achieved"] = round(achieved, 1)
                roofline["frac"] = round(achieved / VALU_PEAK_GINST, 4)
                roofline["valu_insts_per_launch"] = int(insts)
                # The same duration priced two other ways (docs/LAB_NOTES.md, "Reading `frac`"): at the issue cost measured for
                # this loop's instruction mix (tools/ubench/valu_rate*.hip: 2.3-2.6 cycles for the full-rate third, 4.1-4.4 for
                # the rest), and with round 2's instruction count for the same frame -- frac falls when a change removes
                # instructions faster than time (the occupancy mask), this one does not.
                roofline["issue_weighted"] = round(achieved * MEASURED_CYCLES_PER_WAVE_INST / VALU_CYCLES_PER_WAVE_INST / VALU_PEAK_GINST, 4)
                r02 = R02_VALU_INSTS.get(str(args.config))
                if r02 and order_key == "temporal":
                    roofline["frac_r02_work"] = round(r02 / (k_avg * 1e-3) / 1e9 / VALU_PEAK_GINST, 4)
                if pmc.get("SQ_THREAD_CYCLES_VALU"):
                    roofline["lane_utilisation"] = round(float(pmc["SQ_THREAD_CYCLES_VALU"]) / (64.0 * insts), 3)
                roofline["traffic"] = pmc.get("hbm_bytes_per_launch")
                prof_us = pmc.get("rocprof_avg_us")
                roofline["traffic_source"] = {"file": "profiles/pmc_counters.json", "summary": pmc.get("source"), "commit": pmc.get("commit"),
                                              "device_source_hash": pmc.get("device_source_hash"),
                                              "matches_this_build": pmc.get("device_source_hash") == device_source_hash(),
                                              "rocprof_avg_us": prof_us,
                                              "kernel_time_differs_over_5pct": (abs(k_avg * 1e3 - prof_us) / prof_us > 0.05) if prof_us else None,
                                              "how": "rocprofv3 --pmc, separate passes (FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE, SQ_*), counters only"}
            else:
                roofline["note"] = (f"no PMC entry for config {args.config} / {kernel_name} / {order_key} in profiles/pmc_counters.json: "
                                    "achieved and frac cannot be stated for this combination")

            # ---- secondary: several frames per kernel launch (throughput form), identical cameras, replayed from a graph
            fpl = max(1, min(8, args.frames_per_launch))
            if fpl > 1:
                batch_buf = torch.empty((fpl, H, W, 4), dtype=torch.float32, device="cuda")
                arr = rto.Context.frame_array([frame] * fpl)

                def render_chunk(a=arr):
                    if triangles:
                        ctx.render_triangles_batch_device(a, batch_buf.data_ptr(), H * W * 16, True, None, False, stream.cuda_stream)
                    else:
                        ctx.render_batch_device(a, batch_buf.data_ptr(), H * W * 16, None, False, stream.cuda_stream)
                try:
                    for _ in range(3):
                        render_chunk()
                    torch.cuda.synchronize()
                    launches = max(1, min(12, args.steps // fpl))
                    gb = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gb, stream=stream):
                        for _ in range(launches):
                            render_chunk()
                    gb.replay(); torch.cuda.synchronize()
                    reps = max(1, min(40, args.steps // (launches * fpl)))
                    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ea.record(stream)
                    for _ in range(reps):
                        gb.replay()
                    eb.record(stream)
                    torch.cuda.synchronize()
                    ms = ea.elapsed_time(eb) / (reps * launches * fpl)
                    same = all(bool(torch.equal(batch_buf[i], img)) for i in range(fpl))
                    fpl_leg = {"frames_per_launch": fpl, "ms_per_frame": round(ms, 5), "Mrays_per_s": round(rays / ms / 1e3, 1), "frames": reps * launches * fpl,
                               "frames_equal_the_timed_frame": same,
                               "what": f"{fpl} frames of the SAME camera per kernel launch (rto_render_{'triangles_' if triangles else ''}batch_device), graph replay: "
                                       "optimistic (identical rays share cache lines); `orbit.frames_per_launch` has distinct consecutive cameras"}
                except Exception as e:
                    print(f"bench: frames_per_launch leg skipped ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
        orbit = None
        if not triangles and not use_comm:
            n_orbit = 240 if args.orbit_frames is None else args.orbit_frames
            if n_orbit > 0:
                # a camera that moves: theta advances 0.01 rad per frame, plain stream launches (no graph), the launch-order
                # table is rebuilt every 4th frame from the previous frame's costs
                th0, ph0, r0, tg0 = float(cam.theta), float(cam.phi), float(cam.radius), cam.getTarget()
                oframes = []
                for i in range(n_orbit):
                    c2 = rto.Camera(th0 + 0.01 * i, ph0, r0)
                    c2.setTarget(tg0)
                    oframes.append(rto.make_frame(c2.getView(), c2.getPos(), W / H, 45.0, W, H))
                ctx.timing_begin(-1)                      # no per-launch events: only the pair around the whole sequence below
                # untimed: building the frames above left the GPU idle; the orbit, round and round, until it is back at its working
                # clock (as before the headline's timed region), ending where the timed pass begins
                t_end = time.perf_counter() + args.ramp_ms * 1e-3
                while True:
                    for fr in oframes:
                        render_to(fbuf.data_ptr(), fr)
                    torch.cuda.synchronize()
                    if time.perf_counter() >= t_end:
                        break
                for fr in oframes[:16]:
                    render_to(fbuf.data_ptr(), fr)
                torch.cuda.synchronize()
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t_o = time.perf_counter()
                ea.record(stream)
                for fr in oframes:
                    render_to(fbuf.data_ptr(), fr)
                eb.record(stream)
                torch.cuda.synchronize()
                wall = time.perf_counter() - t_o
                orbit = {"frames": n_orbit, "rad_per_frame": 0.01, "launches": "plain stream launches, one per frame; launch-order table rebuilt when the geometry's tile box "
                         f"changes size, every 4th frame of a camera in motion and at least every {args.order_period}-th frame (k_order_build inside the timed sequence); "
                         f"untimed before it: the same orbit round and round for {args.ramp_ms:.0f} ms (clock ramp, as before the headline's region)",
                         "ms_per_frame": round(wall / n_orbit * 1e3, 5), "gpu_ms_per_frame": round(ea.elapsed_time(eb) / n_orbit, 5),
                         "Mrays_per_s": round(rays * n_orbit / wall / 1e6, 1)}
                if not args.no_verify:
                    # the same sequence once more, untimed: the first, the middle and the last frame against the oracle, bit for bit
                    # (each copied out right behind its launch: same tables, same marks as in the timed pass)
                    from oracle import orc   # the checker, outside every timed region
                    nodes_o = ctx.download_nodes()
                    picks = sorted({0, n_orbit // 2, n_orbit - 1})
                    for i, fr in enumerate(oframes):
                        render_to(fbuf.data_ptr(), fr)
                        if i in picks:
                            torch.cuda.synchronize()
                            c2 = rto.Camera(th0 + 0.01 * i, ph0, r0)
                            c2.setTarget(tg0)
                            want_o, _ = orc.render(nodes_o, grid.min, grid.voxelSize, c2.getView(), c2.getPos(), W / H, 45.0, W, H, nthreads=host_cores())
                            if fbuf.cpu().numpy().tobytes() != np.ascontiguousarray(want_o, np.float32).tobytes():
                                sys.exit(f"bench: frame {i} of the orbit differs from the oracle's -- result void")
                    torch.cuda.synchronize()
                    orbit["frames_verified_against_oracle"] = picks
                fpl = max(1, min(8, args.frames_per_launch))
                if fpl > 1:
                    # the same orbit when the cameras of fpl consecutive frames are known together (a recorded path, an offline
                    # sequence): fpl frames per launch, each with its own camera, plain launches
                    obuf = torch.empty((fpl, H, W, 4), dtype=torch.float32, device="cuda")
                    groups = [rto.Context.frame_array(oframes[i:i + fpl]) for i in range(0, n_orbit - n_orbit % fpl, fpl)]
                    if groups:
                        for ga in groups[:4]:
                            ctx.render_batch_device(ga, obuf.data_ptr(), H * W * 16, None, False, stream.cuda_stream)
                        torch.cuda.synchronize()
                        t_o = time.perf_counter()
                        for ga in groups:
                            ctx.render_batch_device(ga, obuf.data_ptr(), H * W * 16, None, False, stream.cuda_stream)
                        torch.cuda.synchronize()
                        wall = time.perf_counter() - t_o
                        nb = len(groups) * fpl
                        orbit["frames_per_launch"] = {"frames_per_launch": fpl, "frames": nb, "ms_per_frame": round(wall / nb * 1e3, 5),
                                                      "Mrays_per_s": round(rays * nb / wall / 1e6, 1), "what": "distinct consecutive cameras, plain launches"}
        dropin = None
        if not triangles and not use_comm and args.dropin_frames > 0:
            # the reference's call, literally (main.cpp:1127-1131, 1357-1363): the C++ class, one
            # renderSceneComputeWithCulling(camera, W, H, aspect, 45, updateFrustum = true) per frame, nothing waited for in between
            # (the reference never reads its texture back either); own context on the same GPU, octree built from the same grid
            rt = rto.RayTracerBVH()
            rt.ensureComputeInitialized()
            rt.setOctreeFromGrid(grid)
            rt.setFrustumCullingEnabled(True)
            for _ in range(10):
                rt.renderSceneComputeWithCulling(cam, W, H, W / H, 45.0, True)
            got = rt.framebuffer()                         # synchronises
            dropin_ok = got is not None and img is not None and got.tobytes() == img.cpu().numpy().tobytes()
            # untimed: building this context's octree and comparing the frame above left the GPU idle for tens of milliseconds; back to
            # its working clock before anything is timed (the first leg otherwise reads 2-3 us high: tools/dropin_host.py)
            t_end = time.perf_counter() + max(args.ramp_ms, 100.0) * 1e-3
            while time.perf_counter() < t_end:
                for _ in range(20):
                    rt.renderSceneComputeWithCulling(cam, W, H, W / H, 45.0, True)
                rt.finish()
            legs = {}
            for name, update in (("update_every_frame", True), ("update_kernel_forced", True), ("no_update", False)):
                # the second leg switches the update's host-side proof off (rto_debug_set_frustum_shortcut): k_cull_desc runs every frame
                _hip.load().rto_debug_set_frustum_shortcut(rt.context_handle, 0 if name == "update_kernel_forced" else 1)
                for _ in range(4):
                    rt.renderSceneComputeWithCulling(cam, W, H, W / H, 45.0, update)
                ts = []
                for _ in range(3):
                    rt.finish()
                    t_d = time.perf_counter()
                    for _ in range(args.dropin_frames):
                        rt.renderSceneComputeWithCulling(cam, W, H, W / H, 45.0, update)
                    rt.finish()
                    ts.append((time.perf_counter() - t_d) / args.dropin_frames)
                legs[name] = round(median(ts) * 1e3, 5)
                if name == "update_every_frame":
                    import ctypes as _C
                    pv = _C.c_int(0)
                    _hip.load().rto_debug_last_frustum_update_proven(rt.context_handle, _C.byref(pv))
                    proven_on_host = bool(pv.value)
                _trace(f"dropin leg {name}: " + " ".join(f"{t * 1e6:.2f}" for t in ts))
            dropin = {"ms_per_call": legs["update_every_frame"], "ms_per_call_update_kernel_forced": legs["update_kernel_forced"],
                      "ms_per_call_without_update": legs["no_update"], "update_proven_on_host": proven_on_host, "calls": args.dropin_frames,
                      "Mrays_per_s": round(rays / legs["update_every_frame"] / 1e3, 1), "frame_equals_timed_frame": bool(dropin_ok),
                      "what": "C++ RayTracerBVH::renderSceneComputeWithCulling(camera, W, H, aspect, 45, updateFrustum=true) per frame, as main.cpp:1357-1363 calls it "
                              "(plain launches, host never waits inside the loop); median of 3 runs after an untimed clock ramp of the same calls.  ms_per_call: the library as shipped -- at this scene "
                              "rto_update_frustum proves on the host that the reference's 150-unit margin lets no node be culled and launches nothing; "
                              "ms_per_call_update_kernel_forced: the same call with that proof switched off (k_cull_desc tests every node every frame)"}
            del rt
        pcie = None
        if not use_comm and not triangles:
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
        if use_comm:
            parallelism = (f"screen split over {world} GPUs, {args.band_rows}-row bands round-robin, ONE grouped RCCL send/recv per {fpg} frame(s) into rank 0 "
                           f"(4-byte Lambert term, geometry columns only{'; rank 0 gathers only' if (args.rehearse_world or world) >= 4 else ''})"
                           f"{', gather k overlaps render k+1' if pipelined else ''}")
        else:
            parallelism = "1 GPU, one kernel launch per frame, " + (f"replayed from a HIP graph of {gframes} frames" if graph is not None else "plain stream launches")
            if world > 1:
                parallelism = f"{world} independent replicas (every GPU renders whole frames of its own; the screen split could not run: see split_error), " + parallelism[7:]
        replicas = world if not use_comm else 1          # > 1 only after a failed split: every rank rendered its own K frames
        result = {
            "metric": "Mrays/s (primary rays), 1920x1080" if (W, H) == (1920, 1080) else f"Mrays/s (primary rays), {W}x{H}",
            "value": round(replicas * rays * args.steps / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak" if replicas > 1 else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                # the driver's record keeps 120 characters of each string: the config id and the sizes come first
                "workload": f"cfg{args.config}: {short}, {info.num_nodes} nodes, {W}x{H}{', MC triangles+shadow' if triangles else ''}, {camtxt} fov45",
                "workload_detail": f"BASELINE config {args.config}: {what} ({info.num_nodes} nodes), {W}x{H} primary rays"
                                   f"{', Marching-Cubes leaf triangles + 1 shadow ray per hit' if triangles else ''}, {camtxt}, fov 45",
                "parallelism": parallelism,
                "kernel": args.kernel, "tile_mask": (not args.no_tile_mask),
                # the host proved at upload that every node plane gridMin + k * voxelSize is computed without rounding: the lean kernels use
                # the 9-plane / 12-comparison child test (bit-identical to the general one: DESIGN section 3)
                "exact_grid_child_test": bool(ctx.debug_set_exact_grid(True)[1]),
                "clock_ramp": (f"{ramp_frames} untimed frames ({args.ramp_ms:.0f} ms) before the {args.warmup} warm-up frames" if ramp_frames else
                               (f"{primed} untimed frames in full batches before the {args.warmup} warm-up frames" if primed else "none")),
            },
            "hit_rays": stats["hits"], "capped_rays": stats["capped"],
            "verified_against_oracle": verified,
            "device": ctx.device_name,
        }
        if args.rehearse_world > 1:
            result["rehearsal"] = {"as_rank": args.rehearse_rank, "of": args.rehearse_world,
                                   "what": "ONE GPU doing what that rank does per frame (render its bands, grouped send/recv with itself) plus the assembly "
                                           "(in a rehearsal the one GPU is also the assembling rank 0); "
                                           "value / ms_per_step are the per-rank pipeline rate of the split, not a frame rate of this machine"}
        if split_error is not None:
            result["split_error"] = split_error
            result["split_fallback"] = ("the screen split (rto_comm_*: RCCL) could not run; the frames of this line were rendered WHOLE, "
                                        f"{args.steps} per GPU on {world} GPU(s) independently: value = all GPUs' rays / the slowest GPU's time")
        if roofline is not None:
            result["roofline"] = roofline
        if fpl_leg is not None:
            result["frames_per_launch"] = fpl_leg
        if dropin is not None:
            result["dropin_call"] = dropin
        if orbit is not None:
            result["orbit"] = orbit
        if latency is not None:
            result["single_frame_latency"] = latency
            result["batched_throughput"] = {"Mrays_per_s": result["value"], "ms_per_frame": result["ms_per_step"], "frames_per_gather": fpg,
                                            "what": "`value`: batches of frames, gather k overlapping render k+1"}
            result["rank_timing"] = rank_timing
            result["ranks_seen"] = rank_timing["ranks_seen"]
            if n1 is not None:
                result["n1_comparator"] = n1
        if pcie is not None:
            result["pcie_inclusive"] = pcie
        if cpu is not None:
            result["cpu_baseline"] = cpu
            result["speedup_vs_cpu_all_cores"] = round(result["value"] / cpu["value"], 1)
        if world == 1 and not use_comm and args.config == "2" and not args.no_extras and img is not None and args.kernel == "auto":
            t_x = time.perf_counter()
            small = bool(args.dim or args.width or args.height)
            legs = {}
            try:
                extras(args, local_rank, stream, ctx, grid, frame, img, small, legs)
            except Exception as e:            # a leg that cannot run (not one that DISAGREES with the oracle: that is a SystemExit) must not cost the headline
                import traceback
                traceback.print_exc()
                legs["extras_error"] = f"{type(e).__name__}: {e}"
            result.update(legs)
            result["extras_seconds"] = round(time.perf_counter() - t_x, 1)
        _trace('printing the line')
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
