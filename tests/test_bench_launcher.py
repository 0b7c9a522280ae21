"""bench.py --gpus N as the driver types it: the parent process touches no GPU, starts N fresh ranks as CHILD processes
(python -m torch.distributed.run, rendezvous on 127.0.0.1), relays rank 0's single JSON line and exits with the
children's status -- loudly non-zero on any failure (here: no GPU in this container)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_parent_builds_the_torchrun_command_line():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "20", "--warmup", "5", "--launcher-dry-run"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 0, p.stderr
    cmd = json.loads(p.stdout.strip().splitlines()[-1])["launcher"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(cmd[cmd.index("--master-port") + 1]) <= 65535
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5", "--launcher-dry-run"]     # the ranks get the caller's arguments


def test_a_rank_under_torchrun_does_not_launch_again():
    """WORLD_SIZE set => this process IS a rank: it must not start another job (it fails later for lack of a GPU)."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-dry-run"], capture_output=True, text=True, env=env, timeout=300)
    assert "launcher" not in p.stdout
    assert p.returncode != 0 and "no GPU visible" in (p.stderr + p.stdout)


def test_failing_ranks_give_a_loud_non_zero_exit():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode != 0
    assert "2-rank job failed" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]        # no result line on failure


def test_ranks_that_hang_are_killed_by_the_watchdog():
    """A rank that dies before its ncclSend leaves the others in a collective for ever: the parent waits --launch-timeout seconds,
    kills exactly the process group it started (fresh children, never an exec), says what they printed and exits 124."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    stub = "import sys, time; print('rank 0: waiting in a collective', flush=True); sys.stderr.write('rank 1 died\\n'); sys.stderr.flush(); time.sleep(600)"
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-timeout", "3", "--launcher-test-command", stub],
                       capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 124, (p.returncode, p.stderr[-500:])
    assert time.time() - t0 < 60
    assert "did not finish within --launch-timeout 3 s" in p.stderr and "rank 1 died" in p.stderr and "waiting in a collective" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]


def test_the_watchdog_lets_a_finished_job_through():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    stub = "print('{\"metric\": \"stub\", \"value\": 1}')"
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-timeout", "30", "--launcher-test-command", stub],
                       capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 0, p.stderr
    assert p.stdout.strip() == '{"metric": "stub", "value": 1}'


def test_bench_camera_jumps_are_the_test_suites_cameras():
    """bench.py's `random_cameras` leg says it uses the generator of test_random_cameras_at_full_size: the same seeds must give the
    same view matrices, positions and fields of view (the product's Camera against the oracle's, both pinned to the reference's)."""
    import importlib.util

    import numpy as np

    import ray_tracing_octrees_amd as rto
    from oracle import orc

    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    grid = rto.VoxelGrid.test_sphere(32)
    og = orc.test_sphere_grid(32)
    dims = np.array(og.dims, np.float32)
    ext = float(dims.max() * og.voxel_size)
    centre = np.asarray(og.min, np.float32) + 0.5 * dims * np.float32(og.voxel_size)
    for seed in range(12):
        cam, fov, kind = bench.seeded_camera(rto, np, seed, grid)
        rng = np.random.default_rng(77000 + seed)                       # tests/test_gpu_parity.py, verbatim
        k2 = ("far", "near", "inside", "past")[seed % 4]
        radius = ext * {"far": rng.uniform(1.5, 5.0), "near": rng.uniform(0.55, 1.0), "inside": rng.uniform(0.05, 0.45), "past": rng.uniform(0.8, 2.0)}[k2]
        oc = orc.Camera(float(rng.uniform(0, 6.28)), float(rng.uniform(-1.4, 1.4)), float(radius))
        aim = rng.uniform(-0.15, 0.15, 3) if k2 != "past" else rng.uniform(0.6, 1.2, 3) * rng.choice([-1.0, 1.0], 3)
        oc.set_target(*[float(x) for x in centre + aim.astype(np.float32) * ext])
        fov2 = float(rng.choice([30.0, 45.0, 70.0]))
        assert (kind, fov) == (k2, fov2)
        assert np.asarray(cam.getView(), np.float32).tobytes() == np.asarray(oc.get_view(), np.float32).tobytes(), seed
        assert np.asarray(cam.getPos(), np.float32).tobytes() == np.asarray(oc.get_pos(), np.float32).tobytes(), seed
