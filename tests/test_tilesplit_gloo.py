"""Multi-rank path on CPU: world_size 2 and 3 with the gloo backend (no GPU needed).
Also checks the partition arithmetic of the Python module against the C ABI's rto_partition_rows."""
import ctypes as C
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from ray_tracing_octrees_amd import hip, tilesplit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_tile_split_gather_assemble_gloo(world):
    env = dict(os.environ, OMP_NUM_THREADS="1", RTO_NO_TORCH="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_tilesplit_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]


def test_partition_arithmetic_matches_c_abi():
    L = hip.load()
    for H in (1, 7, 8, 37, 1080, 2160):
        f = hip.make_frame(np.eye(4, dtype=np.float32), [0, 0, 0], 1.0, 45.0, 64, H)
        for world in (1, 2, 3, 4, 8):
            for band in (8, 16, 64):
                maps = []
                for p in range(world):
                    rows = L.rto_partition_rows(C.byref(f), C.byref(hip.Partition(world, p, band)))
                    assert rows == tilesplit.partition_rows(H, world, p, band)
                    m = tilesplit.partition_row_map(H, world, p, band)
                    assert len(m) == rows
                    maps.append(m)
                allrows = np.sort(np.concatenate(maps))
                np.testing.assert_array_equal(allrows, np.arange(H))      # a partition: every row exactly once
