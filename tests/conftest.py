"""Shared fixtures.  GPU tests are marked `gpu`; everything else runs on a CPU-only box.

The oracle (oracle/) is imported here and in the tests only -- it is the checker.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

SPHERE_CAM = (0.5, 0.7, 1.8)          # Camera(theta 0.5, phi 0.7, r 1.8), fov 45


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_builds():
    """Build the in-tree native pieces once (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as entry

    entry.build()


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as _orc

    return _orc


@pytest.fixture(scope="session")
def golden_meta():
    with open(os.path.join(GOLDEN, "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


class Scene:
    """Oracle-side scene: grid + flat octree (GPUNodes array)."""

    def __init__(self, grid, nodes):
        self.grid = grid
        self.nodes = nodes
        self.min = grid.min
        self.voxel = grid.voxel_size


@pytest.fixture(scope="session")
def scenes(orc, golden):
    cache = {}

    def get(name):
        if name in cache:
            return cache[name]
        if name.startswith("sphere"):
            g = orc.test_sphere_grid(int(name[6:]))
        elif name == "calgary":
            z = golden("ref_scene_cache.npz")
            dims = tuple(int(x) for x in z["dims"])
            data = np.unpackbits(z["packed"])[: dims[0] * dims[1] * dims[2]].reshape(dims[2], dims[1], dims[0])
            g = orc.Grid(dims, z["min"].astype(np.float32), np.float32(z["voxel"]), data)
        elif name == "odd":
            z = golden("ref_octrees_small.npz")
            d = z["odd_grid"]
            g = orc.Grid((d.shape[2], d.shape[1], d.shape[0]), z["odd_min"], np.float32(z["odd_voxel"]), d)
        else:
            raise KeyError(name)
        cache[name] = Scene(g, orc.build_flat_octree(g))
        return cache[name]

    return get


def make_camera(orc, theta, phi, radius, pan=None):
    cam = orc.Camera(theta, phi, radius)
    if pan:
        cam.pan(*pan)
    return cam.get_view(), cam.get_pos()


@pytest.fixture(scope="session")
def camera(orc):
    def get(name="sphere"):
        if name == "sphere":
            return make_camera(orc, *SPHERE_CAM)
        if name == "calgary_default":
            return make_camera(orc, float(np.float32(np.pi / 2)), 0.0, 500.0, (0.0, 100.0))
        if name == "calgary_oblique":
            return make_camera(orc, 0.6, 0.5, 3500.0)
        raise KeyError(name)
    return get


@pytest.fixture(scope="session")
def ctx():
    """One HIP context for the whole GPU session (single process, single device)."""
    import ray_tracing_octrees_amd as rto

    c = rto.Context(0)
    yield c
    c.close()


def bits_equal(a: np.ndarray, b: np.ndarray) -> bool:
    return a.shape == b.shape and bool((a.view(np.uint32) == b.view(np.uint32)).all())


def assert_bit_exact(got: np.ndarray, want: np.ndarray, what: str = ""):
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    neq = got.view(np.uint32) != want.view(np.uint32)
    if neq.any():
        bad = int(neq.reshape(neq.shape[0], neq.shape[1], -1).any(axis=-1).sum()) if neq.ndim == 3 else int(neq.sum())
        raise AssertionError(f"{what}: {bad} elements differ bitwise, max abs diff "
                             f"{np.nanmax(np.abs(got.astype(np.float64) - want.astype(np.float64))):g}")


def partition_row_map(height: int, num_parts: int, part: int, band_rows: int) -> np.ndarray:
    """Test helper: global row of every local row of part `part` of an rto_partition (bands of band_rows rows, band b
    owned by part b % num_parts, a part's bands back to back) -- the definition in include/rto_hip.h, stated independently
    of the library to check partial renders against rows of the oracle's frame."""
    if num_parts <= 1:
        return np.arange(height)
    rows = [y for y in range(height) if (y // band_rows) % num_parts == part]
    return np.asarray(rows, dtype=np.int64)
