"""Python face of the C++ host layer (ray_tracing_octrees_amd/host/*.cpp, librto_host.so).

Names, argument meaning and error behaviour mirror the reference's host API:
  VoxelGrid, createOctreeFromVoxelGrid, freeOctree, getVoxelSafe   453-skeleton/OctreeVoxel.h
  Camera                                                           453-skeleton/Camera.h
  loadVoxelGrid / saveVoxelGrid / loadVoxelGridPartial             453-skeleton/CacheUtils.h
  RayTracerBVH                                                     453-skeleton/RayTracerBVH.h:28-80
Everything here is a thin ctypes wrapper: the work happens in C++ and, for rendering, in the HIP
library the C++ class loads (librto_hip.so).  No oracle, no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _build
from .hip import NODE_DTYPE, RtoError, _f

_lib = None
_vp = C.c_void_p
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_build.LIB_HOST):
        raise RtoError(-4, f"{_build.LIB_HOST} is not built: run __graft_entry__.build()")
    _build.preload_torch_runtime()     # the C++ class dlopens librto_hip.so later: torch's runtime must come first
    L = C.CDLL(_build.LIB_HOST)
    L.rtoh_grid_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, _vp]
    L.rtoh_grid_new.restype = _vp
    L.rtoh_grid_test_sphere.argtypes = [C.c_int]
    L.rtoh_grid_test_sphere.restype = _vp
    L.rtoh_grid_load.argtypes = [C.c_char_p]
    L.rtoh_grid_load.restype = _vp
    L.rtoh_grid_load_partial.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.rtoh_grid_load_partial.restype = _vp
    L.rtoh_grid_save.argtypes = [_vp, C.c_char_p]
    L.rtoh_grid_free.argtypes = [_vp]
    L.rtoh_grid_free.restype = None
    L.rtoh_grid_info.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.rtoh_grid_info.restype = None
    L.rtoh_grid_data.argtypes = [_vp, _vp]
    L.rtoh_grid_data.restype = None
    L.rtoh_grid_count.argtypes = [_vp]
    L.rtoh_grid_count.restype = C.c_int64
    L.rtoh_grid_recenter.argtypes = [_vp]
    L.rtoh_get_voxel_safe.argtypes = [_vp, C.c_int, C.c_int, C.c_int]
    L.rtoh_octree_build.argtypes = [_vp]
    L.rtoh_octree_build.restype = _vp
    L.rtoh_octree_free.argtypes = [_vp]
    L.rtoh_octree_free.restype = None
    L.rtoh_octree_map_size.restype = C.c_int64
    L.rtoh_octree_flatten.argtypes = [_vp, _vp, C.c_int64]
    L.rtoh_octree_flatten.restype = C.c_int64
    L.rtoh_octree_neighbors.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.rtoh_local_mc.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int64]
    L.rtoh_local_mc.restype = C.c_int64
    L.rtoh_mc_renderer.argtypes = [_vp, _vp, _vp, C.c_int64]
    L.rtoh_mc_renderer.restype = C.c_int64
    L.rtoh_build_leaf_triangles.argtypes = [_vp, _vp, C.c_int64, _vp, C.c_int64, _vp]
    L.rtoh_build_leaf_triangles.restype = C.c_int64
    L.rtoh_camera_new.argtypes = [C.c_float, C.c_float, C.c_float]
    L.rtoh_camera_new.restype = _vp
    L.rtoh_camera_free.argtypes = [_vp]
    L.rtoh_camera_free.restype = None
    L.rtoh_camera_pan.argtypes = [_vp, C.c_float, C.c_float]
    L.rtoh_camera_pan.restype = None
    L.rtoh_camera_increment.argtypes = [_vp, C.c_float, C.c_float, C.c_float]
    L.rtoh_camera_increment.restype = None
    L.rtoh_camera_set_target.argtypes = [_vp, _f32p]
    L.rtoh_camera_set_target.restype = None
    L.rtoh_camera_get.argtypes = [_vp, _f32p, _f32p, _f32p, _f32p]
    L.rtoh_camera_get.restype = None
    L.rtoh_mat4_inverse.argtypes = [_f32p, _f32p]
    L.rtoh_mat4_mul.argtypes = [_f32p, _f32p, _f32p]
    L.rtoh_perspective.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, _f32p]
    L.rtoh_radians.argtypes = [C.c_float]
    L.rtoh_radians.restype = C.c_float
    L.rtoh_frustum_test.argtypes = [_f32p, _f32p, _f32p, C.c_int64, C.c_float, _vp]
    L.rtoh_frustum_test.restype = None
    L.rtoh_rt_new.argtypes = [C.c_int]
    L.rtoh_rt_new.restype = _vp
    L.rtoh_rt_free.argtypes = [_vp]
    L.rtoh_rt_free.restype = None
    L.rtoh_rt_set_devices.argtypes = [_vp, C.c_int, C.c_int]
    L.rtoh_rt_set_devices.restype = None
    L.rtoh_rt_ensure_compute_initialized.argtypes = [_vp]
    L.rtoh_rt_ensure_compute_initialized.restype = None
    L.rtoh_rt_set_octree.argtypes = [_vp, _vp, _vp]
    L.rtoh_rt_set_octree.restype = None
    L.rtoh_rt_set_octree_from_grid.argtypes = [_vp, _vp]
    L.rtoh_rt_set_octree_from_grid.restype = None
    L.rtoh_rt_set_frustum_culling_enabled.argtypes = [_vp, C.c_int]
    L.rtoh_rt_set_frustum_culling_enabled.restype = None
    L.rtoh_rt_render_scene_compute.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_float, C.c_float]
    L.rtoh_rt_render_scene_compute.restype = None
    L.rtoh_rt_render_scene_compute_with_culling.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int]
    L.rtoh_rt_render_scene_compute_with_culling.restype = None
    L.rtoh_rt_build_leaf_triangles.argtypes = [_vp]
    L.rtoh_rt_build_leaf_triangles.restype = None
    L.rtoh_rt_build_leaf_triangles_on_host.argtypes = [_vp]
    L.rtoh_rt_build_leaf_triangles_on_host.restype = None
    L.rtoh_rt_render_scene_triangles.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int]
    L.rtoh_rt_render_scene_triangles.restype = None
    L.rtoh_rt_num_nodes.argtypes = [_vp]
    L.rtoh_rt_num_nodes.restype = C.c_int64
    L.rtoh_rt_framebuffer.argtypes = [_vp, _vp, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.rtoh_rt_finish.argtypes = [_vp]
    L.rtoh_rt_finish.restype = None
    L.rtoh_rt_context.argtypes = [_vp]
    L.rtoh_rt_context.restype = _vp
    L.rtoh_rt_last_error.argtypes = [_vp]
    L.rtoh_rt_last_error.restype = C.c_char_p
    _lib = L
    return L


class VoxelGrid:
    """453-skeleton/OctreeVoxel.h:28-42.  data is uint8 (0 EMPTY, 1 FILLED), shape (dimZ, dimY, dimX)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_array(cls, data: np.ndarray, grid_min, voxel_size) -> "VoxelGrid":
        d = np.ascontiguousarray(data, dtype=np.uint8)
        dz, dy, dx = d.shape
        return cls(load().rtoh_grid_new(dx, dy, dz, _f(grid_min[0]), _f(grid_min[1]), _f(grid_min[2]), _f(voxel_size),
                                        d.ctypes.data))

    @classmethod
    def test_sphere(cls, dim: int) -> "VoxelGrid":
        """The reference app's fallback scene (main.cpp:337-372, 1052-1070) after recenterFilledVoxels."""
        return cls(load().rtoh_grid_test_sphere(dim))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rtoh_grid_free(self._h)
            self._h = None

    def _info(self):
        dims = (C.c_int * 3)()
        mn = (C.c_float * 3)()
        vs = C.c_float()
        load().rtoh_grid_info(self._h, dims, mn, C.byref(vs))
        return tuple(dims), np.array(list(mn), np.float32), np.float32(vs.value)

    @property
    def dims(self):
        return self._info()[0]

    @property
    def min(self) -> np.ndarray:
        return self._info()[1]

    @property
    def voxelSize(self) -> np.float32:
        return self._info()[2]

    @property
    def data(self) -> np.ndarray:
        dx, dy, dz = self.dims
        out = np.empty((dz, dy, dx), np.uint8)
        load().rtoh_grid_data(self._h, out.ctypes.data)
        return out

    def recenter(self) -> bool:
        return bool(load().rtoh_grid_recenter(self._h))


def getVoxelSafe(grid: VoxelGrid, x: int, y: int, z: int) -> int:
    return load().rtoh_get_voxel_safe(grid._h, x, y, z)


def loadVoxelGrid(filename: str) -> VoxelGrid | None:
    h = load().rtoh_grid_load(filename.encode())
    return VoxelGrid(h) if h else None


def loadVoxelGridPartial(filename: str, startLayer: int, numLayers: int) -> VoxelGrid | None:
    h = load().rtoh_grid_load_partial(filename.encode(), startLayer, numLayers)
    return VoxelGrid(h) if h else None


def saveVoxelGrid(filename: str, grid: VoxelGrid) -> bool:
    return bool(load().rtoh_grid_save(grid._h, filename.encode()))


class OctreeNode:
    """Opaque handle of the root of a pointer octree (453-skeleton/OctreeVoxel.h:45-62)."""

    def __init__(self, handle):
        self._h = handle

    def flatten(self) -> np.ndarray:
        """BFS numbering of RayTracerBVH::setOctree (RayTracerBVH.cpp:443-490) as a GPUNodes array."""
        n = load().rtoh_octree_flatten(self._h, None, 0)
        out = np.zeros(n, NODE_DTYPE)
        load().rtoh_octree_flatten(self._h, out.ctypes.data, n)
        return out


def createOctreeFromVoxelGrid(grid: VoxelGrid) -> OctreeNode | None:
    h = load().rtoh_octree_build(grid._h)
    return OctreeNode(h) if h else None


def localMC(grid: VoxelGrid, x0: int, y0: int, z0: int, size: int) -> np.ndarray:
    """453-skeleton/OctreeVoxel.cpp:780-879.  Returns (n, 18) float32: v0,v1,v2 then the three (equal) normals."""
    n = load().rtoh_local_mc(grid._h, x0, y0, z0, size, None, 0)
    out = np.zeros((n, 18), np.float32)
    if n:
        load().rtoh_local_mc(grid._h, x0, y0, z0, size, out.ctypes.data, n)
    return out


def buildLeafTriangles(grid: VoxelGrid, nodes: np.ndarray):
    """Triangle buffer of the config-5 ray path: (tris (n, 12) float32, triOffset (numNodes+1,) int32)."""
    nodes = np.ascontiguousarray(nodes)
    n = load().rtoh_build_leaf_triangles(grid._h, nodes.ctypes.data, len(nodes), None, 0, None)
    tris = np.zeros((n, 12), np.float32)
    off = np.zeros(len(nodes) + 1, np.int32)
    load().rtoh_build_leaf_triangles(grid._h, nodes.ctypes.data, len(nodes), tris.ctypes.data, n, off.ctypes.data)
    return tris, off


class MarchingCubesRenderer:
    """453-skeleton/Renderer.h:18-24: localMC over every leaf of the tree, in child order."""

    def render(self, root: OctreeNode, grid: VoxelGrid) -> np.ndarray:
        n = load().rtoh_mc_renderer(root._h, grid._h, None, 0)
        out = np.zeros((n, 18), np.float32)
        if n:
            load().rtoh_mc_renderer(root._h, grid._h, out.ctypes.data, n)
        return out


def freeOctree(root: OctreeNode | None):
    if root is not None and root._h:
        load().rtoh_octree_free(root._h)
        root._h = None


class Camera:
    """453-skeleton/Camera.h:5-44."""

    def __init__(self, theta: float, phi: float, radius: float):
        self._h = load().rtoh_camera_new(_f(theta), _f(phi), _f(radius))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rtoh_camera_free(self._h)
            self._h = None

    def _get(self):
        view = np.zeros(16, np.float32); pos = np.zeros(3, np.float32)
        tgt = np.zeros(3, np.float32); tpr = np.zeros(3, np.float32)
        load().rtoh_camera_get(self._h, view, pos, tgt, tpr)
        return view, pos, tgt, tpr

    def getView(self) -> np.ndarray:
        return self._get()[0]

    def getPos(self) -> np.ndarray:
        return self._get()[1]

    def getTarget(self) -> np.ndarray:
        return self._get()[2]

    def pan(self, dx: float, dy: float):
        load().rtoh_camera_pan(self._h, _f(dx), _f(dy))

    def incrementTheta(self, dt: float):
        load().rtoh_camera_increment(self._h, _f(dt), 0.0, 0.0)

    def incrementPhi(self, dp: float):
        load().rtoh_camera_increment(self._h, 0.0, _f(dp), 0.0)

    def incrementR(self, dr: float):
        load().rtoh_camera_increment(self._h, 0.0, 0.0, _f(dr))

    def setTarget(self, t):
        load().rtoh_camera_set_target(self._h, np.ascontiguousarray(t, dtype=np.float32))

    @property
    def theta(self):
        return self._get()[3][0]

    @property
    def phi(self):
        return self._get()[3][1]

    @property
    def radius(self):
        return self._get()[3][2]


class RayTracerBVH:
    """The C++ drop-in class (host/RayTracerBVH.h), method for method.

    Usage is the reference's (453-skeleton/main.cpp:1127-1131, 1357-1363):
        rt = RayTracerBVH(); rt.ensureComputeInitialized(); rt.setOctree(root, grid)
        rt.renderSceneComputeWithCulling(camera, W, H, aspect, 45.0, updateFrustum)
        img = rt.framebuffer()          # the one addition: read the frame back
    Render calls return None like the reference's void methods; failures go to stderr from C++.
    """

    def __init__(self, device: int = 0):
        self._h = load().rtoh_rt_new(device)
        self._keep = None

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rtoh_rt_free(self._h)
            self._h = None

    def setDevices(self, n: int, bandRows: int = 16):
        """Before ensureComputeInitialized(): split every frame over n GPUs of this node + ONE RCCL gather (rto_comm_*)."""
        load().rtoh_rt_set_devices(self._h, n, bandRows)

    def ensureComputeInitialized(self):
        load().rtoh_rt_ensure_compute_initialized(self._h)

    def setOctree(self, root: OctreeNode | None, grid: VoxelGrid):
        self._keep = (root, grid)
        load().rtoh_rt_set_octree(self._h, root._h if root is not None else None, grid._h)

    def setOctreeFromGrid(self, grid: VoxelGrid):
        """Addition: build the octree on the GPU straight from the voxel grid (rto_build_octree)."""
        self._keep = (None, grid)
        load().rtoh_rt_set_octree_from_grid(self._h, grid._h)

    def setFrustumCullingEnabled(self, enabled: bool):
        load().rtoh_rt_set_frustum_culling_enabled(self._h, 1 if enabled else 0)

    def renderSceneCompute(self, camera: Camera, width: int, height: int, aspect: float, fovDeg: float):
        load().rtoh_rt_render_scene_compute(self._h, camera._h, width, height, _f(aspect), _f(fovDeg))

    def renderSceneComputeWithCulling(self, camera: Camera, width: int, height: int, aspect: float, fovDeg: float,
                                      updateFrustum: bool):
        load().rtoh_rt_render_scene_compute_with_culling(self._h, camera._h, width, height, _f(aspect), _f(fovDeg),
                                                         1 if updateFrustum else 0)

    # -- additions ---------------------------------------------------------
    def buildLeafTriangles(self):
        """Config 5: per-leaf Marching-Cubes triangles of the grid given to setOctree(), built in HBM."""
        load().rtoh_rt_build_leaf_triangles(self._h)

    def buildLeafTrianglesOnHost(self):
        """The same buffer made by the C++ host builder (localMC per leaf) and uploaded: cross-check of the GPU build."""
        load().rtoh_rt_build_leaf_triangles_on_host(self._h)

    def renderSceneTriangles(self, camera: Camera, width: int, height: int, aspect: float, fovDeg: float, shadow: bool = True):
        load().rtoh_rt_render_scene_triangles(self._h, camera._h, width, height, _f(aspect), _f(fovDeg), 1 if shadow else 0)

    def framebuffer(self) -> np.ndarray | None:
        w, h = C.c_int(), C.c_int()
        if not load().rtoh_rt_framebuffer(self._h, None, 0, C.byref(w), C.byref(h)):
            return None
        out = np.empty((h.value, w.value, 4), np.float32)
        load().rtoh_rt_framebuffer(self._h, out.ctypes.data, out.size, C.byref(w), C.byref(h))
        return out

    def finish(self):
        """Wait for the GPU(s): the counterpart of glFinish for timing loops (renders are asynchronous)."""
        load().rtoh_rt_finish(self._h)

    @property
    def numNodes(self) -> int:
        return load().rtoh_rt_num_nodes(self._h)

    @property
    def lastError(self) -> str:
        return load().rtoh_rt_last_error(self._h).decode()

    @property
    def context_handle(self):
        """The rto_context* the C++ object owns (NULL until ensureComputeInitialized() succeeded)."""
        return load().rtoh_rt_context(self._h)


# ---- small math doors used by tests -------------------------------------------------------------
def mat4_inverse(m):
    out = np.zeros(16, np.float32)
    load().rtoh_mat4_inverse(np.ascontiguousarray(m, dtype=np.float32).reshape(16), out)
    return out


def mat4_mul(a, b):
    out = np.zeros(16, np.float32)
    load().rtoh_mat4_mul(np.ascontiguousarray(a, dtype=np.float32).reshape(16),
                         np.ascontiguousarray(b, dtype=np.float32).reshape(16), out)
    return out


def perspective(fovy_rad, aspect, zn, zf):
    out = np.zeros(16, np.float32)
    load().rtoh_perspective(_f(fovy_rad), _f(aspect), _f(zn), _f(zf), out)
    return out


def radians(deg):
    return np.float32(load().rtoh_radians(_f(deg)))


def frustum_test(vp, mins, maxs, margin):
    mins = np.ascontiguousarray(mins, dtype=np.float32).reshape(-1)
    maxs = np.ascontiguousarray(maxs, dtype=np.float32).reshape(-1)
    out = np.zeros(len(mins) // 3, np.int32)
    load().rtoh_frustum_test(np.ascontiguousarray(vp, dtype=np.float32).reshape(16), mins, maxs, len(out), _f(margin),
                             out.ctypes.data)
    return out
