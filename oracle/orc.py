"""ctypes binding of the CPU ORACLE (oracle/liborc.so) and, when built, of the
real-reference library (oracle/_ref/libref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBORC = os.path.join(HERE, "liborc.so")
LIBREF = os.path.join(HERE, "_ref", "libref.so")
LIBREF_VR = os.path.join(HERE, "_ref", "libref_vr.so")
REFERENCE_ROOT = "/root/reference"

NODE_DTYPE = np.dtype(
    [("x", "<i4"), ("y", "<i4"), ("z", "<i4"), ("size", "<i4"),
     ("isLeaf", "<i4"), ("isSolid", "<i4"), ("isUniform", "<i4"), ("child", "<i4", (8,))]
)
assert NODE_DTYPE.itemsize == 60


class _Grid(C.Structure):
    _fields_ = [("dimX", C.c_int32), ("dimY", C.c_int32), ("dimZ", C.c_int32),
                ("minX", C.c_float), ("minY", C.c_float), ("minZ", C.c_float),
                ("voxelSize", C.c_float), ("data", C.c_void_p)]


class _Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("pops", C.c_uint64), ("hits", C.c_uint64),
                ("capped", C.c_uint64), ("internal", C.c_uint64),
                ("max_stack", C.c_uint32), ("pad", C.c_uint32)]


class _Camera(C.Structure):
    _fields_ = [("theta", C.c_float), ("phi", C.c_float), ("radius", C.c_float), ("target", C.c_float * 3)]


@dataclass
class Grid:
    """Host-side VoxelGrid (453-skeleton/OctreeVoxel.h:28-42) as numpy."""
    dims: tuple
    min: np.ndarray          # float32[3]
    voxel_size: np.float32
    data: np.ndarray         # uint8[dimZ, dimY, dimX] (x fastest)

    def c(self) -> _Grid:
        d = np.ascontiguousarray(self.data, dtype=np.uint8)
        self._keep = d
        g = _Grid(self.dims[0], self.dims[1], self.dims[2],
                  float(self.min[0]), float(self.min[1]), float(self.min[2]),
                  float(self.voxel_size), d.ctypes.data)
        return g


def build(ref: bool = False) -> None:
    """(Re)build liborc.so, and libref.so when asked and the reference is present."""
    subprocess.run(["make", "-s", "-C", HERE, "liborc.so"], check=True)
    if ref and os.path.isdir(REFERENCE_ROOT):
        subprocess.run(["make", "-s", "-C", HERE, "ref"], check=True)


_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBORC):
        build()
    L = C.CDLL(LIBORC)
    L.orc_generate_test_sphere.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.orc_recenter_filled_voxels.argtypes = [C.POINTER(_Grid)]
    L.orc_recenter_filled_voxels.restype = C.c_int
    L.orc_load_voxel_grid.argtypes = [C.c_char_p, C.POINTER(_Grid)]
    L.orc_load_voxel_grid.restype = C.c_int
    L.orc_build_flat_octree.argtypes = [C.POINTER(_Grid), C.POINTER(C.c_void_p)]
    L.orc_build_flat_octree.restype = C.c_int64
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_mat4_inverse.argtypes = [_f32p, _f32p]
    L.orc_mat4_mul.argtypes = [_f32p, _f32p, _f32p]
    L.orc_perspective.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, _f32p]
    L.orc_look_at.argtypes = [_f32p, _f32p, _f32p, _f32p]
    L.orc_radians.argtypes = [C.c_float]
    L.orc_radians.restype = C.c_float
    L.orc_camera_init.argtypes = [C.POINTER(_Camera), C.c_float, C.c_float, C.c_float]
    L.orc_camera_pos.argtypes = [C.POINTER(_Camera), _f32p]
    L.orc_camera_view.argtypes = [C.POINTER(_Camera), _f32p]
    L.orc_camera_pan.argtypes = [C.POINTER(_Camera), C.c_float, C.c_float]
    L.orc_frustum_planes.argtypes = [_f32p, _f32p]
    L.orc_frustum_test_aabb.argtypes = [_f32p, _f32p, _f32p, C.c_float]
    L.orc_frustum_test_aabb.restype = C.c_int
    L.orc_cull_compact.argtypes = [C.c_void_p, C.c_int64, _f32p, C.c_float, _f32p, C.c_float, C.c_float,
                                   C.c_void_p, C.c_void_p]
    L.orc_cull_compact.restype = C.c_int64
    L.orc_cull_compact_planes.argtypes = [C.c_void_p, C.c_int64, _f32p, C.c_float, _f32p, C.c_float, C.c_void_p, C.c_void_p]
    L.orc_cull_compact_planes.restype = C.c_int64
    L.orc_render.argtypes = [C.c_void_p, C.c_int64, _f32p, C.c_float, _f32p, _f32p, C.c_float, C.c_float,
                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(_Stats), C.c_int]
    L.orc_render_steps.argtypes = [C.c_void_p, C.c_int64, _f32p, C.c_float, _f32p, _f32p, C.c_float, C.c_float,
                                   C.c_int, C.c_int, C.c_void_p]
    L.orc_render_closest.argtypes = [C.c_void_p, C.c_int64, _f32p, C.c_float, _f32p, _f32p, C.c_float, C.c_float,
                                     C.c_int, C.c_int, C.c_void_p, C.POINTER(_Stats), C.c_int]
    L.orc_octree_ray_skip.argtypes = [C.c_void_p, C.c_int64, _f32p, C.c_float, _f32p, _f32p, C.c_float, C.c_float]
    L.orc_octree_ray_skip.restype = C.c_float
    L.orc_octree_ray_skip_vis.argtypes = [C.c_void_p, C.c_int64, _f32p, C.c_float, _f32p, _f32p, C.c_float, C.c_float, C.c_void_p]
    L.orc_octree_ray_skip_vis.restype = C.c_float
    L.orc_local_mc.argtypes = [C.POINTER(_Grid), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.orc_local_mc.restype = C.c_int64
    L.orc_build_leaf_triangles.argtypes = [C.POINTER(_Grid), C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.orc_build_leaf_triangles.restype = C.c_int64
    L.orc_render_triangles.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, _f32p, C.c_float, _f32p, _f32p, C.c_float,
                                       C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(_Stats), C.c_int]
    L.orc_max_threads.restype = C.c_int
    _lib = L
    return L


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


# ---------------------------------------------------------------- scene
def test_sphere_grid(dim: int) -> Grid:
    """453-skeleton/main.cpp:337-372,1052-1070,1074: shell sphere, min -0.5, voxelSize 1/dim, recentered."""
    data = np.empty((dim, dim, dim), dtype=np.uint8)
    lib().orc_generate_test_sphere(dim, dim, dim, data.ctypes.data)
    g = Grid((dim, dim, dim), np.array([-0.5, -0.5, -0.5], np.float32), np.float32(1.0) / np.float32(dim), data)
    recenter(g)
    return g


def recenter(g: Grid) -> bool:
    cg = g.c()
    ok = lib().orc_recenter_filled_voxels(C.byref(cg))
    g.min = np.array([cg.minX, cg.minY, cg.minZ], np.float32)
    return bool(ok)


def load_voxel_grid(path: str) -> Grid:
    cg = _Grid()
    if not lib().orc_load_voxel_grid(path.encode(), C.byref(cg)):
        raise IOError(f"cannot read voxel grid {path}")
    n = cg.dimX * cg.dimY * cg.dimZ
    data = np.ctypeslib.as_array(C.cast(cg.data, C.POINTER(C.c_uint8)), shape=(n,)).copy()
    lib().orc_free(cg.data)
    return Grid((cg.dimX, cg.dimY, cg.dimZ), np.array([cg.minX, cg.minY, cg.minZ], np.float32),
                np.float32(cg.voxelSize), data.reshape(cg.dimZ, cg.dimY, cg.dimX))


def save_voxel_grid(path: str, g: Grid) -> bool:
    """orc_save_voxel_grid: the sceneCache.bin writer (S/CacheUtils.cpp:5-30)."""
    L = lib()
    L.orc_save_voxel_grid.argtypes = [C.c_char_p, C.POINTER(_Grid)]
    L.orc_save_voxel_grid.restype = C.c_int
    cg = g.c()
    return bool(L.orc_save_voxel_grid(path.encode(), C.byref(cg)))


def build_flat_octree(g: Grid) -> np.ndarray:
    out = C.c_void_p()
    cg = g.c()
    n = lib().orc_build_flat_octree(C.byref(cg), C.byref(out))
    if n == 0:
        return np.zeros(0, NODE_DTYPE)
    arr = np.frombuffer(C.string_at(out.value, n * 60), dtype=NODE_DTYPE).copy()
    lib().orc_free(out)
    return arr


# ---------------------------------------------------------------- camera / glm
class Camera:
    """453-skeleton/Camera.h:5-44 (orbit camera) -- oracle flavour."""

    def __init__(self, theta, phi, radius):
        self._c = _Camera()
        lib().orc_camera_init(C.byref(self._c), theta, phi, radius)

    def pan(self, dx, dy):
        lib().orc_camera_pan(C.byref(self._c), dx, dy)

    def get_view(self) -> np.ndarray:
        out = np.zeros(16, np.float32)
        lib().orc_camera_view(C.byref(self._c), out)
        return out

    def get_pos(self) -> np.ndarray:
        out = np.zeros(3, np.float32)
        lib().orc_camera_pos(C.byref(self._c), out)
        return out

    @property
    def target(self):
        return np.array(list(self._c.target), np.float32)

    def set_target(self, x, y, z):
        """Camera::target is a public member upstream (S/Camera.h); tests aim the orbit camera at arbitrary grids."""
        self._c.target[0], self._c.target[1], self._c.target[2] = float(np.float32(x)), float(np.float32(y)), float(np.float32(z))


def mat4_inverse(m):
    out = np.zeros(16, np.float32)
    lib().orc_mat4_inverse(_f32(m).reshape(16), out)
    return out


def mat4_mul(a, b):
    out = np.zeros(16, np.float32)
    lib().orc_mat4_mul(_f32(a).reshape(16), _f32(b).reshape(16), out)
    return out


def perspective(fovy_rad, aspect, zn, zf):
    out = np.zeros(16, np.float32)
    lib().orc_perspective(fovy_rad, aspect, zn, zf, out)
    return out


def radians(deg):
    return np.float32(lib().orc_radians(deg))


def frustum_planes(vp):
    out = np.zeros(24, np.float32)
    lib().orc_frustum_planes(_f32(vp).reshape(16), out)
    return out


def frustum_test(planes, bmin, bmax, margin):
    return lib().orc_frustum_test_aabb(_f32(planes), _f32(bmin), _f32(bmax), margin)


def cull_compact(nodes, grid_min, voxel_size, view, fov_deg, aspect):
    nodes = np.ascontiguousarray(nodes)
    out = np.zeros(len(nodes), NODE_DTYPE)
    vis = np.zeros(len(nodes), np.uint8)
    n = lib().orc_cull_compact(nodes.ctypes.data, len(nodes), _f32(grid_min), float(voxel_size), _f32(view),
                               float(fov_deg), float(aspect), out.ctypes.data, vis.ctypes.data)
    return out[:n].copy(), vis.astype(bool)


def cull_compact_planes(nodes, grid_min, voxel_size, planes, margin):
    """The reference's visibility loop + compaction (RayTracerBVH.cpp:743-802) for caller-supplied planes and margin."""
    nodes = np.ascontiguousarray(nodes)
    out = np.zeros(len(nodes), NODE_DTYPE)
    vis = np.zeros(len(nodes), np.uint8)
    n = lib().orc_cull_compact_planes(nodes.ctypes.data, len(nodes), _f32(grid_min), float(voxel_size), _f32(planes).reshape(24),
                                      float(margin), out.ctypes.data, vis.ctypes.data)
    return out[:n].copy(), vis.astype(bool)


# ---------------------------------------------------------------- the kernel
def render(nodes, grid_min, voxel_size, view, cam_pos, aspect, fov_deg, W, H, rows=None, nthreads=1,
           out=None):
    """Restatement of 453-skeleton/RayTracerBVH.cpp:226-368. Returns (image[H,W,4] float32, stats dict)."""
    nodes = np.ascontiguousarray(nodes)
    if out is None:
        out = np.zeros((H, W, 4), np.float32)
    y0, y1 = rows if rows is not None else (0, H)
    st = _Stats()
    lib().orc_render(nodes.ctypes.data, len(nodes), _f32(grid_min), float(voxel_size), _f32(view), _f32(cam_pos),
                     float(aspect), float(fov_deg), W, H, y0, y1, out.ctypes.data, C.byref(st), nthreads)
    stats = {k: getattr(st, k) for k in ("rays", "pops", "hits", "capped", "internal", "max_stack")}
    return out, stats


def render_closest(nodes, grid_min, voxel_size, view, cam_pos, aspect, fov_deg, W, H, nthreads=1):
    """The reference's closest-hit traversal (its earlier, block-commented shader, RayTracerBVH.cpp:63-138): (image, stats)."""
    nodes = np.ascontiguousarray(nodes)
    out = np.zeros((H, W, 4), np.float32)
    st = _Stats()
    lib().orc_render_closest(nodes.ctypes.data, len(nodes), _f32(grid_min), float(voxel_size), _f32(view), _f32(cam_pos),
                             float(aspect), float(fov_deg), W, H, out.ctypes.data, C.byref(st), nthreads)
    return out, {k: getattr(st, k) for k in ("rays", "pops", "hits", "capped", "internal", "max_stack")}


def render_steps(nodes, grid_min, voxel_size, view, cam_pos, aspect, fov_deg, W, H):
    nodes = np.ascontiguousarray(nodes)
    steps = np.zeros((H, W), np.int32)
    lib().orc_render_steps(nodes.ctypes.data, len(nodes), _f32(grid_min), float(voxel_size), _f32(view),
                           _f32(cam_pos), float(aspect), float(fov_deg), W, H, steps.ctypes.data)
    return steps


def octree_ray_skip(nodes, grid_min, voxel_size, ro, rd, tmin=0.0, tmax=1e30, visible=None):
    """octreeRaySkip (453-skeleton/VolumeRaycastRenderer.cpp:50-155) for one ray; `visible`: one flag per node (the
    reference's visibility map, :64-67)."""
    nodes = np.ascontiguousarray(nodes)
    if visible is None:
        return lib().orc_octree_ray_skip(nodes.ctypes.data, len(nodes), _f32(grid_min), float(voxel_size),
                                         _f32(ro), _f32(rd), tmin, tmax)
    v = np.ascontiguousarray(visible, dtype=np.uint8)
    assert len(v) == len(nodes)
    return lib().orc_octree_ray_skip_vis(nodes.ctypes.data, len(nodes), _f32(grid_min), float(voxel_size),
                                         _f32(ro), _f32(rd), tmin, tmax, v.ctypes.data)


def octree_ray_skip_many(nodes, grid_min, voxel_size, ro, rds, tmin=0.0, tmax=1e30, visible=None) -> np.ndarray:
    rds = _f32(rds).reshape(-1, 3)
    return np.array([octree_ray_skip(nodes, grid_min, voxel_size, ro, rds[i], tmin, tmax, visible) for i in range(len(rds))], np.float32)


def generate_rays(view, cam_pos, aspect, fov_deg, W, H) -> np.ndarray:
    """generateRay (S/RayTracerBVH.cpp:338-355) for every pixel: (H, W, 3) float32, row 0 = top."""
    L = lib()
    L.orc_generate_rays.argtypes = [_f32p, _f32p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p]
    L.orc_generate_rays.restype = None
    rd = np.zeros((H, W, 3), np.float32)
    L.orc_generate_rays(_f32(view).reshape(16), _f32(cam_pos), aspect, fov_deg, W, H, rd.ctypes.data)
    return rd


def render_skip(nodes, grid_min, voxel_size, view, cam_pos, aspect, fov_deg, W, H, visible=None, nthreads=1):
    """Nearest-hit render mode: per pixel octreeRaySkip(root, ro, generateRay(px, py), 0, 1e30) -> (rgba (H, W, 4), dist (H, W))."""
    L = lib()
    L.orc_render_skip.argtypes = [C.c_void_p, C.c_int64, _f32p, C.c_float, _f32p, _f32p, C.c_float, C.c_float, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_render_skip.restype = None
    nodes = np.ascontiguousarray(nodes)
    v = None if visible is None else np.ascontiguousarray(visible, dtype=np.uint8)
    rgba = np.zeros((H, W, 4), np.float32)
    dist = np.zeros((H, W), np.float32)
    L.orc_render_skip(nodes.ctypes.data, len(nodes), _f32(grid_min), float(voxel_size), _f32(view).reshape(16), _f32(cam_pos),
                      aspect, fov_deg, W, H, None if v is None else v.ctypes.data, rgba.ctypes.data, dist.ctypes.data, nthreads)
    return rgba, dist


def probe_rays(view, eye, aspect) -> np.ndarray:
    """The 7x7 probe directions of drawRaycast (S/VolumeRaycastRenderer.cpp:1602-1630), restated: (49, 3)."""
    L = lib()
    L.orc_probe_rays.argtypes = [_f32p, _f32p, C.c_float, C.c_void_p]
    L.orc_probe_rays.restype = None
    rd = np.zeros((49, 3), np.float32)
    L.orc_probe_rays(_f32(view).reshape(16), _f32(eye), aspect, rd.ctypes.data)
    return rd


def probe_skip_distance(nodes, grid_min, voxel_size, view, eye, aspect, last=0.0, visible=None) -> np.float32:
    """octreeSkipT as drawRaycast computes it (S/VolumeRaycastRenderer.cpp:1602-1663): probes -> 15th percentile x 0.75 -> blend with `last`."""
    L = lib()
    L.orc_probe_skip_distance.argtypes = [C.c_void_p, C.c_int64, _f32p, C.c_float, _f32p, _f32p, C.c_float, C.c_void_p, C.c_float]
    L.orc_probe_skip_distance.restype = C.c_float
    nodes = np.ascontiguousarray(nodes)
    v = None if visible is None else np.ascontiguousarray(visible, dtype=np.uint8)
    return np.float32(L.orc_probe_skip_distance(nodes.ctypes.data, len(nodes), _f32(grid_min), float(voxel_size), _f32(view).reshape(16), _f32(eye),
                                                aspect, None if v is None else v.ctypes.data, float(np.float32(last))))


def local_mc(g: Grid, x0, y0, z0, size) -> np.ndarray:
    out = C.c_void_p()
    cg = g.c()
    n = lib().orc_local_mc(C.byref(cg), x0, y0, z0, size, C.byref(out))
    arr = np.frombuffer(C.string_at(out.value, max(n, 0) * 72), dtype=np.float32).copy().reshape(-1, 18)
    lib().orc_free(out)
    return arr


def build_leaf_triangles(g: Grid, nodes):
    """(tris (n,12) float32, triOffset (numNodes+1,) int32): localMC per leaf in flat-array order."""
    nodes = np.ascontiguousarray(nodes)
    t, o = C.c_void_p(), C.c_void_p()
    cg = g.c()
    n = lib().orc_build_leaf_triangles(C.byref(cg), nodes.ctypes.data, len(nodes), C.byref(t), C.byref(o))
    tris = np.frombuffer(C.string_at(t.value, max(n, 0) * 48), dtype=np.float32).copy().reshape(-1, 12)
    off = np.frombuffer(C.string_at(o.value, (len(nodes) + 1) * 4), dtype=np.int32).copy()
    lib().orc_free(t); lib().orc_free(o)
    return tris, off


def render_triangles(nodes, tris, tri_offset, grid_min, voxel_size, view, cam_pos, aspect, fov_deg, W, H, shadow=True, nthreads=1):
    nodes = np.ascontiguousarray(nodes); tris = np.ascontiguousarray(tris, dtype=np.float32)
    tri_offset = np.ascontiguousarray(tri_offset, dtype=np.int32)
    # the C side indexes these arrays without checks: a wrong-length array was a core dump (round 3, gpurun_out/r3_t33.log)
    if tris.ndim != 2 or tris.shape[1] != 12:
        raise ValueError(f"render_triangles: tris must have shape (n, 12), got {tris.shape}")
    if tri_offset.ndim != 1 or len(tri_offset) != len(nodes) + 1:
        raise ValueError(f"render_triangles: tri_offset needs len(nodes) + 1 = {len(nodes) + 1} entries, got {tri_offset.shape}")
    if len(tri_offset) and (tri_offset[0] != 0 or tri_offset[-1] != len(tris) or (np.diff(tri_offset) < 0).any()):
        raise ValueError("render_triangles: tri_offset must run from 0 to len(tris) without decreasing")
    for name, a, n in (("grid_min", grid_min, 3), ("view", view, 16), ("cam_pos", cam_pos, 3)):
        if np.asarray(a).size != n:
            raise ValueError(f"render_triangles: {name} needs {n} floats, got {np.asarray(a).size}")
    out = np.zeros((H, W, 4), np.float32)
    st = _Stats()
    lib().orc_render_triangles(nodes.ctypes.data, len(nodes), tris.ctypes.data, tri_offset.ctypes.data, _f32(grid_min),
                               float(voxel_size), _f32(view), _f32(cam_pos), float(aspect), float(fov_deg), W, H,
                               1 if shadow else 0, out.ctypes.data, C.byref(st), nthreads)
    return out, {k: getattr(st, k) for k in ("rays", "pops", "hits", "capped")}


def max_threads() -> int:
    return lib().orc_max_threads()


# ---------------------------------------------------------------- real reference (optional)
_ref = None


# ---------------------------------------------------------------- the reference's GLSL text compiled as C++ (optional)
_glsl = {}


def glsl_available() -> bool:
    return all(os.path.exists(os.path.join(os.path.dirname(LIBREF), f"libglsl_{k}.so")) for k in ("first", "closest"))


def glsl_render(kind, nodes, grid_min, voxel_size, view, cam_pos, aspect, fov_deg, W, H):
    """The frame the reference's shader text computes under glm semantics (`make -C oracle glsl`; oracle/glsl_driver.cpp).
    kind: "first" = the live shader (RayTracerBVH.cpp:221-355), "closest" = the earlier block-commented one (:46-166)."""
    if kind not in _glsl:
        L = C.CDLL(os.path.join(os.path.dirname(LIBREF), f"libglsl_{kind}.so"))
        L.glsl_render.argtypes = [C.c_void_p, C.c_int, _f32p, C.c_float, _f32p, _f32p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p]
        L.glsl_render.restype = None
        _glsl[kind] = L
    nodes = np.ascontiguousarray(nodes)
    out = np.zeros((H, W, 4), np.float32)
    _glsl[kind].glsl_render(nodes.ctypes.data, len(nodes), _f32(grid_min), float(voxel_size), _f32(view), _f32(cam_pos),
                            float(aspect), float(fov_deg), W, H, out.ctypes.data)
    return out


def ref_available() -> bool:
    return os.path.exists(LIBREF)


def ref():
    """The compiled reference (oracle/_ref/libref.so); only exists where `make -C oracle ref` ran."""
    global _ref
    if _ref is not None:
        return _ref
    R = C.CDLL(LIBREF)
    R.ref_build_flat_octree.argtypes = [C.POINTER(_Grid), C.POINTER(C.c_void_p)]
    R.ref_build_flat_octree.restype = C.c_int64
    R.ref_free.argtypes = [C.c_void_p]
    R.ref_get_voxel_safe.argtypes = [C.POINTER(_Grid), C.c_int, C.c_int, C.c_int]
    R.ref_get_voxel_safe.restype = C.c_int
    R.ref_camera.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, _f32p, _f32p, _f32p]
    R.ref_glm_inverse.argtypes = [_f32p, _f32p]
    R.ref_glm_mul.argtypes = [_f32p, _f32p, _f32p]
    R.ref_glm_perspective.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, _f32p]
    R.ref_glm_radians.argtypes = [C.c_float]
    R.ref_glm_radians.restype = C.c_float
    R.ref_glm_normalize3.argtypes = [_f32p, _f32p]
    R.ref_glm_normalize4.argtypes = [_f32p, _f32p]
    R.ref_glm_mat_vec.argtypes = [_f32p, _f32p, _f32p]
    R.ref_frustum_test.argtypes = [_f32p, _f32p, _f32p, C.c_int64, C.c_float, C.c_void_p]
    R.ref_load_voxel_grid.argtypes = [C.c_char_p, C.POINTER(_Grid)]
    R.ref_load_voxel_grid.restype = C.c_int
    R.ref_local_mc.argtypes = [C.POINTER(_Grid), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    R.ref_local_mc.restype = C.c_int64
    _ref = R
    return R


def ref_build_flat_octree(g: Grid) -> np.ndarray:
    out = C.c_void_p()
    cg = g.c()
    n = ref().ref_build_flat_octree(C.byref(cg), C.byref(out))
    if n == 0:
        return np.zeros(0, NODE_DTYPE)
    arr = np.frombuffer(C.string_at(out.value, n * 60), dtype=NODE_DTYPE).copy()
    ref().ref_free(out)
    return arr


def ref_camera(theta, phi, radius, pan=None):
    view = np.zeros(16, np.float32)
    pos = np.zeros(3, np.float32)
    tgt = np.zeros(3, np.float32)
    ref().ref_camera(theta, phi, radius, 1 if pan else 0, pan[0] if pan else 0.0, pan[1] if pan else 0.0, view, pos, tgt)
    return view, pos, tgt


def ref_load_voxel_grid(path: str) -> Grid:
    cg = _Grid()
    if not ref().ref_load_voxel_grid(path.encode(), C.byref(cg)):
        raise IOError(path)
    n = cg.dimX * cg.dimY * cg.dimZ
    data = np.ctypeslib.as_array(C.cast(cg.data, C.POINTER(C.c_uint8)), shape=(n,)).copy()
    ref().ref_free(cg.data)
    return Grid((cg.dimX, cg.dimY, cg.dimZ), np.array([cg.minX, cg.minY, cg.minZ], np.float32),
                np.float32(cg.voxelSize), data.reshape(cg.dimZ, cg.dimY, cg.dimX))


def ref_frustum_test(vp, mins, maxs, margin):
    mins = _f32(mins).reshape(-1, 3)
    maxs = _f32(maxs).reshape(-1, 3)
    out = np.zeros(len(mins), np.int32)
    ref().ref_frustum_test(_f32(vp).reshape(16), mins.reshape(-1), maxs.reshape(-1), len(mins), margin, out.ctypes.data)
    return out


def ref_local_mc(g: Grid, x0, y0, z0, size) -> np.ndarray:
    out = C.c_void_p()
    cg = g.c()
    n = ref().ref_local_mc(C.byref(cg), x0, y0, z0, size, C.byref(out))
    arr = np.frombuffer(C.string_at(out.value, max(n, 0) * 72), dtype=np.float32).copy().reshape(-1, 18)
    ref().ref_free(out)
    return arr


# ---- the reference's static octreeRaySkip, through oracle/ref_shim_vr.cpp (make -C oracle refvr)
_refvr = None


def refvr_available() -> bool:
    return os.path.exists(LIBREF_VR)


def refvr():
    global _refvr
    if _refvr is None:
        R = C.CDLL(LIBREF_VR)
        R.refvr_octree_ray_skip.argtypes = [C.POINTER(_Grid), _f32p, _f32p, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_int64, _f32p]
        R.refvr_octree_ray_skip.restype = C.c_int64
        R.refvr_probe_rays.argtypes = [_f32p, _f32p, C.c_float, _f32p]
        _refvr = R
    return _refvr


def ref_octree_ray_skip(g: Grid, ro, rds, tmin=0.0, tmax=1e30, visible=None) -> np.ndarray:
    """The reference's own compiled octreeRaySkip on its own octree of `g`, one call per direction."""
    rds = _f32(rds).reshape(-1, 3)
    out = np.zeros(len(rds), np.float32)
    cg = g.c()
    v = None if visible is None else np.ascontiguousarray(visible, dtype=np.uint8)
    n = refvr().refvr_octree_ray_skip(C.byref(cg), _f32(ro), rds.reshape(-1), len(rds), tmin, tmax,
                                      None if v is None else v.ctypes.data, 0 if v is None else len(v), out)
    if n == 0:
        raise RuntimeError("refvr_octree_ray_skip failed (node count mismatch?)")
    return out


def ref_probe_rays(view, eye, aspect) -> np.ndarray:
    rd = np.zeros(49 * 3, np.float32)
    refvr().refvr_probe_rays(_f32(view).reshape(16), _f32(eye), aspect, rd)
    return rd.reshape(49, 3)
