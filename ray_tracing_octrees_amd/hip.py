"""ctypes binding of the C ABI in include/rto_hip.h (librto_hip.so).

There is no CPU fallback: if the HIP library is missing or no gfx950 device is
usable, `load()` / `Context()` raise `RtoError`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _build

RTO_OK = 0
RTO_E_INVALID, RTO_E_NO_OCTREE, RTO_E_HIP, RTO_E_NO_DEVICE, RTO_E_UNSUPPORTED, RTO_E_TIMEOUT = -1, -2, -3, -4, -5, -6
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_PACKED, KERNEL_PACKED_V1, KERNEL_PACKED_PERSISTENT, KERNEL_PACKED_V3 = 0, 1, 2, 3, 4, 5

# struct GPUNodes (453-skeleton/RayTracerBVH.h:21-26)
NODE_DTYPE = np.dtype(
    [("x", "<i4"), ("y", "<i4"), ("z", "<i4"), ("size", "<i4"),
     ("isLeaf", "<i4"), ("isSolid", "<i4"), ("isUniform", "<i4"), ("child", "<i4", (8,))]
)

# every symbol include/rto_hip.h declares
SYMBOLS = (
    "rto_create", "rto_destroy", "rto_last_error", "rto_device_name",
    "rto_upload_octree", "rto_build_octree", "rto_download_nodes", "rto_debug_set_build_path", "rto_last_build_ms", "rto_octree_info_get", "rto_set_kernel", "rto_set_launch_order", "rto_forget_stream",
    "rto_update_frustum", "rto_debug_update_frustum_planes", "rto_debug_set_frustum_shortcut", "rto_debug_last_frustum_update_proven", "rto_download_visible_nodes",
    "rto_render_device", "rto_render_host", "rto_partition_rows", "rto_assemble_device",
    "rto_render_shade_device", "rto_assemble_shade_device", "rto_assemble_batch_device", "rto_render_batch_device", "rto_assemble_batch_all_device", "rto_render_resident", "rto_resident_frame", "rto_download_resident",
    "rto_upload_leaf_triangles", "rto_build_leaf_triangles", "rto_download_leaf_triangles", "rto_render_triangles_device", "rto_render_triangles_host", "rto_render_triangles_shade_device",
    "rto_octree_ray_skip", "rto_frame_stats", "rto_render_steps_host", "rto_debug_timeline", "rto_debug_tile_cost", "rto_debug_set_tile_order", "rto_debug_sort_violations", "rto_last_kernel_ms", "rto_timing_begin", "rto_timing_read", "rto_stream", "rto_synchronize",
    "rto_comm_unique_id", "rto_comm_create", "rto_comm_create_all", "rto_comm_destroy", "rto_comm_last_error", "rto_comm_submit",
    "rto_comm_submit_all", "rto_comm_render_resident_all", "rto_comm_flush", "rto_comm_flush_timeout", "rto_comm_is_dead", "rto_comm_ranks_seen", "rto_comm_debug_abort", "rto_comm_stream", "rto_comm_debug_rehearse", "rto_comm_debug_last_payload", "rto_comm_debug_set_timing", "rto_comm_debug_last_timing", "rto_comm_debug_set_rehearsal_clear", "rto_debug_fault_alloc", "rto_render_triangles_batch_device",
    "rto_debug_set_tile_mask", "rto_debug_tile_mask_info", "rto_render_closest_device", "rto_render_closest_host", "rto_render_skip_device", "rto_render_skip_host", "rto_probe_skip_device", "rto_probe_skip_host",
    "rto_scene_bounds_get", "rto_scene_bounds_of_nodes", "rto_split_plan_make", "rto_split_part_of_rank", "rto_split_rows_of_part", "rto_split_row_source",
)
SPLIT_MAX_FRAMES = 32
COMM_ID_BYTES = 128
RESIDENT_OCTREE, RESIDENT_TRIANGLES, RESIDENT_TRIANGLES_SHADOW = 0, 1, 2


class RtoError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rto error {code}: {msg}")
        self.code = code


class Frame(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("cam_pos", C.c_float * 3), ("aspect", C.c_float),
                ("fov_deg", C.c_float), ("width", C.c_int32), ("height", C.c_int32)]


class Partition(C.Structure):
    _fields_ = [("num_parts", C.c_int32), ("part", C.c_int32), ("band_rows", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("pops", C.c_uint64), ("hits", C.c_uint64), ("capped", C.c_uint64)]


class OctreeInfo(C.Structure):
    _fields_ = [("num_nodes", C.c_int64), ("num_internal", C.c_int64), ("root_size", C.c_int32),
                ("depth", C.c_int32), ("canonical", C.c_int32), ("culling_active", C.c_int32),
                ("visible_nodes", C.c_int64)]


class SceneBounds(C.Structure):
    """rto_scene_bounds: what the screen rectangles and the split plan need to know of a scene."""
    _fields_ = [("grid_min", C.c_float * 3), ("voxel_size", C.c_float), ("root_size", C.c_int32),
                ("solid_lo", C.c_int32 * 3), ("solid_hi", C.c_int32 * 3)]


class SplitPlan(C.Structure):
    """rto_split_plan (include/rto_hip.h): everything the ranks of a screen split must agree on."""
    _fields_ = [("world", C.c_int32), ("band_rows", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("n_frames", C.c_int32),
                ("render_parts", C.c_int32), ("first_render_rank", C.c_int32), ("rows_part0", C.c_int32), ("cropped", C.c_int32),
                ("win_x0", C.c_int32 * SPLIT_MAX_FRAMES), ("win_w", C.c_int32 * SPLIT_MAX_FRAMES), ("win_off", C.c_int64 * SPLIT_MAX_FRAMES),
                ("frame_floats", C.c_int64), ("full_floats", C.c_int64), ("pack_floats", C.c_int64)]


_lib = None


def _f(x) -> float:
    """A Python float that holds exactly the binary32 value of x."""
    return float(np.float32(x))


def lib_path() -> str:
    # RTO_HIP_LIB: developer aid for A/B builds of the kernels (another librto_hip.so with the same ABI)
    return os.environ.get("RTO_HIP_LIB") or _build.LIB_HIP


def load():
    """dlopen librto_hip.so and declare prototypes.  Raises RtoError when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RtoError(RTO_E_NO_DEVICE, f"{path} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                        "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    _build.preload_torch_runtime()     # one HIP runtime per process (see _build.preload_torch_runtime)
    L = C.CDLL(path)
    vp = C.c_void_p
    L.rto_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.rto_destroy.argtypes = [vp]
    L.rto_destroy.restype = None
    L.rto_last_error.argtypes = [vp]
    L.rto_last_error.restype = C.c_char_p
    L.rto_device_name.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.rto_upload_octree.argtypes = [vp, vp, C.c_int64, C.POINTER(C.c_float), C.c_float]
    L.rto_octree_info_get.argtypes = [vp, C.POINTER(OctreeInfo)]
    L.rto_build_octree.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_float]
    L.rto_download_nodes.argtypes = [vp, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.rto_debug_set_build_path.argtypes = [vp, C.c_int]
    L.rto_last_build_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.rto_set_kernel.argtypes = [vp, C.c_int]
    L.rto_set_launch_order.argtypes = [vp, C.c_int, C.c_int]
    L.rto_forget_stream.argtypes = [vp, vp]
    L.rto_debug_update_frustum_planes.argtypes = [vp, C.POINTER(C.c_float), C.c_float]
    L.rto_debug_set_frustum_shortcut.argtypes = [vp, C.c_int]
    L.rto_debug_last_frustum_update_proven.argtypes = [vp, C.POINTER(C.c_int)]
    L.rto_debug_sort_violations.argtypes = [vp, C.POINTER(C.c_int)]
    L.rto_debug_set_tile_mask.argtypes = [vp, C.c_int]
    L.rto_debug_tile_mask_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.rto_update_frustum.argtypes = [vp, C.POINTER(C.c_float), C.c_float, C.c_float, C.c_int]
    L.rto_download_visible_nodes.argtypes = [vp, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.rto_render_device.argtypes = [vp, C.POINTER(Frame), C.POINTER(Partition), vp, vp]
    L.rto_render_host.argtypes = [vp, C.POINTER(Frame), vp]
    L.rto_partition_rows.argtypes = [C.POINTER(Frame), C.POINTER(Partition)]
    L.rto_assemble_device.argtypes = [vp, C.POINTER(Frame), C.POINTER(Partition), vp, vp, vp]
    L.rto_render_resident.argtypes = [vp, C.POINTER(Frame), C.c_int]
    L.rto_resident_frame.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.rto_download_resident.argtypes = [vp, vp]
    L.rto_render_shade_device.argtypes = [vp, C.POINTER(Frame), C.POINTER(Partition), vp, vp]
    L.rto_assemble_shade_device.argtypes = [vp, C.POINTER(Frame), C.POINTER(Partition), vp, vp, vp]
    L.rto_assemble_batch_device.argtypes = [vp, C.POINTER(Frame), C.POINTER(Partition), vp, C.c_int, C.c_int, C.c_int, vp, vp]
    L.rto_render_batch_device.argtypes = [vp, C.POINTER(Frame), C.c_int, C.POINTER(Partition), C.c_int, vp, C.c_size_t, vp]
    L.rto_assemble_batch_all_device.argtypes = [vp, C.POINTER(Frame), C.c_int, C.POINTER(Partition), vp, C.c_int, vp, C.c_size_t, vp]
    L.rto_render_triangles_shade_device.argtypes = [vp, C.POINTER(Frame), C.POINTER(Partition), C.c_int, vp, vp]
    L.rto_frame_stats.argtypes = [vp, C.POINTER(Frame), C.POINTER(Stats)]
    L.rto_upload_leaf_triangles.argtypes = [vp, vp, C.c_int64, vp]
    L.rto_render_triangles_device.argtypes = [vp, C.POINTER(Frame), C.POINTER(Partition), C.c_int, vp, vp]
    L.rto_build_leaf_triangles.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    L.rto_download_leaf_triangles.argtypes = [vp, vp, C.c_int64, vp, C.POINTER(C.c_int64)]
    L.rto_render_triangles_host.argtypes = [vp, C.POINTER(Frame), C.c_int, vp, C.POINTER(Stats)]
    L.rto_octree_ray_skip.argtypes = [vp, C.POINTER(C.c_float), vp, C.c_int64, C.c_float, C.c_float, C.c_int, vp]
    L.rto_render_steps_host.argtypes = [vp, C.POINTER(Frame), vp]
    L.rto_render_closest_device.argtypes = [vp, C.POINTER(Frame), C.POINTER(Partition), vp, vp]
    L.rto_render_closest_host.argtypes = [vp, C.POINTER(Frame), vp, C.POINTER(Stats)]
    L.rto_render_skip_device.argtypes = [vp, C.POINTER(Frame), C.POINTER(Partition), C.c_int, vp, vp, vp]
    L.rto_render_skip_host.argtypes = [vp, C.POINTER(Frame), C.c_int, vp, vp]
    L.rto_probe_skip_device.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_int, vp, vp]
    L.rto_probe_skip_host.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_int, C.POINTER(C.c_float), vp]
    L.rto_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.rto_debug_tile_cost.argtypes = [vp, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.rto_debug_set_tile_order.argtypes = [vp, vp, C.c_int64]
    L.rto_debug_timeline.argtypes = [vp, C.POINTER(Frame), vp, C.c_int64, C.POINTER(C.c_int64)]
    L.rto_synchronize.argtypes = [vp]
    L.rto_timing_begin.argtypes = [vp, C.c_int]
    L.rto_timing_read.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.rto_stream.argtypes = [vp]
    L.rto_stream.restype = vp
    L.rto_comm_unique_id.argtypes = [vp]
    L.rto_comm_create.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.POINTER(vp)]
    L.rto_comm_create_all.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(vp)]
    L.rto_comm_destroy.argtypes = [vp]
    L.rto_comm_destroy.restype = None
    L.rto_comm_last_error.argtypes = [vp]
    L.rto_comm_last_error.restype = C.c_char_p
    L.rto_comm_submit.argtypes = [vp, C.POINTER(Frame), C.c_int, C.c_int, vp, C.c_size_t]
    L.rto_comm_submit_all.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(Frame), C.c_int, C.c_int, vp, C.c_size_t]
    L.rto_comm_render_resident_all.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(Frame), C.c_int]
    L.rto_comm_flush.argtypes = [vp]
    L.rto_comm_flush_timeout.argtypes = [vp, C.c_int]
    L.rto_comm_is_dead.argtypes = [vp]
    L.rto_comm_ranks_seen.argtypes = [vp, C.POINTER(C.c_int)]
    L.rto_comm_debug_abort.argtypes = [vp]
    L.rto_comm_debug_rehearse.argtypes = [vp, C.c_int, C.c_int]
    L.rto_comm_debug_set_rehearsal_clear.argtypes = [vp, C.c_int]
    L.rto_comm_debug_set_timing.argtypes = [vp, C.c_int]
    L.rto_comm_debug_last_timing.argtypes = [vp, C.POINTER(C.c_float)]
    L.rto_debug_fault_alloc.argtypes = [C.c_long]
    L.rto_comm_debug_last_payload.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.rto_render_triangles_batch_device.argtypes = [vp, C.POINTER(Frame), C.c_int, C.POINTER(Partition), C.c_int, C.c_int, vp, C.c_size_t, vp]
    L.rto_comm_stream.argtypes = [vp]
    L.rto_comm_stream.restype = vp
    L.rto_scene_bounds_get.argtypes = [vp, C.POINTER(SceneBounds)]
    L.rto_scene_bounds_of_nodes.argtypes = [vp, C.c_int64, C.POINTER(C.c_float), C.c_float, C.POINTER(SceneBounds)]
    L.rto_split_plan_make.argtypes = [C.POINTER(SceneBounds), C.POINTER(Frame), C.c_int, C.c_int, C.c_int, C.POINTER(SplitPlan)]
    L.rto_split_part_of_rank.argtypes = [C.POINTER(SplitPlan), C.c_int]
    L.rto_split_rows_of_part.argtypes = [C.POINTER(SplitPlan), C.c_int]
    L.rto_split_row_source.argtypes = [C.POINTER(SplitPlan), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    _lib = L
    return L


def make_frame(view, cam_pos, aspect, fov_deg, width, height) -> Frame:
    f = Frame()
    v = np.asarray(view, dtype=np.float32).reshape(16)
    p = np.asarray(cam_pos, dtype=np.float32).reshape(3)
    v = np.ascontiguousarray(v)
    p = np.ascontiguousarray(p)
    C.memmove(f.view, v.ctypes.data, 64)       # bit copies: no double rounding
    C.memmove(f.cam_pos, p.ctypes.data, 12)
    f.aspect = _f(aspect)
    f.fov_deg = _f(fov_deg)
    f.width, f.height = int(width), int(height)
    return f


class Context:
    """One rto_context == one GPU."""

    def __init__(self, device: int = 0):
        self._L = load()
        h = C.c_void_p()
        rc = self._L.rto_create(device, C.byref(h))
        if rc != RTO_OK:
            raise RtoError(rc, self._L.rto_last_error(None).decode())
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.rto_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != RTO_OK:
            raise RtoError(rc, self._L.rto_last_error(self._h).decode())

    @property
    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        self._check(self._L.rto_device_name(self._h, buf, 256))
        return buf.value.decode()

    # -- octree ------------------------------------------------------------
    def upload_octree(self, nodes: np.ndarray, grid_min, voxel_size):
        nodes = np.ascontiguousarray(nodes)
        if nodes.dtype.itemsize != 60:
            raise RtoError(RTO_E_INVALID, "nodes must be an array of 60-byte GPUNodes records")
        gm = (C.c_float * 3)(*[_f(x) for x in grid_min])
        self._check(self._L.rto_upload_octree(self._h, nodes.ctypes.data, len(nodes), gm, _f(voxel_size)))

    def build_octree(self, voxels: np.ndarray, grid_min, voxel_size):
        """N4: createOctreeFromVoxelGrid + setOctree on the GPU. voxels: uint8 (dimZ, dimY, dimX), 0 EMPTY / 1 FILLED."""
        v = np.ascontiguousarray(voxels, dtype=np.uint8)
        dz, dy, dx = v.shape
        gm = (C.c_float * 3)(*[_f(x) for x in grid_min])
        self._check(self._L.rto_build_octree(self._h, v.ctypes.data, dx, dy, dz, gm, _f(voxel_size)))

    def debug_set_build_path(self, level_by_level: bool):
        """Force rto_build_octree's level-by-level form (True) or restore the automatic choice (False)."""
        self._check(self._L.rto_debug_set_build_path(self._h, 1 if level_by_level else 0))

    def download_nodes(self) -> np.ndarray:
        cnt = C.c_int64()
        self._check(self._L.rto_download_nodes(self._h, None, 0, C.byref(cnt)))
        out = np.zeros(cnt.value, NODE_DTYPE)
        self._check(self._L.rto_download_nodes(self._h, out.ctypes.data, cnt.value, C.byref(cnt)))
        return out

    def last_build_ms(self):
        k, u = C.c_float(), C.c_float()
        self._check(self._L.rto_last_build_ms(self._h, C.byref(k), C.byref(u)))
        return k.value, u.value

    def info(self) -> OctreeInfo:
        o = OctreeInfo()
        self._check(self._L.rto_octree_info_get(self._h, C.byref(o)))
        return o

    def set_kernel(self, kernel: int):
        self._check(self._L.rto_set_kernel(self._h, kernel))

    def set_launch_order(self, policy: int, refresh_period: int = 0):
        """0 = centre-out, 1 = temporal (an earlier frame's per-tile cost; default). refresh_period: rebuild the
        table every n-th frame (0 keeps the current period, default 4)."""
        self._check(self._L.rto_set_launch_order(self._h, policy, refresh_period))

    # -- culling -----------------------------------------------------------
    def update_frustum(self, view, fov_deg, aspect, enable=True):
        v = np.ascontiguousarray(np.asarray(view, dtype=np.float32).reshape(16))
        self._check(self._L.rto_update_frustum(self._h, v.ctypes.data_as(C.POINTER(C.c_float)),
                                               _f(fov_deg), _f(aspect), 1 if enable else 0))

    def debug_update_frustum_planes(self, planes, margin: float):
        """Developer aid: a frustum update with caller-supplied planes (6 x (nx, ny, nz, d), normalised) and margin."""
        pl = np.ascontiguousarray(np.asarray(planes, dtype=np.float32).reshape(24))
        self._check(self._L.rto_debug_update_frustum_planes(self._h, pl.ctypes.data_as(C.POINTER(C.c_float)), _f(margin)))

    def debug_set_frustum_shortcut(self, enabled: bool):
        """False: rto_update_frustum always runs its kernel (the host-side proof that no node can be culled is skipped)."""
        self._check(self._L.rto_debug_set_frustum_shortcut(self._h, 1 if enabled else 0))

    def debug_last_frustum_update_proven(self) -> bool:
        v = C.c_int()
        self._check(self._L.rto_debug_last_frustum_update_proven(self._h, C.byref(v)))
        return bool(v.value)

    def forget_stream(self, stream: int):
        """Drop the launch-order tables kept for `stream` (call before destroying the stream)."""
        self._check(self._L.rto_forget_stream(self._h, C.c_void_p(stream) if stream else None))

    def debug_set_exact_grid(self, enabled: bool = True):
        """Test / A-B hook (not part of rto_hip.h): False = the general 12-plane child test even on a grid whose node planes are computed
        without rounding; True = automatic (default).  Returns (the resident grid is exact, the 9-plane form is in use).  Pixels never depend on it."""
        fn = self._L.rto_debug_set_exact_grid
        fn.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        info = (C.c_int * 2)()
        self._check(fn(self._h, 1 if enabled else 0, info))
        return bool(info[0]), bool(info[1])

    def debug_set_tile_mask(self, mode):
        """0 / False: occupancy mask off; 1 / True: on (default); 2: on, built by a launch of its own in front of the frame and
        consulted by every wave (tests).  Pixels never depend on it."""
        self._check(self._L.rto_debug_set_tile_mask(self._h, int(mode)))

    def debug_tile_mask_info(self):
        """(depth the mask's cells are taken from, number of cells); (0, 0): no mask for this octree."""
        lv, n = C.c_int(), C.c_int()
        self._check(self._L.rto_debug_tile_mask_info(self._h, C.byref(lv), C.byref(n)))
        return lv.value, n.value

    def debug_sort_violations(self) -> int:
        n = C.c_int()
        self._check(self._L.rto_debug_sort_violations(self._h, C.byref(n)))
        return n.value

    def download_visible_nodes(self) -> np.ndarray:
        cnt = C.c_int64()
        self._check(self._L.rto_download_visible_nodes(self._h, None, 0, C.byref(cnt)))
        out = np.zeros(cnt.value, NODE_DTYPE)
        if cnt.value:
            self._check(self._L.rto_download_visible_nodes(self._h, out.ctypes.data, cnt.value, C.byref(cnt)))
        return out

    # -- render ------------------------------------------------------------
    def render_host(self, frame: Frame) -> np.ndarray:
        out = np.empty((frame.height, frame.width, 4), np.float32)
        self._check(self._L.rto_render_host(self._h, C.byref(frame), out.ctypes.data))
        return out

    def render_resident(self, frame: Frame, mode: int = 0):
        """Asynchronous render into the context's own device framebuffer (0 octree, 1 triangles, 2 triangles + shadow)."""
        self._check(self._L.rto_render_resident(self._h, C.byref(frame), mode))

    def resident_frame(self):
        """(device pointer, width, height) of the frame render_resident left on the GPU."""
        p, w, h = C.c_void_p(), C.c_int(), C.c_int()
        self._check(self._L.rto_resident_frame(self._h, C.byref(p), C.byref(w), C.byref(h)))
        return p.value, w.value, h.value

    def download_resident(self) -> np.ndarray:
        _, w, h = self.resident_frame()
        out = np.empty((h, w, 4), np.float32)
        self._check(self._L.rto_download_resident(self._h, out.ctypes.data))
        return out

    def render_device(self, frame: Frame, d_out: int, part: Partition | None = None, stream: int = 0):
        self._check(self._L.rto_render_device(self._h, C.byref(frame), C.byref(part) if part else None,
                                              C.c_void_p(d_out), C.c_void_p(stream) if stream else None))

    def assemble_device(self, frame: Frame, part: Partition, d_gathered: int, d_frame: int, stream: int = 0):
        self._check(self._L.rto_assemble_device(self._h, C.byref(frame), C.byref(part), C.c_void_p(d_gathered),
                                                C.c_void_p(d_frame), C.c_void_p(stream) if stream else None))

    def render_shade_device(self, frame: Frame, d_shade: int, part: Partition | None = None, stream: int = 0):
        """4 bytes per pixel (Lambert term of the hit, -1 = miss): the multi-GPU gather payload."""
        self._check(self._L.rto_render_shade_device(self._h, C.byref(frame), C.byref(part) if part else None,
                                                    C.c_void_p(d_shade), C.c_void_p(stream) if stream else None))

    def assemble_shade_device(self, frame: Frame, part: Partition, d_gathered: int, d_frame: int, stream: int = 0):
        self._check(self._L.rto_assemble_shade_device(self._h, C.byref(frame), C.byref(part), C.c_void_p(d_gathered),
                                                      C.c_void_p(d_frame), C.c_void_p(stream) if stream else None))

    def assemble_batch_device(self, frame: Frame, part: Partition, d_gathered: int, batch: int, index: int, shade: bool, d_frame: int,
                              stream: int = 0):
        """Frame `index` of a gather that carried `batch` frames per rank ([rank][batch][rows][width])."""
        self._check(self._L.rto_assemble_batch_device(self._h, C.byref(frame), C.byref(part), C.c_void_p(d_gathered), batch, index,
                                                      1 if shade else 0, C.c_void_p(d_frame), C.c_void_p(stream) if stream else None))

    @staticmethod
    def frame_array(frames):
        """ctypes array of rto_frame for the batch entry points (build once per batch)."""
        return (Frame * len(frames))(*frames)

    def render_batch_device(self, frames_arr, d_out: int, frame_stride_bytes: int, part: Partition | None, shade: bool, stream: int = 0):
        self._check(self._L.rto_render_batch_device(self._h, frames_arr, len(frames_arr), C.byref(part) if part else None, 1 if shade else 0,
                                                    C.c_void_p(d_out), frame_stride_bytes, C.c_void_p(stream) if stream else None))

    def render_triangles_batch_device(self, frames_arr, d_out: int, frame_stride_bytes: int, shadow: bool = True, part: Partition | None = None,
                                      shade: bool = False, stream: int = 0):
        """Config 5's frames, up to 8 per kernel launch (rto_render_triangles_batch_device)."""
        self._check(self._L.rto_render_triangles_batch_device(self._h, frames_arr, len(frames_arr), C.byref(part) if part else None,
                                                              1 if shadow else 0, 1 if shade else 0, C.c_void_p(d_out), frame_stride_bytes,
                                                              C.c_void_p(stream) if stream else None))

    def assemble_batch_all_device(self, frames_arr, part: Partition, d_gathered: int, shade: bool, d_frames: int, frame_stride_bytes: int,
                                  stream: int = 0):
        self._check(self._L.rto_assemble_batch_all_device(self._h, frames_arr, len(frames_arr), C.byref(part), C.c_void_p(d_gathered),
                                                          1 if shade else 0, C.c_void_p(d_frames), frame_stride_bytes,
                                                          C.c_void_p(stream) if stream else None))

    def partition_rows(self, frame: Frame, part: Partition | None) -> int:
        return self._L.rto_partition_rows(C.byref(frame), C.byref(part) if part else None)

    def upload_leaf_triangles(self, tris: np.ndarray, tri_offset: np.ndarray):
        """Config 5: tris (n, 12) float32 = v0, v1, v2, face normal; tri_offset (numNodes+1,) int32."""
        tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 12)
        off = np.ascontiguousarray(tri_offset, dtype=np.int32)
        self._keep_tris = (tris, off)
        self._check(self._L.rto_upload_leaf_triangles(self._h, tris.ctypes.data if len(tris) else None, len(tris), off.ctypes.data))

    def build_leaf_triangles(self, voxels: np.ndarray | None = None):
        """Config 5: the leaf-triangle buffer built on the GPU for the resident octree.  voxels: uint8 (dimZ, dimY, dimX);
        None reuses the voxels rto_build_octree kept in HBM."""
        if voxels is None:
            self._check(self._L.rto_build_leaf_triangles(self._h, None, 0, 0, 0))
            return
        v = np.ascontiguousarray(voxels, dtype=np.uint8)
        dz, dy, dx = v.shape
        self._check(self._L.rto_build_leaf_triangles(self._h, v.ctypes.data, dx, dy, dz))

    def download_leaf_triangles(self):
        n = C.c_int64()
        self._check(self._L.rto_download_leaf_triangles(self._h, None, 0, None, C.byref(n)))
        tris = np.zeros((n.value, 12), np.float32)
        off = np.zeros(self.info().num_nodes + 1, np.int32)
        self._check(self._L.rto_download_leaf_triangles(self._h, tris.ctypes.data if n.value else None, n.value, off.ctypes.data, C.byref(n)))
        return tris, off

    def render_triangles_host(self, frame: Frame, shadow: bool = True, stats: bool = False):
        out = np.empty((frame.height, frame.width, 4), np.float32)
        s = Stats()
        self._check(self._L.rto_render_triangles_host(self._h, C.byref(frame), 1 if shadow else 0, out.ctypes.data,
                                                      C.byref(s) if stats else None))
        return (out, {"rays": s.rays, "pops": s.pops, "hits": s.hits}) if stats else out

    def render_triangles_device(self, frame: Frame, d_out: int, shadow: bool = True, part: Partition | None = None, stream: int = 0):
        self._check(self._L.rto_render_triangles_device(self._h, C.byref(frame), C.byref(part) if part else None,
                                                        1 if shadow else 0, C.c_void_p(d_out), C.c_void_p(stream) if stream else None))

    def render_triangles_shade_device(self, frame: Frame, d_shade: int, shadow: bool = True, part: Partition | None = None, stream: int = 0):
        self._check(self._L.rto_render_triangles_shade_device(self._h, C.byref(frame), C.byref(part) if part else None,
                                                              1 if shadow else 0, C.c_void_p(d_shade), C.c_void_p(stream) if stream else None))

    def octree_ray_skip(self, ro, rd, t_min=0.0, t_max=1e30, use_visibility=False) -> np.ndarray:
        """octreeRaySkip (VolumeRaycastRenderer.cpp:50-155) for n rays sharing the origin ro; rd: (n, 3)."""
        rd = np.ascontiguousarray(rd, dtype=np.float32).reshape(-1, 3)
        out = np.empty(len(rd), np.float32)
        o = (C.c_float * 3)(*[_f(x) for x in ro])
        self._check(self._L.rto_octree_ray_skip(self._h, o, rd.ctypes.data, len(rd), _f(t_min), _f(t_max),
                                                1 if use_visibility else 0, out.ctypes.data))
        return out

    def render_closest_host(self, frame: Frame, stats: bool = False):
        """The reference's closest-hit traversal (its earlier, block-commented shader): RGBA frame [, {rays, pops, hits}]."""
        out = np.empty((frame.height, frame.width, 4), np.float32)
        st = Stats()
        self._check(self._L.rto_render_closest_host(self._h, C.byref(frame), out.ctypes.data, C.byref(st) if stats else None))
        return (out, {"rays": st.rays, "pops": st.pops, "hits": st.hits}) if stats else out

    def render_closest_device(self, frame: Frame, d_out: int, part: Partition | None = None, stream: int = 0):
        self._check(self._L.rto_render_closest_device(self._h, C.byref(frame), C.byref(part) if part else None, C.c_void_p(d_out), C.c_void_p(stream)))

    def render_skip_host(self, frame: Frame, use_visibility=False, rgba=True, dist=True):
        """Nearest-hit render mode (octreeRaySkip per pixel): (rgba (H, W, 4) or None, dist (H, W) or None)."""
        o_rgba = np.empty((frame.height, frame.width, 4), np.float32) if rgba else None
        o_dist = np.empty((frame.height, frame.width), np.float32) if dist else None
        self._check(self._L.rto_render_skip_host(self._h, C.byref(frame), 1 if use_visibility else 0,
                                                 o_rgba.ctypes.data if rgba else None, o_dist.ctypes.data if dist else None))
        return o_rgba, o_dist

    def render_skip_device(self, frame: Frame, d_rgba: int, d_dist: int, use_visibility=False, part: Partition | None = None, stream: int = 0):
        self._check(self._L.rto_render_skip_device(self._h, C.byref(frame), C.byref(part) if part else None, 1 if use_visibility else 0,
                                                   C.c_void_p(d_rgba) if d_rgba else None, C.c_void_p(d_dist) if d_dist else None,
                                                   C.c_void_p(stream) if stream else None))

    def probe_skip_host(self, view, cam_pos, aspect, last=0.0, use_visibility=False, with_probes=False):
        """octreeSkipT as drawRaycast computes it, one launch: returns the new value (and the 49 probe distances)."""
        v = np.ascontiguousarray(np.asarray(view, dtype=np.float32).reshape(16))
        p = np.ascontiguousarray(np.asarray(cam_pos, dtype=np.float32).reshape(3))
        io = C.c_float(_f(last))
        probes = np.zeros(49, np.float32) if with_probes else None
        self._check(self._L.rto_probe_skip_host(self._h, v.ctypes.data_as(C.POINTER(C.c_float)), p.ctypes.data_as(C.POINTER(C.c_float)), _f(aspect),
                                                1 if use_visibility else 0, C.byref(io), probes.ctypes.data if with_probes else None))
        return (np.float32(io.value), probes) if with_probes else np.float32(io.value)

    def probe_skip_device(self, view, cam_pos, aspect, d_skip: int, use_visibility=False, stream: int = 0):
        v = np.ascontiguousarray(np.asarray(view, dtype=np.float32).reshape(16))
        p = np.ascontiguousarray(np.asarray(cam_pos, dtype=np.float32).reshape(3))
        self._check(self._L.rto_probe_skip_device(self._h, v.ctypes.data_as(C.POINTER(C.c_float)), p.ctypes.data_as(C.POINTER(C.c_float)), _f(aspect),
                                                  1 if use_visibility else 0, C.c_void_p(d_skip), C.c_void_p(stream) if stream else None))

    def frame_stats(self, frame: Frame) -> dict:
        s = Stats()
        self._check(self._L.rto_frame_stats(self._h, C.byref(frame), C.byref(s)))
        return {"rays": s.rays, "pops": s.pops, "hits": s.hits, "capped": s.capped}

    def render_steps(self, frame: Frame) -> np.ndarray:
        out = np.zeros((frame.height, frame.width), np.int32)
        self._check(self._L.rto_render_steps_host(self._h, C.byref(frame), out.ctypes.data))
        return out

    def debug_timeline(self, frame: Frame) -> np.ndarray:
        """(tiles, 8) int32 records, see rto_debug_timeline."""
        n = C.c_int64()
        self._check(self._L.rto_debug_timeline(self._h, C.byref(frame), None, 0, C.byref(n)))
        out = np.zeros((n.value, 8), np.int32)
        self._check(self._L.rto_debug_timeline(self._h, C.byref(frame), out.ctypes.data, n.value, C.byref(n)))
        return out

    def debug_tile_cost(self) -> np.ndarray:
        n = C.c_int64()
        self._check(self._L.rto_debug_tile_cost(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.int32)
        self._check(self._L.rto_debug_tile_cost(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def debug_set_tile_order(self, order):
        if order is None:
            self._check(self._L.rto_debug_set_tile_order(self._h, None, 0))
            return
        o = np.ascontiguousarray(order, dtype=np.int32)
        self._check(self._L.rto_debug_set_tile_order(self._h, o.ctypes.data, len(o)))

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        self._check(self._L.rto_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def timing_begin(self, capacity: int):
        self._check(self._L.rto_timing_begin(self._h, capacity))

    def timing_read(self) -> np.ndarray:
        n = C.c_int()
        self._check(self._L.rto_timing_read(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.float32)
        if n.value:
            self._check(self._L.rto_timing_read(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def scene_bounds(self) -> SceneBounds:
        b = SceneBounds()
        self._check(self._L.rto_scene_bounds_get(self._h, C.byref(b)))
        return b

    @property
    def stream(self) -> int:
        """The context's own hipStream_t as an integer handle."""
        return self._L.rto_stream(self._h) or 0

    def synchronize(self):
        self._check(self._L.rto_synchronize(self._h))


def comm_unique_id() -> bytes:
    """rank 0: the 128 bytes every other rank needs for Comm(ctx, world, rank, id)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = load().rto_comm_unique_id(buf)
    if rc != RTO_OK:
        raise RtoError(rc, load().rto_last_error(None).decode())
    return buf.raw


class Comm:
    """rto_comm: one rank of the screen-split renderer (one process per GPU).  submit() renders this rank's bands of a batch
    of frames, ONE grouped RCCL send/recv lands them on rank 0, which assembles them into `d_frames`; flush() waits."""

    def __init__(self, ctx: Context, world: int, rank: int, unique_id: bytes, band_rows: int = 16):
        self._L = load()
        self.ctx, self.world, self.rank = ctx, world, rank
        h = C.c_void_p()
        idbuf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        rc = self._L.rto_comm_create(ctx._h, world, rank, idbuf, band_rows, C.byref(h))
        if rc != RTO_OK:
            raise RtoError(rc, self._L.rto_last_error(ctx._h).decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.rto_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != RTO_OK:
            raise RtoError(rc, self._L.rto_comm_last_error(self._h).decode())

    def submit(self, frames_arr, d_frames: int = 0, frame_stride_bytes: int = 0, mode: int = RESIDENT_OCTREE):
        """frames_arr: Context.frame_array([...]); d_frames (rank 0): device pointer of len(frames_arr) RGBA32F frames."""
        self._check(self._L.rto_comm_submit(self._h, frames_arr, len(frames_arr), mode, C.c_void_p(d_frames) if d_frames else None, frame_stride_bytes))

    def flush(self, timeout_ms: int = 0):
        """Waits for every submitted batch.  timeout_ms > 0: at most that long -- on expiry (a peer that never sent) the
        communicator is aborted and dead, RtoError(RTO_E_TIMEOUT)."""
        self._check(self._L.rto_comm_flush_timeout(self._h, int(timeout_ms)))

    def is_dead(self) -> bool:
        return bool(self._L.rto_comm_is_dead(self._h))

    def ranks_seen(self) -> int:
        """ncclCommCount: the ranks RCCL says take part in this communicator."""
        n = C.c_int(0)
        self._check(self._L.rto_comm_ranks_seen(self._h, C.byref(n)))
        return int(n.value)

    def debug_abort(self):
        """Test hook: what a flush timeout does (ncclCommAbort, the communicator is dead)."""
        self._check(self._L.rto_comm_debug_abort(self._h))

    def debug_last_payload(self):
        """(floats shipped for the last batch, floats whole rows would have been) for this rank."""
        a, b = C.c_int64(), C.c_int64()
        self._check(self._L.rto_comm_debug_last_payload(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def debug_set_rehearsal_clear(self, enabled: bool):
        self._check(self._L.rto_comm_debug_set_rehearsal_clear(self._h, 1 if enabled else 0))

    def debug_set_timing(self, enabled: bool):
        self._check(self._L.rto_comm_debug_set_timing(self._h, 1 if enabled else 0))

    def debug_last_timing(self):
        """(render ms, gather (+ assembly on rank 0) ms) of the batch submitted last; call after flush()."""
        ms = (C.c_float * 2)()
        self._check(self._L.rto_comm_debug_last_timing(self._h, ms))
        return float(ms[0]), float(ms[1])

    def debug_rehearse(self, as_world: int, as_rank: int = 0):
        """One-rank communicator only: split frames as rank `as_rank` of `as_world` GPUs (0 switches it off)."""
        self._check(self._L.rto_comm_debug_rehearse(self._h, as_world, as_rank))

    @property
    def stream(self) -> int:
        return self._L.rto_comm_stream(self._h) or 0


class CommGroup:
    """rto_comm_create_all: one process driving several GPUs (one Context each, all on different devices).  render_resident()
    sends one frame through every rank into rank 0's resident framebuffer; submit()/flush() are the batched, pipelined form."""

    def __init__(self, contexts, band_rows: int = 16):
        self._L = load()
        self.contexts = list(contexts)
        n = len(self.contexts)
        ctxs = (C.c_void_p * n)(*[c._h for c in self.contexts])
        self._handles = (C.c_void_p * n)()
        rc = self._L.rto_comm_create_all(ctxs, n, band_rows, self._handles)
        if rc != RTO_OK:
            self._handles = None
            raise RtoError(rc, self._L.rto_last_error(self.contexts[0]._h).decode())

    def close(self):
        if getattr(self, "_handles", None):
            for h in self._handles:
                if h:
                    self._L.rto_comm_destroy(h)
            self._handles = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != RTO_OK:
            raise RtoError(rc, self._L.rto_comm_last_error(self._handles[0]).decode())

    def render_resident(self, frame: Frame, mode: int = RESIDENT_OCTREE):
        self._check(self._L.rto_comm_render_resident_all(self._handles, len(self.contexts), C.byref(frame), mode))

    def submit(self, frames_arr, d_frames: int, frame_stride_bytes: int, mode: int = RESIDENT_OCTREE):
        self._check(self._L.rto_comm_submit_all(self._handles, len(self.contexts), frames_arr, len(frames_arr), mode,
                                                C.c_void_p(d_frames), frame_stride_bytes))

    def flush(self):
        for h in self._handles:
            self._check(self._L.rto_comm_flush(h))

    def ranks_seen(self) -> int:
        n = C.c_int(0)
        self._check(self._L.rto_comm_ranks_seen(self._handles[0], C.byref(n)))
        return int(n.value)

    def is_dead(self):
        """one flag per member"""
        return [bool(self._L.rto_comm_is_dead(h)) for h in self._handles]

    def debug_abort(self, member: int = 0):
        """Test hook: abort ONE member as a flush timeout would; the whole group is dead afterwards."""
        self._check(self._L.rto_comm_debug_abort(self._handles[member]))
