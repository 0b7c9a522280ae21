"""ray_tracing_octrees_amd -- MI355X-native ray->octree traversal (drop-in for the
reference's RayTracerBVH hot path).

Layers (all of them thin; the product is the HIP library):
  include/rto_hip.h                     C ABI of librto_hip.so (hand-written gfx950 kernels)
  ray_tracing_octrees_amd/csrc/         the kernels + ABI implementation
  ray_tracing_octrees_amd/host/         C++17 host layer with the reference's class/API names
  ray_tracing_octrees_amd/hip.py        ctypes binding of the C ABI
  ray_tracing_octrees_amd.RayTracerBVH  Python mirror of 453-skeleton/RayTracerBVH.h:28-80
"""
from __future__ import annotations

import sys

import numpy as np

from . import hip
from .hip import (KERNEL_AUTO, KERNEL_GENERIC, KERNEL_PACKED, NODE_DTYPE, Context, Frame, Partition,
                  RtoError, make_frame)

__all__ = ["RayTracerBVH", "Context", "Frame", "Partition", "RtoError", "make_frame", "NODE_DTYPE",
           "KERNEL_AUTO", "KERNEL_GENERIC", "KERNEL_PACKED", "hip"]


class RayTracerBVH:
    """Same call sequence and error behaviour as the reference class
    (453-skeleton/RayTracerBVH.h:28-80, used at 453-skeleton/main.cpp:1127-1131, 1357-1363):

        rt = RayTracerBVH(); rt.ensureComputeInitialized(); rt.setOctree(nodes, grid_min, voxel_size)
        img = rt.renderSceneComputeWithCulling(view, cam_pos, W, H, aspect, 45.0, update_frustum)

    Differences forced by the medium: the octree arrives as the flat GPUNodes array (what
    setOctree builds at RayTracerBVH.cpp:443-490; the C++ host layer does that step from an
    OctreeNode* tree), the camera as its view matrix + position (what getView()/getPos() return),
    and the frame is handed back as an (H, W, 4) float32 array instead of being drawn to a quad.
    Like the reference, rendering before initialisation prints to stderr and returns nothing,
    and rendering with no nodes is a silent no-op.
    """

    def __init__(self, device: int = 0):
        self._device = device
        self._ctx: Context | None = None
        self._num_nodes = 0
        self._frustum_culling_enabled = True   # m_frustumCullingEnabled(true), RayTracerBVH.cpp:403

    # -- reference API -----------------------------------------------------
    def ensureComputeInitialized(self):
        """RayTracerBVH.cpp:508-612 (shader compile) -> create the HIP context."""
        if self._ctx is None:
            self._ctx = Context(self._device)

    def setOctree(self, nodes: np.ndarray, grid_min, voxel_size: float):
        """RayTracerBVH.cpp:430-505.  An empty array is ignored, like a null root (:439)."""
        if nodes is None or len(nodes) == 0:
            self._num_nodes = 0
            return
        self.ensureComputeInitialized()
        self._ctx.upload_octree(nodes, grid_min, voxel_size)
        self._num_nodes = len(nodes)

    def setFrustumCullingEnabled(self, enabled: bool):
        self._frustum_culling_enabled = bool(enabled)

    def renderSceneCompute(self, view, cam_pos, width, height, aspect, fovDeg):
        """RayTracerBVH.cpp:614-704."""
        if self._ctx is None:
            print("[RayTracerBVH] Compute pipeline not initialized or failed.", file=sys.stderr)
            return None
        if self._num_nodes <= 0:
            return None
        return self._ctx.render_host(make_frame(view, cam_pos, aspect, fovDeg, width, height))

    def renderSceneComputeWithCulling(self, view, cam_pos, width, height, aspect, fovDeg, updateFrustum: bool):
        """RayTracerBVH.cpp:706-892: optional frustum update, then the same dispatch."""
        if self._ctx is None:
            print("[RayTracerBVH] Compute pipeline not initialized or failed.", file=sys.stderr)
            return None
        if self._num_nodes <= 0:
            print("No nodes to render.")
            return None
        if updateFrustum:
            self._ctx.update_frustum(view, fovDeg, aspect, enable=True)
        return self._ctx.render_host(make_frame(view, cam_pos, aspect, fovDeg, width, height))

    # -- snake_case aliases ---------------------------------------------------
    ensure_compute_initialized = ensureComputeInitialized
    set_octree = setOctree
    set_frustum_culling_enabled = setFrustumCullingEnabled
    render_scene_compute = renderSceneCompute
    render_scene_compute_with_culling = renderSceneComputeWithCulling

    @property
    def context(self) -> Context | None:
        return self._ctx
