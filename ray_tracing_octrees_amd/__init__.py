"""ray_tracing_octrees_amd -- MI355X-native ray->octree traversal (drop-in for the
reference's RayTracerBVH hot path).

Layers (all thin; the product is the HIP library):
  include/rto_hip.h                 C ABI of librto_hip.so (hand-written gfx950 kernels)
  ray_tracing_octrees_amd/csrc/     the kernels + ABI implementation
  ray_tracing_octrees_amd/host/     C++17 host layer with the reference's class/API names
                                    (VoxelGrid, OctreeNode, createOctreeFromVoxelGrid, Camera, Frustum,
                                     CacheUtils, RayTracerBVH) -> librto_host.so
  ray_tracing_octrees_amd.hip       ctypes binding of the C ABI (flat GPUNodes arrays, device pointers)
  ray_tracing_octrees_amd.host      ctypes face of the C++ host layer (same names as the reference)

The HIP library is mandatory: nothing in this package computes pixels on the CPU.
"""
from __future__ import annotations

from . import hip, host
from .hip import (KERNEL_AUTO, KERNEL_GENERIC, KERNEL_PACKED, KERNEL_PACKED_PERSISTENT, KERNEL_PACKED_V1, KERNEL_PACKED_V3, NODE_DTYPE, Context, Frame, Partition,
                  RtoError, make_frame)
from .host import (Camera, MarchingCubesRenderer, OctreeNode, RayTracerBVH, VoxelGrid, buildLeafTriangles,
                   createOctreeFromVoxelGrid,
                   freeOctree, getVoxelSafe, loadVoxelGrid, loadVoxelGridPartial, localMC, saveVoxelGrid)

__all__ = [
    "RayTracerBVH", "VoxelGrid", "OctreeNode", "Camera", "createOctreeFromVoxelGrid", "freeOctree",
    "getVoxelSafe", "loadVoxelGrid", "loadVoxelGridPartial", "saveVoxelGrid", "localMC", "MarchingCubesRenderer", "buildLeafTriangles",
    "Context", "Frame", "Partition", "RtoError", "make_frame", "NODE_DTYPE",
    "KERNEL_AUTO", "KERNEL_GENERIC", "KERNEL_PACKED", "KERNEL_PACKED_PERSISTENT", "KERNEL_PACKED_V1", "KERNEL_PACKED_V3", "hip", "host",
]
