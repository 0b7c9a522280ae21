// CacheUtils.h -- the reference's voxel-grid cache file (sceneCache.bin), 453-skeleton/CacheUtils.h:1-9.
// Layout (CacheUtils.cpp:5-30): int32 dimX,dimY,dimZ; float minX,minY,minZ,voxelSize; uint64 count; count bytes.
#pragma once

#include <string>

#include "OctreeVoxel.h"

bool saveVoxelGrid(const std::string& filename, const VoxelGrid& grid);
bool loadVoxelGrid(const std::string& filename, VoxelGrid& grid);
// Z-slab [startLayer, startLayer + numLayers) only; grid.dimZ/minZ are adjusted to the slab.
bool loadVoxelGridPartial(const std::string& filename, VoxelGrid& grid, int startLayer, int numLayers);
