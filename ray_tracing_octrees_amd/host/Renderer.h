// Renderer.h -- the reference's mesh-extractor interface (453-skeleton/Renderer.h:10-24): kept so that code
// written against it compiles; only the Marching-Cubes extractor is provided (its triangles feed the
// leaf-triangle ray path).  VoxelCubeRenderer / dual contouring are rasterised-mesh modes outside this repo's scope.
#pragma once

#include <vector>

#include "OctreeVoxel.h"

class Renderer {
public:
    virtual std::vector<MCTriangle> render(const OctreeNode* node, const VoxelGrid& grid, int x0, int y0, int z0, int size) = 0;
    virtual ~Renderer() = default;
};

// localMC on every leaf below `node`, children in index order (453-skeleton/Renderer.cpp:14-36)
class MarchingCubesRenderer : public Renderer {
public:
    std::vector<MCTriangle> render(const OctreeNode* node, const VoxelGrid& grid, int x0, int y0, int z0, int size) override;
};
