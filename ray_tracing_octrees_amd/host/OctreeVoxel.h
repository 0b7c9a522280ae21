// OctreeVoxel.h -- host-side scene structures with the reference's names and semantics
// (453-skeleton/OctreeVoxel.h:1-86), so code written against the reference compiles against
// this header unchanged apart from the math type: rtmath::vec3 stands where glm::vec3 stood
// (same 3-float layout).
//
// createOctreeFromVoxelGrid() returns the same pointer tree the reference builds
// (453-skeleton/OctreeVoxel.cpp:704-778: root (0,0,0) of edge pow2 >= max dim; a node is a leaf
// iff its edge is 1 or every voxel under it -- out-of-grid voxels count as EMPTY -- is equal;
// children in bit order x=1,y=2,z=4) but decides uniformity from a bottom-up occupancy pyramid
// (one pass over the voxels) instead of re-scanning the region at every level.
#pragma once

#include <array>
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "rtmath.h"

enum class VoxelState : uint8_t { EMPTY = 0, FILLED = 1 };

// corner pairs of the 12 cube edges, Marching-Cubes numbering (OctreeVoxel.h:16-20)
extern const int edgeToCorner[12][2];

struct MCTriangle {
    rtmath::vec3 v[3];
    rtmath::vec3 normal[3];
};

struct VoxelGrid {
    int dimX = 0, dimY = 0, dimZ = 0;
    float minX = 0.f, minY = 0.f, minZ = 0.f;
    float voxelSize = 1.f;
    std::vector<VoxelState> data;   // x fastest

    int index(int x, int y, int z) const { return x + y * dimX + z * (dimX * dimY); }
};

struct OctreeNode {
    int x, y, z;
    int size;
    bool isLeaf;
    bool isSolid;
    bool isUniform;
    OctreeNode* parent;
    OctreeNode* children[8];

    OctreeNode(int x_, int y_, int z_, int size_)
        : x(x_), y(y_), z(z_), size(size_), isLeaf(false), isSolid(false), isUniform(false), parent(nullptr) {
        for (auto& c : children) c = nullptr;
    }
};

// Position-keyed index of the nodes of the most recently built tree (reference: global
// g_octreeMap defined in Renderer.cpp:11 and refilled by createOctreeFromVoxelGrid).
extern std::unordered_map<long long, OctreeNode*> g_octreeMap;
long long buildKey(int x, int y, int z);

OctreeNode* createOctreeFromVoxelGrid(const VoxelGrid& grid);
void freeOctree(OctreeNode* node);
VoxelState getVoxelSafe(const VoxelGrid& grid, int x, int y, int z);

OctreeNode* getParentCube(OctreeNode* node);
int getSubcubeIndex(int x, int y, int z, int halfSize, int x0, int y0, int z0);
std::vector<OctreeNode*> getNeighbors(OctreeNode* node, const std::unordered_map<long long, OctreeNode*>& nodeMap);

// Marching Cubes over the cells [x0,x0+size) x [y0,..) x [z0,..) of the grid (cell = 8 neighbouring voxels;
// FILLED -> -1, EMPTY / out of grid -> +1, iso 0), same output order as upstream (OctreeVoxel.cpp:780-879).
std::vector<MCTriangle> localMC(const VoxelGrid& grid, int x0, int y0, int z0, int size);

// Leaf-triangle buffer of the triangle ray path (config 5; no upstream counterpart): for every node of a flat
// GPUNodes array (15 int32 each) that is a leaf, its localMC triangles, 12 floats each (v0, v1, v2, face normal),
// in node order; triOffset[i]..triOffset[i+1] is node i's range (what MarchingCubesRenderer emits per leaf).
struct GPUNodesView { const int32_t* data; int64_t count; };
void buildLeafTriangles(const VoxelGrid& grid, const GPUNodesView& nodes, std::vector<float>& tris, std::vector<int32_t>& triOffset);

// Test scene of the reference app (453-skeleton/main.cpp:337-372, 1052-1070, 376-422); these are
// file-static helpers of main.cpp upstream, exposed here because every benchmark config uses them.
std::vector<float> generateTestVolume(int dimX, int dimY, int dimZ);
VoxelGrid makeTestSphereGrid(int dim);            // density > 0 -> FILLED, min -0.5, voxelSize 1/dim, recentred
bool recenterFilledVoxels(VoxelGrid& grid);       // false when nothing is FILLED
