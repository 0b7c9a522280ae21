// OctreeVoxel.cpp -- see OctreeVoxel.h.  Semantics follow 453-skeleton/OctreeVoxel.cpp:534-778 and
// 453-skeleton/main.cpp:337-422; the construction strategy (occupancy pyramid) is this repo's own.
#include "OctreeVoxel.h"

#include <algorithm>
#include <cmath>
#include <limits>

const int edgeToCorner[12][2] = { { 0, 1 }, { 1, 2 }, { 2, 3 }, { 3, 0 }, { 4, 5 }, { 5, 6 },
                                  { 6, 7 }, { 7, 4 }, { 0, 4 }, { 1, 5 }, { 2, 6 }, { 3, 7 } };

std::unordered_map<long long, OctreeNode*> g_octreeMap;

long long buildKey(int x, int y, int z) {
    // same packing as the reference (OctreeVoxel.cpp:552-554): x<<20 | y<<10 | z
    return (static_cast<long long>(x) << 20) | (static_cast<long long>(y) << 10) | static_cast<long long>(z);
}

VoxelState getVoxelSafe(const VoxelGrid& grid, int x, int y, int z) {
    const bool inside = x >= 0 && y >= 0 && z >= 0 && x < grid.dimX && y < grid.dimY && z < grid.dimZ;
    return inside ? grid.data[grid.index(x, y, z)] : VoxelState::EMPTY;
}

OctreeNode* getParentCube(OctreeNode* node) { return node ? node->parent : nullptr; }

int getSubcubeIndex(int x, int y, int z, int halfSize, int x0, int y0, int z0) {
    return (x >= x0 + halfSize ? 1 : 0) | (y >= y0 + halfSize ? 2 : 0) | (z >= z0 + halfSize ? 4 : 0);
}

std::vector<OctreeNode*> getNeighbors(OctreeNode* node, const std::unordered_map<long long, OctreeNode*>& nodeMap) {
    // the 6 face neighbours of equal size, order +x -x +y -y +z -z (OctreeVoxel.cpp:559-630)
    std::vector<OctreeNode*> out;
    if (!node) return out;
    static const int dir[6][3] = { { 1, 0, 0 }, { -1, 0, 0 }, { 0, 1, 0 }, { 0, -1, 0 }, { 0, 0, 1 }, { 0, 0, -1 } };
    for (const auto& d : dir) {
        auto it = nodeMap.find(buildKey(node->x + d[0] * node->size, node->y + d[1] * node->size, node->z + d[2] * node->size));
        if (it != nodeMap.end()) out.push_back(it->second);
    }
    return out;
}

// ------------------------------------------------------------------ octree construction
namespace {

// Occupancy pyramid: level 0 = voxels; a level-l cell covers a 2^l cube of voxels clipped to the
// grid.  State: 0 all EMPTY (incl. the out-of-grid part), 1 all FILLED, 2 mixed.
struct Pyramid {
    struct Level { int nx, ny, nz; std::vector<uint8_t> s; };
    std::vector<Level> levels;

    uint8_t state(int level, int x, int y, int z) const {   // x,y,z in voxel units
        const Level& L = levels[level];
        const int i = x >> level, j = y >> level, k = z >> level;
        if (i >= L.nx || j >= L.ny || k >= L.nz) return 0;    // wholly outside the grid
        return L.s[(size_t)i + (size_t)j * L.nx + (size_t)k * L.nx * L.ny];
    }
};

Pyramid buildPyramid(const VoxelGrid& g, int rootLevel) {
    Pyramid p;
    p.levels.resize(rootLevel + 1);
    Pyramid::Level& L0 = p.levels[0];
    L0.nx = g.dimX; L0.ny = g.dimY; L0.nz = g.dimZ;
    L0.s.resize((size_t)g.dimX * g.dimY * g.dimZ);
    for (size_t i = 0; i < L0.s.size(); i++) L0.s[i] = g.data[i] == VoxelState::FILLED ? 1 : 0;
    for (int l = 1; l <= rootLevel; l++) {
        const Pyramid::Level& C = p.levels[l - 1];
        Pyramid::Level& P = p.levels[l];
        P.nx = (C.nx + 1) / 2; P.ny = (C.ny + 1) / 2; P.nz = (C.nz + 1) / 2;
        P.s.resize((size_t)P.nx * P.ny * P.nz);
        // A parent cell sticks out of the grid (its out-of-grid voxels are EMPTY) when a child slot is
        // missing at level l-1, or when the grid dimension is not a multiple of the parent's extent.
        const int ext = 1 << l;
        for (int k = 0; k < P.nz; k++)
            for (int j = 0; j < P.ny; j++)
                for (int i = 0; i < P.nx; i++) {
                    bool any0 = (i + 1) * ext > g.dimX || (j + 1) * ext > g.dimY || (k + 1) * ext > g.dimZ;
                    bool any1 = false, mixed = false;
                    for (int c = 0; c < 8 && !mixed; c++) {
                        const int ci = 2 * i + (c & 1), cj = 2 * j + ((c >> 1) & 1), ck = 2 * k + (c >> 2);
                        if (ci >= C.nx || cj >= C.ny || ck >= C.nz) continue;   // covered by any0
                        const uint8_t s = C.s[(size_t)ci + (size_t)cj * C.nx + (size_t)ck * C.nx * C.ny];
                        if (s == 2) mixed = true;
                        else if (s == 1) any1 = true;
                        else any0 = true;
                    }
                    P.s[(size_t)i + (size_t)j * P.nx + (size_t)k * P.nx * P.ny] = (mixed || (any0 && any1)) ? 2 : (any1 ? 1 : 0);
                }
    }
    return p;
}

OctreeNode* buildNode(const Pyramid& pyr, int x0, int y0, int z0, int size, int level,
                      std::unordered_map<long long, OctreeNode*>& nodeMap) {
    OctreeNode* node = new OctreeNode(x0, y0, z0, size);
    nodeMap[buildKey(x0, y0, z0)] = node;     // later (smaller) nodes at the same corner overwrite, as upstream
    const uint8_t s = pyr.state(level, x0, y0, z0);
    if (size == 1 || s != 2) {
        node->isLeaf = true;
        node->isUniform = true;
        node->isSolid = (s == 1);
        return node;
    }
    const int half = size / 2;
    for (int i = 0; i < 8; i++) {
        OctreeNode* child = buildNode(pyr, x0 + ((i & 1) ? half : 0), y0 + ((i & 2) ? half : 0),
                                      z0 + ((i & 4) ? half : 0), half, level - 1, nodeMap);
        child->parent = node;
        node->children[i] = child;
    }
    return node;
}

}  // namespace

OctreeNode* createOctreeFromVoxelGrid(const VoxelGrid& grid) {
    if (grid.dimX == 0 || grid.dimY == 0 || grid.dimZ == 0) return nullptr;
    const int maxDim = std::max({ grid.dimX, grid.dimY, grid.dimZ });
    int rootLevel = 0;
    while ((1 << rootLevel) < maxDim) rootLevel++;
    g_octreeMap.clear();
    const Pyramid pyr = buildPyramid(grid, rootLevel);
    return buildNode(pyr, 0, 0, 0, 1 << rootLevel, rootLevel, g_octreeMap);
}

void freeOctree(OctreeNode* node) {
    if (!node) return;
    for (OctreeNode* c : node->children) freeOctree(c);
    delete node;
}

// ------------------------------------------------------------------ test scene (main.cpp helpers)
std::vector<float> generateTestVolume(int dimX, int dimY, int dimZ) {
    // hollow shell: density +1 where rInner <= |p - centre| <= rOuter, else -1 (main.cpp:337-372)
    std::vector<float> volume((size_t)dimX * dimY * dimZ, 0.f);
    const float cx = 0.5f * (dimX - 1), cy = 0.5f * (dimY - 1), cz = 0.5f * (dimZ - 1);
    const float minDim = std::min({ float(dimX), float(dimY), float(dimZ) });
    const float rOuter = 0.4f * minDim, rInner = 0.2f * minDim;
    size_t idx = 0;
    for (int z = 0; z < dimZ; z++)
        for (int y = 0; y < dimY; y++)
            for (int x = 0; x < dimX; x++, idx++) {
                const float dx = x - cx, dy = y - cy, dz = z - cz;
                const float dist = std::sqrt(dx * dx + dy * dy + dz * dz);
                volume[idx] = (dist < rInner || dist > rOuter) ? -1.0f : 1.0f;
            }
    return volume;
}

bool recenterFilledVoxels(VoxelGrid& grid) {
    // shift grid.min so the bounding box of FILLED voxel centres is centred on the origin (main.cpp:376-422)
    const float big = std::numeric_limits<float>::max();
    float lo[3] = { big, big, big }, hi[3] = { -big, -big, -big };
    for (int z = 0; z < grid.dimZ; ++z)
        for (int y = 0; y < grid.dimY; ++y)
            for (int x = 0; x < grid.dimX; ++x) {
                if (grid.data[(size_t)x + (size_t)y * grid.dimX + (size_t)z * grid.dimX * grid.dimY] != VoxelState::FILLED) continue;
                const float c[3] = { grid.minX + (x + 0.5f) * grid.voxelSize, grid.minY + (y + 0.5f) * grid.voxelSize,
                                     grid.minZ + (z + 0.5f) * grid.voxelSize };
                for (int a = 0; a < 3; a++) {
                    if (c[a] < lo[a]) lo[a] = c[a];
                    if (c[a] > hi[a]) hi[a] = c[a];
                }
            }
    if (lo[0] > hi[0]) return false;
    grid.minX -= 0.5f * (lo[0] + hi[0]);
    grid.minY -= 0.5f * (lo[1] + hi[1]);
    grid.minZ -= 0.5f * (lo[2] + hi[2]);
    return true;
}

VoxelGrid makeTestSphereGrid(int dim) {
    // main.cpp:1052-1070 (the !useGDB branch) followed by :1074
    VoxelGrid grid;
    grid.dimX = grid.dimY = grid.dimZ = dim;
    grid.minX = grid.minY = grid.minZ = -0.5f;
    grid.voxelSize = 1.f / dim;
    const std::vector<float> vol = generateTestVolume(dim, dim, dim);
    grid.data.resize(vol.size());
    for (size_t i = 0; i < vol.size(); i++) grid.data[i] = vol[i] > 0.0f ? VoxelState::FILLED : VoxelState::EMPTY;
    recenterFilledVoxels(grid);
    return grid;
}
