// LocalMC.cpp -- per-leaf Marching Cubes with the semantics of 453-skeleton/OctreeVoxel.cpp:633-640 (vertexInterp)
// and :780-879 (localMC).  The case table (mc_cases.inc) was derived by probing the reference's compiled function
// on all 256 corner configurations (tools/gen_mc_tables.py); tests compare against triangles the reference produced.
#include <cmath>

#include "OctreeVoxel.h"
#include "mc_cases.inc"

using rtmath::vec3;

namespace {

// corner i of a cell at (x,y,z): offsets in the reference's numbering (OctreeVoxel.cpp:802-817)
const int kCorner[8][3] = { { 0, 0, 0 }, { 1, 0, 0 }, { 1, 1, 0 }, { 0, 1, 0 }, { 0, 0, 1 }, { 1, 0, 1 }, { 1, 1, 1 }, { 0, 1, 1 } };

inline vec3 vertexInterp(float iso, const vec3& p1, const vec3& p2, float v1, float v2) {
    if (std::abs(iso - v1) < 0.00001f) return p1;
    if (std::abs(iso - v2) < 0.00001f) return p2;
    if (std::abs(v1 - v2) < 0.00001f) return p1;
    const float mu = (iso - v1) / (v2 - v1);
    return p1 + mu * (p2 - p1);
}

inline int hexval(char c) { return c <= '9' ? c - '0' : c - 'a' + 10; }

}  // namespace

namespace {
template <class Emit>
void forEachLocalMCTriangle(const VoxelGrid& grid, int x0, int y0, int z0, int size, Emit emit);
}

std::vector<MCTriangle> localMC(const VoxelGrid& grid, int x0, int y0, int z0, int size) {
    std::vector<MCTriangle> out;
    forEachLocalMCTriangle(grid, x0, y0, z0, size, [&](const MCTriangle& t) { out.push_back(t); });
    return out;
}

void buildLeafTriangles(const VoxelGrid& grid, const GPUNodesView& nodes, std::vector<float>& tris, std::vector<int32_t>& triOffset) {
    tris.clear();
    triOffset.assign((size_t)nodes.count + 1, 0);
    for (int64_t i = 0; i < nodes.count; i++) {
        triOffset[(size_t)i] = (int32_t)(tris.size() / 12);
        const int32_t* nd = nodes.data + i * 15;                  // x, y, z, size, isLeaf, ...
        if (nd[4] != 1) continue;
        forEachLocalMCTriangle(grid, nd[0], nd[1], nd[2], nd[3], [&](const MCTriangle& t) {
            for (int v = 0; v < 3; v++) { tris.push_back(t.v[v].x); tris.push_back(t.v[v].y); tris.push_back(t.v[v].z); }
            tris.push_back(t.normal[0].x); tris.push_back(t.normal[0].y); tris.push_back(t.normal[0].z);
        });
    }
    triOffset[(size_t)nodes.count] = (int32_t)(tris.size() / 12);
}

namespace {
template <class Emit>
void forEachLocalMCTriangle(const VoxelGrid& grid, int x0, int y0, int z0, int size, Emit emit) {
    const float vx = grid.voxelSize;
    auto scalar = [&](int x, int y, int z) -> float {
        if (x < 0 || y < 0 || z < 0 || x >= grid.dimX || y >= grid.dimY || z >= grid.dimZ) return 1.0f;
        return grid.data[grid.index(x, y, z)] == VoxelState::FILLED ? -1.0f : 1.0f;
    };
    for (int z = z0; z < z0 + size && z < grid.dimZ - 1; z++)
        for (int y = y0; y < y0 + size && y < grid.dimY - 1; y++)
            for (int x = x0; x < x0 + size && x < grid.dimX - 1; x++) {
                vec3 pos[8];
                float val[8];
                int cubeIndex = 0;
                for (int i = 0; i < 8; i++) {
                    const int cx = x + kCorner[i][0], cy = y + kCorner[i][1], cz = z + kCorner[i][2];
                    pos[i] = vec3(grid.minX + cx * vx, grid.minY + cy * vx, grid.minZ + cz * vx);
                    val[i] = scalar(cx, cy, cz);
                    if (val[i] < 0) cubeIndex |= 1 << i;
                }
                const char* edges = kMcCaseEdges[cubeIndex];
                if (edges[0] == 'f') continue;                       // no sign change in this cell
                vec3 vert[12];
                bool have[12] = { false };
                for (const char* e = edges; *e != 'f'; e++) {
                    const int id = hexval(*e);
                    if (have[id]) continue;
                    const int a = edgeToCorner[id][0], b = edgeToCorner[id][1];
                    vert[id] = vertexInterp(0.0f, pos[a], pos[b], val[a], val[b]);
                    have[id] = true;
                }
                for (const char* e = edges; *e != 'f'; e += 3) {
                    MCTriangle tri;
                    tri.v[0] = vert[hexval(e[0])];
                    tri.v[1] = vert[hexval(e[1])];
                    tri.v[2] = vert[hexval(e[2])];
                    const vec3 n = rtmath::normalize(rtmath::cross(tri.v[1] - tri.v[0], tri.v[2] - tri.v[0]));
                    tri.normal[0] = tri.normal[1] = tri.normal[2] = n;
                    emit(tri);
                }
            }
}
}  // namespace
